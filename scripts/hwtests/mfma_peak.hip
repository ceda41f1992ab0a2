// Hardware probe: what does the matrix pipe SUSTAIN on this box?  Pure MFMA loops (no memory traffic), 2 waves per
// SIMD on every CU, for the two bf16 shapes the implicit-GEMM kernel uses.  Prints TFLOP/s over a ~0.3 ms run and
// over a ~3 ms run (clock management reacts within the first few hundred microseconds of a dense MFMA burst).
// build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// TOGGLE: the operands change every iteration (xorshift on the packed words, 8 VALU per 16 MFMAs) - random-looking bf16
// data as in a real GEMM, instead of constant registers whose datapath barely switches.
template <int SHAPE, bool TOGGLE = false, int LDSR = 0, int GLD = 0>
__global__ __launch_bounds__(512) void k(int iters, float* out, const u32x4* __restrict__ gbuf = nullptr) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];          // 64 KB, read with conflict-free 16-byte accesses
  if (LDSR) { for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = u32x4{(unsigned)i * 2654435761u, (unsigned)i, 0x3f803f80u, 0x3f003f00u}; __syncthreads(); }
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f - e * 0.01f); }
  u32x4 sa = __builtin_bit_cast(u32x4, a), sb = __builtin_bit_cast(u32x4, b);
  auto step = [&](int it) {
    if constexpr (LDSR > 0) {       // operands come from LDS: LDSR reads per 16 MFMAs, the last two feed the MFMAs
      u32x4 acc4 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < LDSR; ++r) {
        const u32x4 v = lds[(threadIdx.x + 64 * r + 37 * it) & 4095];
        if (r == LDSR - 1) sa = v; else if (r == LDSR - 2) sb = v; else { acc4[0] ^= v[0]; acc4[1] ^= v[1]; }
      }
      if constexpr (GLD > 0) {      // weights straight from L1/L2 (a 64 KB table shared by all workgroups)
#pragma unroll
        for (int r = 0; r < GLD; ++r) { const u32x4 v = gbuf[(threadIdx.x + 64 * r + 41 * it) & 4095]; acc4[0] ^= v[0]; acc4[1] ^= v[1]; }
      }
      sa[3] ^= acc4[0] & 1u; sb[3] ^= acc4[1] & 1u;
      u32x4 ma, mb;
      for (int e = 0; e < 4; ++e) { ma[e] = (sa[e] & 0x807f807fu) | 0x3f003f00u; mb[e] = (sb[e] & 0x807f807fu) | 0x3f003f00u; }
      a = __builtin_bit_cast(bf16x8, ma); b = __builtin_bit_cast(bf16x8, mb);
    } else if constexpr (TOGGLE) {
      for (int e = 0; e < 4; ++e) {
        sa[e] ^= sa[e] << 13; sa[e] ^= sa[e] >> 17; sa[e] ^= sa[e] << 5;
        sb[e] = sb[e] * 1664525u + 1013904223u;
      }
      // keep exponents moderate so that nothing overflows to inf (power follows the mantissa/sign toggling anyway)
      u32x4 ma, mb;
      for (int e = 0; e < 4; ++e) { ma[e] = (sa[e] & 0x807f807fu) | 0x3f003f00u; mb[e] = (sb[e] & 0x807f807fu) | 0x3f003f00u; }
      a = __builtin_bit_cast(bf16x8, ma); b = __builtin_bit_cast(bf16x8, mb);
    }
  };
  float r = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 c[16];
    for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      step(it);
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) r += c[i][0];
  } else {
    f32x16 c[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      step(it);
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) r += c[i][0];
  }
  if (r == 123.456f) out[0] = r;
}

template <int SHAPE, bool TOGGLE, int LDSR = 0, int GLD = 0>
void run(const char* name, int iters, int threads = 512) {
  float* out; hipMalloc(&out, 4);
  u32x4* gbuf; hipMalloc(&gbuf, 65536); hipMemset(gbuf, 0x3f, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE, TOGGLE, LDSR, GLD><<<256, threads>>>(iters, out, gbuf);   // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE, TOGGLE, LDSR, GLD><<<256, threads>>>(iters, out, gbuf);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per wave per iteration: SHAPE 16: 16 MFMAs x 16*16*32 MACs; SHAPE 32: 8 x 32*32*16 MACs  (both 131072 MACs)
  const double flops = 2.0 * 131072.0 * iters * (threads / 64) /*waves*/ * 256 /*CUs*/;
  printf("%s iters %7d: %8.3f ms  %7.1f TFLOP/s\n", name, iters, ms, flops / ms / 1e9);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<16, false>("16x16x32 constant operands", 2000);
    run<16, false>("16x16x32 constant operands", 20000);
    run<32, false>("32x32x16 constant operands", 20000);
    run<16, true>("16x16x32 changing operands", 2000);
    run<16, true>("16x16x32 changing operands", 20000);
    run<32, true>("32x32x16 changing operands", 2000);
    run<32, true>("32x32x16 changing operands", 20000);
    run<16, true, 2>("16x16x32 operands from LDS, 2 reads/16 MFMA", 20000);
    run<16, true, 4>("16x16x32 operands from LDS, 4 reads/16 MFMA", 20000);
    run<16, true, 5>("16x16x32 operands from LDS, 5 reads/16 MFMA", 20000);
    run<16, true, 6>("16x16x32 operands from LDS, 6 reads/16 MFMA (the 256x256 kernel's ratio)", 20000);
    run<16, true, 8>("16x16x32 operands from LDS, 8 reads/16 MFMA", 20000);
    run<16, true, 0>("ONE wave per SIMD: 16x16x32 changing operands", 20000, 256);
    run<16, true, 2>("ONE wave per SIMD: 2 LDS reads/16 MFMA", 20000, 256);
    run<16, true, 4>("ONE wave per SIMD: 4 LDS reads/16 MFMA (128x128 wave tiles)", 20000, 256);
    run<16, true, 4, 2>("16x16x32: 4 LDS reads + 2 global (L1-hit) loads /16 MFMA", 20000);
    run<16, true, 2, 4>("16x16x32: 2 LDS reads + 4 global (L1-hit) loads /16 MFMA", 20000);
    run<16, true, 12>("16x16x32 operands from LDS, 12 reads/16 MFMA", 20000);
  }
  return 0;
}
