// Hardware probe: what does ONE vector-memory instruction cost the matrix pipe of its SIMD on gfx950?
// 256 workgroups x 8 waves (two per SIMD, no barriers, no LDS reads); per iteration a wave issues 16 independent 16x16x32 bf16 MFMAs
// (256 clocks of its SIMD's matrix pipe shared with its partner: 512 clocks per iteration and SIMD if nothing else costs anything) and N
// vector-memory instructions of one kind on a 64 KiB table that stays in L1 / L2 (no HBM traffic):
//   dma16   buffer_load_dwordx4 ... lds   (LDS-DMA, 1 KiB per instruction: what streams weights / patches / x in the conv kernels)
//   load16  buffer_load_dwordx4 to registers (never waited for inside the loop)      load4   buffer_load_dword
//   store16 buffer_store_dwordx4                                                      store4  buffer_store_dword
// Output: clocks per iteration and SIMD (from the elapsed time at the clock the MFMA-only loop implies) and the slope per instruction.
// build: hipcc -O3 --offload-arch=gfx950 vmem_issue_cost.hip -o vmem_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

enum { DMA16 = 0, LOAD16 = 1, STORE16 = 2, LOAD4 = 3, STORE4 = 4 };

template <int KIND, int N, bool MFMA>
__global__ __launch_bounds__(512) void k(int iters, float* out, unsigned int* table) {
  __shared__ __attribute__((aligned(16))) char lds[8 * 4 * 1024];          // 4 KiB of LDS-DMA landing space per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(table, 0, 65536, 0x00020000);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + 0.01f * e); b[e] = (__bf16)(1.0f - 0.01f * e); }
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
  u32x4 sink = {0u, 0u, 0u, 0u};
  unsigned int sink1 = 0u;
  for (int it = 0; it < iters; ++it) {
    if (MFMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const unsigned int off = (unsigned int)(((it * N + n) * 1024 + wave * 8192 + lane * 16) & 65535);
      if (KIND == DMA16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + wave * 4096 + (n & 3) * 1024), 16, (int)off, 0, 0, 0);
      else if (KIND == LOAD16) { const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0)); sink[n & 3] ^= v[n & 3]; }
      else if (KIND == STORE16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)it, 1u, 2u, 3u}, rs, (int)off, 0, 0);
      else if (KIND == LOAD4) sink1 ^= (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off & ~15u) / 4 * 4, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b32((unsigned)it, rs, (int)off, 0, 0);
    }
    if (MFMA) {
#pragma unroll
      for (int i = 8; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += c[i][0];
  if (r == 123.456f || (sink[0] ^ sink[1] ^ sink[2] ^ sink[3] ^ sink1) == 0x12345u) out[0] = r;
}

static double g_clock_ghz = 0.0;      // the shader clock implied by the MFMA-only loop (512 clocks per iteration and SIMD)

template <int KIND, int N, bool MFMA>
double run(const char* name, double base_clk) {
  const int iters = 20000;
  float* out; hipMalloc(&out, 4);
  unsigned int* table; hipMalloc(&table, 65536); hipMemset(table, 0, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, N, MFMA><<<256, 512>>>(iters, out, table);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND, N, MFMA><<<256, 512>>>(iters, out, table);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns_iter = ms * 1e6 / iters;
  if (MFMA && N == 0) g_clock_ghz = 512.0 / ns_iter;
  const double clk = ns_iter * g_clock_ghz;
  if (N == 0) printf("%-34s %8.1f ns per iteration = %6.0f clocks per SIMD%s\n", name, ns_iter, clk, MFMA ? "   (defines the clock: 2 waves x 16 MFMAs x 16 clocks)" : "");
  else printf("%-34s %8.1f ns per iteration = %6.0f clocks per SIMD: +%5.0f clocks for 2 x %d instructions = %5.1f per instruction\n", name, ns_iter, clk, clk - base_clk, N, (clk - base_clk) / (2.0 * N));
  hipFree(out); hipFree(table);
  return clk;
}

template <int KIND> void sweep(const char* kind) {
  char name[64];
  snprintf(name, sizeof name, "MFMA + 1 x %s", kind); run<KIND, 1, true>(name, 512.0);
  snprintf(name, sizeof name, "MFMA + 2 x %s", kind); run<KIND, 2, true>(name, 512.0);
  snprintf(name, sizeof name, "MFMA + 4 x %s", kind); run<KIND, 4, true>(name, 512.0);
  snprintf(name, sizeof name, "no MFMA, 4 x %s", kind); run<KIND, 4, false>(name, 0.0);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<DMA16, 0, true>("16 MFMAs per wave, nothing else", 0.0);
    sweep<DMA16>("dma16");
    sweep<LOAD16>("load16");
    sweep<STORE16>("store16");
    sweep<LOAD4>("load4");
    sweep<STORE4>("store4");
  }
  printf("shader clock implied by the MFMA-only loop: %.2f GHz\n", g_clock_ghz);
  return 0;
}
