// Hardware probe: cycles per ds_read_b128 wave-instruction for a given per-lane address pattern (bank-conflict model of gfx950).
// One workgroup of 8 waves (2 per SIMD) on one CU; every wave issues `reads` ds_read_b128 at lane_off[lane] (+ a per-wave base),
// back to back, and the shader-clock time of the slowest wave is divided by the reads of one wave: 8 waves share the LDS port,
// so a conflict-free pattern costs 8 waves x 1 KiB / (bytes per clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const unsigned* lane_off, int reads, unsigned long long* out, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* p = smem + lane_off[lane] + (wave & 1) * 32768;
  u32x4 acc = {0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reads; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(p)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      acc ^= v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0;
  sink[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
static double run(const std::vector<unsigned>& off, int reads = 4096) {
  unsigned* d; unsigned long long* o; unsigned* s;
  hipMalloc(&d, 256); hipMalloc(&o, 64); hipMalloc(&s, 2048);
  hipMemcpy(d, off.data(), 256, hipMemcpyHostToDevice);
  k<<<1, 512>>>(d, reads, o, s);
  k<<<1, 512>>>(d, reads, o, s);
  unsigned long long h[8];
  hipMemcpy(h, o, 64, hipMemcpyDeviceToHost);
  unsigned long long m = 0; for (int i = 0; i < 8; ++i) m = h[i] > m ? h[i] : m;
  hipFree(d); hipFree(o); hipFree(s);
  return (double)m / reads;
}
int main() {
  auto report = [&](const std::string& name, const std::vector<unsigned>& off) { printf("%-64s %6.1f clk per wave-read (8 waves)\n", name.c_str(), run(off)); };
  std::vector<unsigned> off(64);
  // linear: lane l reads bytes [16 l, 16 l + 16): the conflict-free reference
  for (int l = 0; l < 64; ++l) off[l] = l * 16; report("linear 1 KiB", off);
  // all lanes the same bank group, different rows: worst case
  for (int l = 0; l < 64; ++l) off[l] = l * 256; report("stride 256 B (64-way on 64 banks)", off);
  for (int l = 0; l < 64; ++l) off[l] = l * 128; report("stride 128 B", off);
  // gather tile (tile 6): 128-byte rows, chunk ^ (row >> 1) & 7, rows fr, chunk 4 kk + fh
  for (int kk = 0; kk < 2; ++kk) { for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; off[l] = fr * 128 + (((4 * kk + fh) ^ ((fr >> 1) & 7)) << 4); }
    report("tile 6: rows x 128 B, key (row >> 1) & 7, kk = " + std::to_string(kk), off); }
  // patch kernel 1: pixel (R * 18 + fr + tx) * 128 B, chunk ^ ((fr + tx) >> 1) & 7
  for (int tx = 0; tx < 3; ++tx) for (int R = 0; R < 2; ++R) { for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; off[l] = (R * 18 + fr + tx) * 128 + ((fh ^ (((fr + tx) >> 1) & 7)) << 4); }
    report("patch 1: pitch 18, key (col >> 1) & 7, tx = " + std::to_string(tx) + " row " + std::to_string(R), off); }
  // patch kernel 2: pixel idx * 64 B, chunk ^ (idx >> 2) & 3, pitch 20
  for (int tx = 0; tx < 3; ++tx) for (int R = 0; R < 2; ++R) { for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; int idx = R * 20 + fr + tx; off[l] = idx * 64 + ((fh ^ ((idx >> 2) & 3)) << 4); }
    report("patch 2: 64-B pixels, key (idx >> 2) & 3, tx = " + std::to_string(tx) + " row " + std::to_string(R), off); }
  // weights of patch kernel 2: rows x 64 B, key (row >> 2) & 3
  for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; off[l] = fr * 64 + ((fh ^ ((fr >> 2) & 3)) << 4); } report("64-B rows, key (row >> 2) & 3, aligned", off);
  // 64-B pixels, no swizzle at all: 16 consecutive pixels = 1 KiB contiguous
  for (int tx = 0; tx < 3; ++tx) { for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; off[l] = (fr + tx) * 64 + fh * 16; } report("64-B pixels, NO swizzle, tx = " + std::to_string(tx), off); }
  // 128-B pixels, chunks 0-3 of each, no swizzle
  for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; off[l] = fr * 128 + fh * 16; } report("128-B rows, NO swizzle (chunks 0-3)", off);
  // 128-B pixels, key = pixel index >> 1 (the tile-6 rule applied to the pixel index), tx shifts
  for (int tx = 0; tx < 3; ++tx) { for (int l = 0; l < 64; ++l) { int fr = l & 15, fh = l >> 4; int idx = 18 + fr + tx; off[l] = idx * 128 + ((fh ^ ((idx >> 1) & 7)) << 4); } report("128-B pixels, key (idx >> 1) & 7, tx = " + std::to_string(tx), off); }
  return 0;
}
