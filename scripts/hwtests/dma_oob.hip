// Hardware probe: does an out-of-range `buffer_load_dwordx4 ... lds` write zeros into LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, int bytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned smem[256];
  for (int i = threadIdx.x; i < 256; i += 64) smem[i] = 0xFFFFFFFFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // even lanes in range, odd lanes masked with the 0x80000000 offset; last in-range lane straddles the end
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, (int)off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
  unsigned *d, *o; std::vector<unsigned> h(256), r(256);
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, 1024 - 24, o);     // last 24 bytes out of range
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    unsigned got = r[l * 4 + e], want;
    int byte = l * 16 + e * 4;
    want = (l & 1) ? 0u : (byte + 4 <= 1024 - 24 ? h[l * 4 + e] : 0u);
    if (got != want) { if (bad < 12) printf("lane %d elem %d got %08x want %08x\n", l, e, got, want); ++bad; }
  }
  printf("dma_oob: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK: masked/out-of-range LDS-DMA lanes write zeros", bad);
  return 0;
}
