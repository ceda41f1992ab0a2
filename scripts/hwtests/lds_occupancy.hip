// Hardware probe: how many 256-thread workgroups with a given LDS allocation does one CU of gfx950 hold (occupancy API + a timing check:
// a kernel in which every workgroup spins for a fixed time - 512 workgroups take one or two rounds).
// build: hipcc -O3 --offload-arch=gfx950 lds_occupancy.hip -o lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(long clocks, int* out) {
  extern __shared__ char lds[];
  const long t0 = clock64();
  lds[threadIdx.x] = (char)threadIdx.x;
  while (clock64() - t0 < clocks) {}
  if (lds[(threadIdx.x + 1) & 255] == 77 && clocks < 0) out[0] = 1;
}
int main() {
  int dev_lds = 0; hipDeviceGetAttribute(&dev_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
  int blk_lds = 0; hipDeviceGetAttribute(&blk_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
  printf("LDS per CU (attribute) %d, per block %d\n", dev_lds, blk_lds);
  int* out; hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int kb : {32, 64, 72, 76, 78, 79, 80, 81, 96}) {
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin, 256, kb * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<<<512, 256, kb * 1024>>>(100000, out); hipDeviceSynchronize();
    hipEventRecord(e0); spin<<<512, 256, kb * 1024>>>(100000, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%3d KiB per workgroup: occupancy API %d per CU; 512 workgroups spinning 1 ms-ish each: %.1f us (%s)\n", kb, nb, ms * 1e3, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
