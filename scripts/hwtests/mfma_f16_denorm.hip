// Hardware probe: does the fp16 MFMA keep SUBNORMAL fp16 operands, and does the f32 -> f16 conversion produce them?
// (the f32x3 product of okp_igemm_kernel.h splits fp32 operands into fp16 hi + lo halves; lo of a value below 0.12 is an
// fp16 subnormal - if the matrix pipe flushed those, the correction terms of small operands would be lost.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float* out) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a_val; b[e] = (_Float16)b_val; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)a[0]; out[3] = (float)(_Float16)(a_val - (float)(_Float16)a_val); }
}
int main() {
  float* o; hipMalloc(&o, 64);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1024.0f}, {1024.0f, 3.0e-6f}, {5.9604645e-8f, 16384.0f}, {0.1f + 1e-5f, 1.0f}};
  for (auto& cs : cases) {
    k<<<1, 64>>>(cs[0], cs[1], o);
    float h[4]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    const float a16 = (float)(_Float16)cs[0], b16 = (float)(_Float16)cs[1];
    printf("a=%g (fp16 %g, device cvt %g) b=%g: 32x32x16 -> %g, 16x16x32 -> %g; exact 16*a16*b16 = %g / 32*a16*b16 = %g; lo(a) = %g\n",
           cs[0], a16, h[2], cs[1], h[0], h[1], 16.0 * a16 * b16, 32.0 * a16 * b16, h[3]);
  }
  return 0;
}
