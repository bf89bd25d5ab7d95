// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950?  (decides whether the split-fp16 scheme can keep
// its low parts UNSCALED and fold all three products into ONE accumulator: see DESIGN.md, next levers)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float tests[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /* 2^-20: subnormal */, 1.0f}, {5.9604644775390625e-08f /* 2^-24: smallest */, 1.0f},
                            {3.0517578125e-05f /* 2^-15: subnormal */, 3.0517578125e-05f}, {6.103515625e-05f /* 2^-14: min normal */, 1.0f}};
  for (auto& t : tests) {
    k<<<1, 64>>>(t[0], t[1], d);
    float h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %.10e  b = %.10e  ->  D = %.10e  (exact %.10e)\n", t[0], t[1], h, 16.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
  }
  return 0;
}
