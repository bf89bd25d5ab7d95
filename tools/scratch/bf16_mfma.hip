#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A[32][16], B[16][32] as floats; device converts to bf16 the way the kernels do
__global__ void k(const float* A, const float* B, float* C, int use_bf) {
  const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    const float av = A[i * 16 + kh * 8 + j], bv = B[(kh * 8 + j) * 32 + i];
    if (use_bf) { a[j] = __builtin_bit_cast(_Float16, (__bf16)av); b[j] = __builtin_bit_cast(_Float16, (__bf16)bv); }
    else { a[j] = (_Float16)av; b[j] = (_Float16)bv; }
  }
  f32x16 c = {};
  if (use_bf) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = c[r];
}
int main() {
  float hA[512], hB[512], hC[1024], ref[1024];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 37 % 17) - 8) / 8.f; hB[i] = (float)((i * 53 % 23) - 11) / 16.f; }
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[r * 16 + k] * hB[k * 32 + c]; ref[r * 32 + c] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  for (int bf = 0; bf < 2; ++bf) {
    k<<<1, 64>>>(dA, dB, dC, bf);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    float e = 0; for (int i = 0; i < 1024; ++i) e = fmaxf(e, fabsf(hC[i] - ref[i]));
    printf("use_bf %d max err %g\n", bf, e);
  }
  return 0;
}
