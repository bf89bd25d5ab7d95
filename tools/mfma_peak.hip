// Calibration: pure v_mfma_f32_32x32x2_f32 issue rate on this box (no memory traffic).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && ./tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, acc[3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES>
void run(int blocks, const char* name) {
  float* d; hipMalloc(&d, (size_t)blocks * WAVES * 64 * 4);
  const int iters = 4000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<WAVES><<<blocks, WAVES * 64>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<WAVES><<<blocks, WAVES * 64>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double fl = (double)blocks * WAVES * iters * 16 * 4096.0;
  printf("%-28s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  hipFree(d);
}
int main() {
  run<4>(256, "4 waves/CU (1/SIMD)");
  run<4>(512, "8 waves/CU (2/SIMD, 2 blk)");
  run<8>(256, "8 waves/CU (2/SIMD, 1 blk)");
  run<4>(1024, "16 waves/CU");
  return 0;
}
