// Calibration: pure MFMA issue rate on this box (no memory traffic), fp32 (v_mfma_f32_32x32x2_f32) and
// fp16 (v_mfma_f32_32x32x16_f16), with the sustained shader clock (s_memtime cycles / s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && ./tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int WAVES, bool F16>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
  half8 hx, hy;
  for (int j = 0; j < 8; ++j) { hx[j] = (_Float16)(x + j); hy[j] = (_Float16)(y - j); }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (F16) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hx, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hx, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hy, acc[3], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, acc[3], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int WAVES, bool F16>
void run(int blocks, const char* name) {
  float* d; hipMalloc(&d, (size_t)blocks * WAVES * 64 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  const int iters = F16 ? 20000 : 4000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<WAVES, F16><<<blocks, WAVES * 64>>>(d, 10, clk);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<WAVES, F16><<<blocks, WAVES * 64>>>(d, iters, clk);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double fl = (double)blocks * WAVES * iters * 16 * (F16 ? 32768.0 : 4096.0);
  printf("%-5s %-28s blocks %5d  %.3f ms  %.1f TFLOP/s   %.0f memtime ticks/MFMA/SIMD, memtime/realtime %.2f\n", F16 ? "f16" : "f32", name,
         blocks, ms, fl / ms / 1e9, (double)h[0] / (iters * 16.0) / ((WAVES * (blocks / 256) + 3) / 4),
         (double)h[0] / (double)h[1]);
  hipFree(d); hipFree(clk);
}
int main() {
  run<4, false>(256, "4 waves/CU (1/SIMD)");
  run<8, false>(256, "8 waves/CU (2/SIMD, 1 blk)");
  run<4, true>(256, "4 waves/CU (1/SIMD)");
  run<8, true>(256, "8 waves/CU (2/SIMD, 1 blk)");
  run<4, true>(1024, "16 waves/CU");
  return 0;
}
