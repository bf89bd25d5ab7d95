// Calibration: what one CU pays per 1-KiB wave store instruction (global_store_dwordx4), four waves per CU, for the
// access patterns an output-tile epilogue can produce:
//   0  coalesced: 64 lanes x 16 B contiguous (the LDS-transposed flush: two pixels x 512 B per instruction)
//   1  register epilogue, fp32 output: lane (pixel i = lane & 31, half kh = lane >> 5) writes 16 B of its own 64 B run
//      at pixel * 512 B + kh * 64 B + 16 j  (four instructions j = 0..3 complete the run)
//   2  the same with pixel stride 2048 B (512-channel layer)
//   3  register epilogue, split output: 16 B at pixel * stride + kh * 32 B + 16 (j & 1) + 64 (j >> 1)
// hipcc --offload-arch=gfx950 -O3 tools/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int STRIDE>
__global__ __launch_bounds__(256) void k(unsigned char* dst, unsigned long long* clk, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kh = lane >> 5;
  // each block owns 256 pixels x 2048 B; each wave 64 pixels
  unsigned char* base = dst + ((size_t)blockIdx.x * 256 + wave * 64) * (size_t)STRIDE;  // (L2-resident when STRIDE is small)
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const float4 v = make_float4(lane, wave, it, 2.f);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 8; ++t) {      // 8 "accumulator tiles": 32 store instructions per iteration, like one output tile
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned char* p;
        if (MODE == 0) p = base + (size_t)(t * 4 + j) * 1024 + lane * 16;                       // 2 px x 512 B
        else if (MODE == 1) p = base + (size_t)((t & 1) * 32 + i) * STRIDE + (t >> 1) * 128 + kh * 64 + j * 16;
                else p = base + (size_t)((t & 1) * 32 + i) * STRIDE + (t >> 1) * 128 + kh * 32 + (j & 1) * 16 + (j >> 1) * 64;
        *(float4*)p = v;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int MODE, int STRIDE>
void run(const char* name, unsigned char* dst, unsigned long long* clk) {
  const int iters = 50;
  k<MODE, STRIDE><<<256, 256>>>(dst, clk, 2);
  hipDeviceSynchronize();
  k<MODE, STRIDE><<<256, 256>>>(dst, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-44s %.0f cycles per 32 instructions per wave (4 waves/CU) = %.1f per CU-instruction, %.1f B/clk per CU\n", name,
         (double)h / iters, (double)h / iters / 128.0, 4 * 32 * 1024.0 * iters / (double)h);
}
int main() {
  unsigned char* dst; unsigned long long* clk;
  hipMalloc(&dst, (size_t)256 * 256 * 2048); hipMalloc(&clk, 8);
  run<0, 512>("coalesced (LDS-transposed flush)", dst, clk);
  run<1, 512>("register epilogue fp32, 512-B pixels", dst, clk);
  run<3, 512>("register epilogue split, 512-B pixels", dst, clk);
  run<1, 2048>("register epilogue fp32, 2048-B pixels", dst, clk);
  run<3, 2048>("register epilogue split, 2048-B pixels", dst, clk);
  return 0;
}
