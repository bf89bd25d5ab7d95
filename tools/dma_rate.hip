// Calibration: what one wave pays to ISSUE weight-staging memory instructions (cycles per 1-KiB wave instruction),
// L2-resident source: LDS DMA (global_load_lds_dwordx4, raw_buffer_load_lds x4) against a plain global_load_dwordx4.
// hipcc --offload-arch=gfx950 -O3 tools/dma_rate.hip -o tools/dma_rate && ./tools/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * 65536 + wave * 16384;
  float4 acc = make_float4(0, 0, 0, 0);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 16384, 0x00020000);
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned char* p = base + j * 1024 + lane * 16;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(lds + wave * 16384 + j * 1024), 16, 0, 0);
      } else if (MODE == 1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 16384 + j * 1024), 16,
                                             lane * 16, j * 1024, 0, 0);
      } else {
        const float4 v = *(const float4*)p;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + lds[threadIdx.x * 16];
  if (blockIdx.x == 7 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int MODE>
void run(int waves, const char* name, const unsigned char* src, float* out, unsigned long long* clk) {
  const int iters = 200;
  k<MODE><<<256, waves * 64, 65536>>>(src, out, clk, 2);
  hipDeviceSynchronize();
  k<MODE><<<256, waves * 64, 65536>>>(src, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-34s %d wave(s)/CU: %.0f cycles per 1-KiB instruction per wave (incl. the drain every 16), %.1f B/clk per CU\n", name, waves,
         (double)h / (iters * 16.0), waves * 1024.0 * iters * 16.0 / (double)h);
}
int main() {
  unsigned char* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 256 * 65536); hipMemset(src, 1, 256 * 65536);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 8);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int w = 1; w <= 4; w *= 4) {
    run<0>(w, "global_load_lds_dwordx4", src, out, clk);
    run<1>(w, "raw_buffer_load_lds (16 B)", src, out, clk);
    run<2>(w, "global_load_dwordx4 (to VGPRs)", src, out, clk);
  }
  return 0;
}
