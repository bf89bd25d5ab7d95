// Do the four waves of a block pay for issuing their LDS-DMA pieces AT THE SAME TIME?  256 threads per block, one block per CU:
// every wave runs 144 MFMAs per "stage" (six half-steps of 24) and issues six 1-KiB global_load_lds pieces per stage, plus a
// barrier per stage like the conv kernel.  Variant 0: every wave issues piece k at the start of half-step k (what the dual-tile
// kernel does).  Variant 1: wave w issues its piece w * 6 MFMAs into the half-step (staggered by a quarter of a half-step).
// Variant 2: no DMA at all (the floor).  Variant 3: all six pieces at the start of the stage.
// hipcc --offload-arch=gfx950 -O3 tools/scratch/dma_stagger.hip -o tools/bin/dma_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int OFF>   // OFF: MFMA index inside the half-step after which the piece is issued (-1: none)
__device__ __forceinline__ void half_step(f32x16 (&acc)[8], const half8 (&A)[4], const half8 (&B)[2], const unsigned char* src, unsigned lds, unsigned lane16) {
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(B[i & 1], A[(i >> 1) & 3], acc[i & 7], 0, 0, 0);
    if (i == OFF) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane16), "s"(src));
  }
}

template <int VARIANT, int W>
__device__ void body(float* out, const unsigned char* wsrc, int stages, unsigned char* smem, int lane) {
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 A[4], B[2];
  for (int i = 0; i < 4; ++i) A[i] = *(const half8*)(smem + 65536 + (lane + 64 * i) * 16);
  for (int i = 0; i < 2; ++i) B[i] = *(const half8*)(smem + 65536 + 8192 + (lane + 64 * i) * 16);
  const unsigned lane16 = lane * 16;
  for (int st = 0; st < stages; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* src = wsrc + (size_t)((st & 63) * 24 + W * 6) * 1024;
    const unsigned lds0 = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(smem + ((st & 1) * 24 + W * 6) * 1024);
    if (VARIANT == 3) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + k * 1024), "v"(lane16), "s"(src + k * 1024));
    }
    constexpr int OFF = VARIANT == 0 ? 0 : VARIANT == 1 ? W * 6 : -1;
    half_step<OFF>(acc, A, B, src, lds0, lane16);
    half_step<OFF>(acc, A, B, src + 1024, lds0 + 1024, lane16);
    half_step<OFF>(acc, A, B, src + 2048, lds0 + 2048, lane16);
    half_step<OFF>(acc, A, B, src + 3072, lds0 + 3072, lane16);
    half_step<OFF>(acc, A, B, src + 4096, lds0 + 4096, lane16);
    half_step<OFF>(acc, A, B, src + 5120, lds0 + 5120, lane16);
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k(float* out, const unsigned char* wsrc, int stages) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 80 * 1024 / 4; i += 256) ((unsigned*)smem)[i] = ((i * 2654435761u) & 0x83FF83FFu) | 0x30003000u;
  __syncthreads();
  const unsigned char* ws = wsrc + (size_t)(blockIdx.x & 3) * (64 * 24 * 1024);
  if (wave == 0) body<VARIANT, 0>(out, ws, stages, smem, lane);
  else if (wave == 1) body<VARIANT, 1>(out, ws, stages, smem, lane);
  else if (wave == 2) body<VARIANT, 2>(out, ws, stages, smem, lane);
  else body<VARIANT, 3>(out, ws, stages, smem, lane);
}

template <int V>
static float run(float* d, unsigned char* w, int stages) {
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float t = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), 96 * 1024, 0, d, w, stages);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&t, a, b);
  }
  return t;
}

int main() {
  float* d; unsigned char* w;
  hipMalloc(&d, 256 * 256 * 4);
  hipMalloc(&w, 4 * 64 * 24 * 1024 + 65536);
  hipMemset(w, 0x3c, 4 * 64 * 24 * 1024 + 65536);
  const int stages = 2000;
  const float t2 = run<2>(d, w, stages), t0 = run<0>(d, w, stages), t1 = run<1>(d, w, stages), t3 = run<3>(d, w, stages);
  printf("per stage (144 MFMAs = 4608 cycles at the MFMA rate), us: no DMA %.3f | all waves at the half-step starts %.3f (+%.1f %%) | staggered by wave %.3f (+%.1f %%) | all six at the stage start %.3f (+%.1f %%)\n",
         1e3 * t2 / stages, 1e3 * t0 / stages, 100 * (t0 / t2 - 1), 1e3 * t1 / stages, 100 * (t1 / t2 - 1), 1e3 * t3 / stages, 100 * (t3 / t2 - 1));
  return 0;
}
