// Does VALU work of one wave slow the MFMA stream of its SIMD partner?  512 threads per block, one block per CU: waves 0-3 (one
// per SIMD) run an MFMA stream shaped like the split-fp16 K loop (24 MFMAs per step over 8 accumulator tiles, optionally with the
// 16 ds_read_b128 fragment reads of a consumer step), waves 4-7 run vector work (plain fma chains, or a conv-epilogue-like mix of
// fma / max / cvt / pack with LDS stores).  Times: MFMA alone, VALU alone, both.  both ~ max: the pipes overlap; both ~ sum: they do not.
// hipcc --offload-arch=gfx950 -O3 tools/scratch/coissue.hip -o tools/bin/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int LDSREAD, int FLAVOR, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 64 * 1024 / 4; i += 512) ((unsigned*)smem)[i] = (mix(i + 1) & 0x83FF83FFu) | 0x30003000u;
  __syncthreads();
  if (wave < 4) {
    if (!(mode & 1)) return;
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    f32x16 acc[4][2];
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    half8 A[8], B[4];
    for (int i = 0; i < 8; ++i) A[i] = *(const half8*)(smem + ((lane + 64 * i) * 16));
    for (int i = 0; i < 4; ++i) B[i] = *(const half8*)(smem + 16384 + ((lane + 64 * i) * 16));
    for (int it = 0; it < mfma_iters; ++it) {
      if (LDSREAD) {
        const int o = (it & 15) * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = *(const half8*)(smem + o + ((lane + 64 * i) * 16));
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = *(const half8*)(smem + 32768 + o + ((lane + 64 * i) * 16));
      }
#pragma unroll
      for (int prod = 0; prod < 3; ++prod)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int tm = s >> 1, tn = s & 1;
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(B[2 * tn + (prod == 1)], A[2 * tm + (prod == 2)], acc[tm][tn], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[a][c][r];
    if (s == 12345.f) out[blockIdx.x * 512 + tid] = s;
  } else {
    if (!(mode & 2)) return;
    if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
    if (FLAVOR == 0) {
      float v[16];
      for (int r = 0; r < 16; ++r) v[r] = 1.0f + lane * 0.001f + r;
      const float a = 1.0001f, b = 0.5f;
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], a, b);
      }
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += v[r];
      if (s == 12345.f) out[blockIdx.x * 512 + tid] = s;
    } else {
      // conv-epilogue-like: 16 values: fma, add bias, relu, split to fp16 hi / lo pairs, amax, one 16-byte LDS store per 4 values
      float cm[16], cc[16];
      for (int r = 0; r < 16; ++r) { cm[r] = lane * 0.01f + r; cc[r] = 0.25f * r - lane; }
      float amax = 0.f;
      unsigned char* dst = smem + 49152 + tid * 32;
      for (int it = 0; it < valu_iters; ++it) {
        float hi8[8], lo8[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float x0 = fmaxf(__builtin_fmaf(cc[r], 1.0f / 2048.0f, cm[r]) + 0.125f, 0.f);
          float x1 = fmaxf(__builtin_fmaf(cc[r + 1], 1.0f / 2048.0f, cm[r + 1]) + 0.125f, 0.f);
          amax = fmaxf(amax, fmaxf(x0, x1));
          const half2v h = __builtin_convertvector(f32x2{x0, x1}, half2v);
          const half2v l = __builtin_convertvector((f32x2{x0, x1} - __builtin_convertvector(h, f32x2)) * 2048.0f, half2v);
          hi8[r >> 1] = __builtin_bit_cast(float, h);
          lo8[r >> 1] = __builtin_bit_cast(float, l);
          cm[r] += 0.5f; cm[r + 1] -= 0.25f;
        }
        *(float4*)dst = make_float4(hi8[0], hi8[1], hi8[2], hi8[3]);
        *(float4*)(dst + 16) = make_float4(hi8[4], hi8[5], hi8[6], hi8[7]);
        *(float4*)(dst + 16384) = make_float4(lo8[0], lo8[1], lo8[2], lo8[3]);
        *(float4*)(dst + 16384 + 16) = make_float4(lo8[4], lo8[5], lo8[6], lo8[7]);
      }
      if (amax == 12345.f) out[blockIdx.x * 512 + tid] = amax;
    }
  }
}

template <int L, int F, int P>
static void run(const char* name, int mi, int vi) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<L, F, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float t[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL((k<L, F, P>), dim3(256), dim3(512), 96 * 1024, 0, d, mi, vi, mode);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&t[mode], a, b);
    }
  }
  printf("%-44s mfma %.3f ms  valu %.3f ms  both %.3f ms   (sum %.3f, max %.3f)\n", name, t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
  hipFree(d);
}

int main() {
  // 24 MFMAs x 32 cycles = 768 cycles per mfma iter; FLAVOR 0: 128 fma per valu iter; FLAVOR 1: ~90 VALU + 4 ds_write_b128 per iter
  run<0, 0, 0>("bare MFMA | fma chains", 4000, 9000);
  run<1, 0, 0>("MFMA + frag reads | fma chains", 4000, 9000);
  run<1, 1, 0>("MFMA + frag reads | epilogue mix", 4000, 12000);
  run<1, 1, 1>("same, MFMA waves s_setprio 1", 4000, 12000);
  run<1, 1, 2>("same, VALU waves s_setprio 1", 4000, 12000);
  run<1, 1, 0>("MFMA + frag reads | epilogue mix, half load", 4000, 6000);
  run<1, 0, 0>("MFMA + frag reads | fma chains, half load", 4000, 4500);
  return 0;
}
