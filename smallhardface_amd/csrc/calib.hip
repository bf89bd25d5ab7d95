// Calibration of the roofline's MFMA bound on the box at hand (include/shf_hip.h: shf_calib_matrix_pipe).
//
// The dense 16-bit MFMA peak (2.5 PFLOP/s at 2.4 GHz) is only reachable with operands that do not toggle: under a
// matrix load the chip runs into its power limit and lowers the clock, and how far depends on the DATA.  This kernel is
// the convolution's matrix-pipe diet and nothing else -- one wave per SIMD, 8 accumulator tiles, 8 activation and 4
// weight fragment registers, three v_mfma_f32_32x32x16 products per pair in the conv kernels' order, no memory
// traffic, no LDS, no VALU in the loop -- with random signs and mantissas over 8 binades and a chosen fraction of the
// activation fragments zero (post-ReLU maps).  What it sustains is the ceiling a convolution kernel could reach on
// this box if moving its bytes cost no power at all; bench.py reports the dominant kernel's issued rate against it
// beside the nominal peak (tools/mfma_power.hip is the stand-alone form with more variants; DESIGN.md has the table).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// two 16-bit floats per dword: random sign and mantissa, exponents over the 8 binades 2^-4 .. 2^3
template <bool BF>
__device__ inline unsigned random_pair(unsigned h) {
  if (BF) return (h & 0x807F807Fu) | ((123 + (h & 7)) << 7) | ((123 + ((h >> 3) & 7)) << 23);
  return (h & 0x83FF83FFu) | ((11 + (h & 7)) << 10) | ((11 + ((h >> 3) & 7)) << 26);
}

template <bool BF>
__global__ __launch_bounds__(256) void matrix_pipe_kernel(float* out, int iters, int zero_eighths, int constant) {
  f32x16 acc[4][2];
  for (int a = 0; a < 4; ++a)
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  u4 A[8], B[4];   // A[2 tm] / A[2 tm + 1]: activation hi / lo fragments; B[2 tn] / B[2 tn + 1]: weight hi / lo
  unsigned seed = mix32(blockIdx.x * 256u + threadIdx.x + 1u);
  const unsigned one = BF ? 0x3F803F80u : 0x3C003C00u;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) {
      seed = mix32(seed + 0x9e3779b9u);
      A[i][j] = constant ? one : random_pair<BF>(seed);
      if ((int)((seed >> 16) & 7) < zero_eighths) A[i][j] = 0u;
    }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      seed = mix32(seed + 0x9e3779b9u);
      B[i][j] = constant ? one : random_pair<BF>(seed);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int prod = 0; prod < 3; ++prod)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const u4 a = A[2 * tm + (prod == 2)], b = B[2 * tn + (prod == 1)];
          if constexpr (BF)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc[tm][tn], 0, 0, 0);
          else
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, b), __builtin_bit_cast(half8, a), acc[tm][tn], 0, 0, 0);
        }
    if ((it & 63) == 63) {   // keep the sums finite
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][c][r] *= 0x1p-20f;
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a)
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 16; ++r) s += acc[a][c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

namespace shf {

int calib_matrix_pipe(int bf16, int zero_eighths, int constant, int iters, int reps, double* tflops) {
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return (int)e;
  float* d = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  float ms = 0.f;
  auto launch = [&](int n) {
    if (bf16) matrix_pipe_kernel<true><<<cus, 256, 0, s>>>(d, n, zero_eighths, constant);
    else matrix_pipe_kernel<false><<<cus, 256, 0, s>>>(d, n, zero_eighths, constant);
  };
  e = hipMalloc(&d, (size_t)cus * 256 * sizeof(float));
  if (e == hipSuccess) e = hipStreamCreate(&s);
  if (e == hipSuccess) e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  if (e == hipSuccess) {
    for (int r = 0; r < (reps + 1) / 2; ++r) launch(iters);   // the power management settles within a few launches
    e = hipEventRecord(a, s);
  }
  if (e == hipSuccess) {
    for (int r = 0; r < reps; ++r) launch(iters);
    e = hipEventRecord(b, s);
  }
  if (e == hipSuccess) e = hipEventSynchronize(b);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess && tflops)
    *tflops = (double)reps * cus * 4.0 * (double)iters * 24.0 * 32768.0 / ((double)ms * 1e9);
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (s) (void)hipStreamDestroy(s);
  if (d) (void)hipFree(d);
  return (int)e;
}

}  // namespace shf
