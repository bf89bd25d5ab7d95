// Calibration: what the matrix pipe SUSTAINS on this box when it is fed like the split-fp16 convolution feeds it --
// one wave per SIMD, 8 accumulator tiles, three products per (A, B) pair -- as a function of what the operands look
// like.  No memory traffic at all: the difference between the rows is the power the operands' bit toggling costs (the
// chip lowers its clock under the matrix load), i.e. the ceiling a real kernel can reach before it moves a single byte.
// Clock and pipe occupancy per row: tools/mfma_power_table.py over a rocprofv3 counter pass of this program.
//   TYPE    0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_16x16x32_f16
//   RANDOM  0: every fragment register holds 1.0   1: random sign + mantissa, exponents spread over 8 binades
//   REFRESH 1: every fragment gets new mantissa / sign bits every 24 MFMAs (a half-step of the conv)
//   ZERO8   eighths of the ACTIVATION dwords (second MFMA source) that are zero (post-ReLU maps); 16 + n: n eighths of the
//           WEIGHT dwords (first MFMA source) instead
//   ORDER   0: the conv kernel's order (product outer, tm, tn inner)   1: snake (one operand changes per MFMA)
//           2: product INNERMOST -- the three products of an accumulator tile back to back (dependent on SrcC: does the pipe
//              forward the accumulator, and does that save register-file energy?)   3: the same over pairs of tiles (two chains
//              interleaved)   4: over quads of tiles
//   TRUNC   low mantissa bits forced to zero in the lo fragments (a[odd], b[odd])
// hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/bin/mfma_power && ./tools/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// two 16-bit floats per dword: random sign + mantissa, exponent field spread over 8 binades around 1.0
template <int TYPE>
__device__ inline unsigned rnd_pair(unsigned h) {
  if (TYPE == 1) {   // bf16: s | e8 | m7
    const unsigned e0 = 123 + (h & 7), e1 = 123 + ((h >> 3) & 7);
    return (h & 0x807F807Fu) | (e0 << 7) | (e1 << 23);
  }
  const unsigned e0 = 11 + (h & 7), e1 = 11 + ((h >> 3) & 7);   // fp16: s | e5 | m10
  return (h & 0x83FF83FFu) | (e0 << 10) | (e1 << 26);
}

template <int TYPE, int RANDOM, int REFRESH, int ZERO8, int ORDER, int TRUNC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  constexpr unsigned MANT = TYPE == 1 ? 0x807F807Fu : 0x83FF83FFu;
  constexpr unsigned ONE = TYPE == 1 ? 0x3F803F80u : 0x3C003C00u;
  constexpr unsigned TMASK = ~(((1u << TRUNC) - 1u) * 0x00010001u);
  f32x16 acc[4][2];
  for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  u4 A[8], B[4], K[8];   // a[2 tm] hi, a[2 tm + 1] lo (activations; K: keep masks); b[2 tn] hi, b[2 tn + 1] lo (weights)
  unsigned seed = mix(blockIdx.x * 256u + threadIdx.x + 1u);
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) {
    seed = mix(seed + 0x9e3779b9u);
    A[i][j] = RANDOM == 0 ? ONE : rnd_pair<TYPE>(seed);
    K[i][j] = (ZERO8 < 16 && (int)((seed >> 16) & 7) < ZERO8) ? 0u : 0xFFFFFFFFu;
    if (i & 1) K[i][j] &= TMASK;
    A[i][j] &= K[i][j];
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    seed = mix(seed + 0x9e3779b9u);
    B[i][j] = RANDOM == 0 ? ONE : rnd_pair<TYPE>(seed);
    if (i & 1) B[i][j] &= TMASK;
    if (ZERO8 >= 16 && (int)((seed >> 16) & 7) < ZERO8 - 16) B[i][j] = 0u;
  }
  for (int it = 0; it < iters; ++it) {
    if (REFRESH) {
      // new sign / mantissa bits, same exponents: the masks are wave-uniform (scalar unit), one vector instruction per
      // fragment dword = 48 per 24 MFMAs
      const unsigned ts = __builtin_amdgcn_readfirstlane(mix((unsigned)it + 77u));
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] ^= ((ts >> ((i + 2 * j) & 15)) * 0x00010001u) & MANT & K[i][j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          B[i][j] ^= ((ts >> ((i + 2 * j + 5) & 15)) * 0x00010001u) & MANT & ((i & 1) ? TMASK : 0xFFFFFFFFu);
    }
#pragma unroll
    for (int idx = 0; idx < 24; ++idx) {
        // (ORDER >= 2: groups of G = 1 / 2 / 4 accumulator tiles, the three products innermost over the group)
        constexpr int G = ORDER == 2 ? 1 : ORDER == 3 ? 2 : 4;
        const int prod = ORDER < 2 ? idx / 8 : (idx / G) % 3;
        const int s = ORDER < 2 ? idx % 8 : (idx / (3 * G)) * G + idx % G;
        const int tm = s >> 1, tn = ORDER == 1 ? ((s & 1) ^ (tm & 1)) : (s & 1);
        const u4 a = A[2 * tm + (prod == 2)], b = B[2 * tn + (prod == 1)];
        if constexpr (TYPE == 0)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, b), __builtin_bit_cast(half8, a), acc[tm][tn], 0, 0, 0);
        else if constexpr (TYPE == 1)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc[tm][tn], 0, 0, 0);
        else {
          // the same FLOPs as two 16x16x32 instructions (their results land in quarters of the 16 accumulator registers)
          f32x4 q0 = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
          f32x4 q1 = {acc[tm][tn][4], acc[tm][tn][5], acc[tm][tn][6], acc[tm][tn][7]};
          q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, b), __builtin_bit_cast(half8, a), q0, 0, 0, 0);
          q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), q1, 0, 0, 0);
          for (int r = 0; r < 4; ++r) { acc[tm][tn][r] = q0[r]; acc[tm][tn][4 + r] = q1[r]; }
        }
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
  for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[a][c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

template <int TYPE, int RANDOM, int REFRESH, int ZERO8, int ORDER, int TRUNC>
void run(const char* name) {
  const int blocks = 256, iters = 60000;     // 24 MFMAs x 60 000 x 32 cycles ~ 20-35 ms per launch, 12 launches
  float* d;
  CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  k<TYPE, RANDOM, REFRESH, ZERO8, ORDER, TRUNC><<<blocks, 256>>>(d, 1000);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 4; ++rep) k<TYPE, RANDOM, REFRESH, ZERO8, ORDER, TRUNC><<<blocks, 256>>>(d, iters);   // settle the power management
  CK(hipEventRecord(a));
  const int reps = 8;
  for (int rep = 0; rep < reps; ++rep) k<TYPE, RANDOM, REFRESH, ZERO8, ORDER, TRUNC><<<blocks, 256>>>(d, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double fl = (double)reps * blocks * 4 * (double)iters * 24 * 32768.0;
  const double tf = fl / ms / 1e9;
  printf("k<%d, %d, %d, %d, %d, %d>  %-34s %8.1f TFLOP/s issued  (%.3f of the 2500 dense 16-bit peak)\n", TYPE, RANDOM, REFRESH, ZERO8,
         ORDER, TRUNC, name, tf, tf / 2500.0);
  CK(hipFree(d));
}
int main() {
  run<0, 0, 0, 0, 0, 0>("f16 const");
  run<0, 1, 0, 0, 0, 0>("f16 random");
  run<0, 1, 0, 4, 0, 0>("f16 random, act half zero");
  run<0, 1, 0, 5, 0, 0>("f16 random, act 5/8 zero");
  run<0, 1, 0, 8, 0, 0>("f16 random, act all zero");
  run<0, 1, 0, 20, 0, 0>("f16 random, WEIGHTS half zero");
  run<0, 1, 0, 0, 1, 0>("f16 random, snake order");
  run<0, 1, 0, 4, 1, 0>("f16 random, half zero, snake");
  run<0, 1, 0, 4, 2, 0>("f16 random, half zero, acc chained");
  run<0, 1, 0, 4, 3, 0>("f16 random, half zero, 2 chains");
  run<0, 1, 0, 4, 4, 0>("f16 random, half zero, 4 chains");
  run<0, 1, 0, 4, 0, 0>("f16 random, act half zero (again)");
  run<0, 1, 0, 0, 2, 0>("f16 random dense, acc chained");
  run<0, 1, 0, 4, 0, 3>("f16 random, half zero, lo 8 bits");
  run<0, 1, 0, 4, 0, 5>("f16 random, half zero, lo 6 bits");
  run<0, 1, 1, 4, 0, 0>("f16 refresh, half zero");
  run<2, 1, 0, 4, 0, 0>("f16 16x16x32 random, half zero");
  run<1, 0, 0, 0, 0, 0>("bf16 const");
  run<1, 1, 0, 0, 0, 0>("bf16 random");
  run<1, 1, 0, 4, 0, 0>("bf16 random, act half zero");
  run<0, 0, 0, 0, 0, 0>("f16 const");
  return 0;
}
