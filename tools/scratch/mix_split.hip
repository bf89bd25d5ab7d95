// Is lo = fp16((x - f32(hi)) * 2048) -- the split-fp16 low part -- formed bit for bit by v_fma_mixlo / mixhi_f16 from the packed hi
// halves (fma(f32(hi), -2048, x * 2048), one rounding to fp16)?  Counts mismatches over random and edge-case inputs.
// hipcc --offload-arch=gfx950 -O3 tools/scratch/mix_split.hip -o tools/bin/mix_split && ./tools/bin/mix_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned lo_ref(f32x2 v, half2v h) {
  const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * 2048.0f;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2v));
}
__device__ inline unsigned lo_mix(f32x2 v, half2v h) {
  const f32x2 v2 = v * 2048.0f;
  const float k = -2048.0f;
  unsigned d = 0u;
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
               "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
               : "+v"(d) : "v"(hb), "s"(k), "v"(v2[0]), "v"(v2[1]));
  return d;
}
__global__ void k(const float* x, unsigned* bad, unsigned* first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const f32x2 v = {x[2 * i], x[2 * i + 1]};
  const half2v h = __builtin_convertvector(v, half2v);
  const unsigned a = lo_ref(v, h), b = lo_mix(v, h);
  if (a != b) {
    if (atomicAdd(bad, 1u) == 0u) { first[0] = __float_as_uint(v[0]); first[1] = __float_as_uint(v[1]); first[2] = a; first[3] = b; }
  }
}
int main() {
  const int n = 1 << 24;
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned m = s & 0x007fffffu, sign = (s >> 31) << 31;
    int e = 127 - 30 + (int)((s >> 23) & 63);          // 2^-30 .. 2^33: beyond fp16 at both ends (inf / subnormal / zero hi)
    if (e > 127 + 15) e = 127 + 15;                    // (|x| < 65504 like the range guard guarantees)
    unsigned u = sign | ((unsigned)e << 23) | m;
    if ((i & 1023) == 0) u = 0u;                       // exact zeros (post-ReLU maps)
    if ((i & 1023) == 1) u = sign | ((unsigned)e << 23);   // exact powers of two
    memcpy(&h[i], &u, 4);
  }
  float* dx; unsigned *db, *df;
  hipMalloc(&dx, n * 4); hipMalloc(&db, 4); hipMalloc(&df, 16);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(db, 0, 4); hipMemset(df, 0, 16);
  k<<<n / 2 / 256, 256>>>(dx, db, df, n);
  unsigned bad = 0, first[4];
  hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost); hipMemcpy(first, df, 16, hipMemcpyDeviceToHost);
  printf("pairs %d mismatches %u", n / 2, bad);
  if (bad) printf("  first: x = %08x %08x ref %08x mix %08x", first[0], first[1], first[2], first[3]);
  printf("\n");
  return bad != 0;
}
