// dev probe: validate the DPP / permlane cross-lane sums used by scan.hip against a host reference
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float swap32_add(float a, float b) {
  const v2u r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned x = r.x, y = r.y;   // (indexing r[0]/r[1] through a bit_cast is miscompiled by ROCm 7.2: both read element 0)
  return __uint_as_float(x) + __uint_as_float(y);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  const v2u r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned x = r.x, y = r.y;   // (indexing r[0]/r[1] through a bit_cast is miscompiled by ROCm 7.2: both read element 0)
  return __uint_as_float(x) + __uint_as_float(y);
}
template <int G> __device__ __forceinline__ float group_allsum_f32(float v) {
  if (G >= 2) v += dpp_mov<0xB1>(v);
  if (G >= 4) v += dpp_mov<0x4E>(v);
  if (G >= 8) v += dpp_mov<0x141>(v);
  if (G >= 16) v += dpp_mov<0x140>(v);
  if (G >= 32) v = swap16_add(v, v);
  if (G >= 64) v = swap32_add(v, v);
  return v;
}
__device__ __forceinline__ float reduce4_rows(float p0, float p1, float p2, float p3) {
  const float m01 = swap32_add(p0, p1);
  const float m23 = swap32_add(p2, p3);
  float m = swap16_add(m01, m23);
  m += dpp_mov<0xB1>(m);
  m += dpp_mov<0x4E>(m);
  m += dpp_mov<0x141>(m);
  m += dpp_mov<0x140>(m);
  return m;
}
__global__ void probe(const float* in, float* out) {
  int l = threadIdx.x;
  float v = in[l];
  out[0 * 64 + l] = group_allsum_f32<2>(v);
  out[1 * 64 + l] = group_allsum_f32<4>(v);
  out[2 * 64 + l] = group_allsum_f32<8>(v);
  out[3 * 64 + l] = group_allsum_f32<16>(v);
  out[4 * 64 + l] = group_allsum_f32<32>(v);
  out[5 * 64 + l] = group_allsum_f32<64>(v);
  out[6 * 64 + l] = reduce4_rows(in[l], in[64 + l], in[128 + l], in[192 + l]);
}
int main() {
  float h[256], o[7 * 64];
  for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 101) + 0.5f;
  float *d, *e;
  (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&e, sizeof o);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, e);
  (void)hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
  int Gs[6] = {2, 4, 8, 16, 32, 64};
  for (int k = 0; k < 6; ++k) {
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      float s = 0; int g0 = l / Gs[k] * Gs[k];
      for (int j = 0; j < Gs[k]; ++j) s += h[g0 + j];
      if (fabsf(s - o[k * 64 + l]) > 1e-3f) { if (!bad) printf("G=%d lane %d got %g want %g\n", Gs[k], l, o[k * 64 + l], s); ++bad; }
    }
    printf("G=%d bad=%d\n", Gs[k], bad);
  }
  int perm[4] = {0, 2, 1, 3}, bad = 0;
  for (int l = 0; l < 64; ++l) {
    int u = perm[l >> 4]; float s = 0;
    for (int j = 0; j < 64; ++j) s += h[u * 64 + j];
    if (fabsf(s - o[6 * 64 + l]) > 1e-3f) { if (!bad) printf("reduce4 lane %d got %g want %g\n", l, o[6 * 64 + l], s); ++bad; }
  }
  printf("reduce4 bad=%d\n", bad);
  return 0;
}
