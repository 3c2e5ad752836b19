// Is r = x - bf16(x) exact when formed by v_dot2c_f32_bf16 (r = x + hi * (-1) + hi' * 0), i.e. can the
// three-piece split of csrc/common.h drop its two expand instructions per level?  Compares the split
// (hi, mid, lo) built with dot2 against the reference split bit for bit over random fp32 values of every
// binade, denormals, zeros, and pairs whose partner is huge / inf.
// hipcc --offload-arch=gfx950 -O3 -x hip tools/csrc/dot2_split_exact.hip -o /tmp/dot2_split_exact && /tmp/dot2_split_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <random>
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_ref(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){x0, x1}, bf16x2v));
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){r0, r1}, bf16x2v));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){s0, s1}, bf16x2v));
}
__device__ __forceinline__ void split_dot2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  // (-1, 0) and (0, -1) as opaque registers: written as constants the compiler encodes (-1, 0) as
  // the INLINE constant -1.0, which the instruction reads as the 32-bit pattern 0xbf800000 = (0, -1)
  uint32_t c0 = 0x0000bf80u, c1 = 0xbf800000u;
  asm volatile("" : "+s"(c0), "+s"(c1));
  const bf16x2v n0 = __builtin_bit_cast(bf16x2v, c0), n1 = __builtin_bit_cast(bf16x2v, c1);
  const bf16x2v hv = __builtin_convertvector((f32x2v){x0, x1}, bf16x2v);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hv, n0, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(hv, n1, x1, false);
  const bf16x2v mv = __builtin_convertvector((f32x2v){r0, r1}, bf16x2v);
  const float s0 = __builtin_amdgcn_fdot2_f32_bf16(mv, n0, r0, false), s1 = __builtin_amdgcn_fdot2_f32_bf16(mv, n1, r1, false);
  h = __builtin_bit_cast(uint32_t, hv); m = __builtin_bit_cast(uint32_t, mv);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){s0, s1}, bf16x2v));
}
__global__ void k(const float* x, int n, unsigned long long* bad, uint32_t* first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  uint32_t h0, m0, l0, h1, m1, l1;
  split_ref(x[2 * i], x[2 * i + 1], h0, m0, l0);
  split_dot2(x[2 * i], x[2 * i + 1], h1, m1, l1);
  if (h0 != h1 || m0 != m1 || l0 != l1) {
    if (atomicAdd(bad, 1ull) == 0) { first[0] = __float_as_uint(x[2 * i]); first[1] = __float_as_uint(x[2 * i + 1]);
      first[2] = m0; first[3] = m1; first[4] = l0; first[5] = l1; }
  }
}
int main() {
  const int n = 1 << 24;
  std::vector<float> h(n);
  std::mt19937 rng(7);
  const char* names[] = {"normal(0,1)", "every binade (random bits, finite)", "denormals and tiny", "pairs (x, huge)", "pairs (x, inf)"};
  float* dx; unsigned long long* dbad; uint32_t* dfirst;
  hipMalloc(&dx, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dfirst, 32);
  for (int mode = 0; mode < 5; ++mode) {
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int i = 0; i < n; ++i) {
      float v;
      if (mode == 0) v = nd(rng);
      else if (mode == 1) { uint32_t u; do { u = rng(); } while (((u >> 23) & 0xFF) == 0xFF); memcpy(&v, &u, 4); }
      else if (mode == 2) { uint32_t u = rng() & 0x80FFFFFFu; if (i % 3 == 0) u |= (1u << 23); if (i % 7 == 0) u = 0; memcpy(&v, &u, 4); }
      else { v = nd(rng); if (i & 1) v = mode == 3 ? 3.0e38f : INFINITY; }
      h[i] = v;
    }
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    k<<<n / 2 / 256, 256>>>(dx, n, dbad, dfirst);
    unsigned long long bad; uint32_t f[8];
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, dfirst, 32, hipMemcpyDeviceToHost);
    printf("%-36s %d pairs: %llu differ", names[mode], n / 2, bad);
    if (bad) printf("  (first: x = %08x %08x, mid %08x vs %08x, lo %08x vs %08x)", f[0], f[1], f[2], f[3], f[4], f[5]);
    printf("\n");
  }
  return 0;
}
