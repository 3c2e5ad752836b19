// Probe (GPU box): what ds_read_b64_tr_b16 (gfx950) returns, and the K-major operand recipe built on it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/csrc/tr_read_layout tools/csrc/tr_read_layout.hip
// Part 1: LDS holds u16 lds[i] = i; lane l passes the address of lds[4 l] (8 bytes per lane, the
//   wave covers 512 contiguous bytes); prints what every lane receives.  Hypothesis: each group
//   of 16 lanes transposes its 4 x 16 block: lane l, element j = lds[64 (l / 16) + 16 j + l % 16].
// Part 2: an operand tile stored K-MAJOR in LDS ([16 k][32 m] bf16 in [k/4][m/16][4][16] blocks)
//   is read with two such reads per lane into the v_mfma_f32_32x32x16_bf16 A/B layout (lane l:
//   row l % 32, k = 8 (l / 32) + 0..7) and checked through the matrix instruction: C = A^T-stored
//   x B^T-stored against the host product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

__global__ void part1(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(lds + 4 * l));
  const u16x4 u = __builtin_bit_cast(u16x4, v);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = u[j];
}

// element (k, m) of a [16 k][32 m] tile in [k/4][m/16][4][16] blocks
__host__ __device__ inline int blk(int k, int m) { return ((k >> 2) * 2 + (m >> 4)) * 64 + (k & 3) * 16 + (m & 15); }

__device__ inline bf16x8 frag_tr(const uint16_t* img, int l) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  // lane l: group g = l / 16 -> (m block = g & 1, k half = g >> 1); inside the group lane i passes
  // the address of row i / 4, columns 4 (i % 4).. of the block; first read k = 8 half + 0..3, second + 4..7
  const int i = l & 15, mb = (l >> 4) & 1, kh = l >> 5;
  const uint16_t* p0 = img + blk(8 * kh + (i >> 2), 16 * mb + 4 * (i & 3));
  const uint16_t* p1 = img + blk(8 * kh + 4 + (i >> 2), 16 * mb + 4 * (i & 3));
  const u16x4 a = __builtin_bit_cast(u16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)p0));
  const u16x4 b = __builtin_bit_cast(u16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)p1));
  const u16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ void part2(const float* At, const float* Bt, float* C) {   // At [16 k][32 m], Bt [16 k][32 n]
  __shared__ __attribute__((aligned(16))) uint16_t ia[512], ib[512];
  const int l = threadIdx.x;
  for (int e = l; e < 512; e += 64) {
    const int k = e / 32, m = e % 32;
    ia[blk(k, m)] = (uint16_t)(__float_as_uint(At[e]) >> 16);
    ib[blk(k, m)] = (uint16_t)(__float_as_uint(Bt[e]) >> 16);
  }
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(ia, l), frag_tr(ib, l), acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

int main() {
  uint16_t* d1; uint16_t h1[256];
  hipMalloc(&d1, sizeof h1);
  part1<<<1, 64>>>(d1);
  hipMemcpy(h1, d1, sizeof h1, hipMemcpyDeviceToHost);
  int bad1 = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h1[l * 4 + j] != 64 * (l / 16) + 16 * j + l % 16) ++bad1;
  printf("part 1 (4 x 16 transpose per 16 lanes): %s (%d mismatches)\n", bad1 ? "WRONG" : "CONFIRMED", bad1);
  if (bad1) for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h1[l*4], h1[l*4+1], h1[l*4+2], h1[l*4+3]);
  float hA[512], hB[512], hC[1024];
  srand(2);
  for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 9 - 4); hB[i] = (float)(rand() % 7 - 3); }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  part2<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
    float s = 0; for (int k = 0; k < 16; ++k) s += hA[k * 32 + m] * hB[k * 32 + n];
    if (hC[m * 32 + n] != s) ++bad2; }
  printf("part 2 (K-major tile -> MFMA operand through two transpose reads): %s (%d of 1024 mismatches)\n",
         bad2 ? "WRONG" : "CONFIRMED", bad2);
  return (bad1 || bad2) ? 1 : 0;
}
