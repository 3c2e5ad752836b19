// Probe (GPU box): operand layout of v_mfma_f32_32x32x16_bf16 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o tools/csrc/mfma_bf16_layout tools/csrc/mfma_bf16_layout.hip
// Hypothesis: lane l supplies A[row = l % 32][k = 8 * (l / 32) + j] and B[col = l % 32][same k],
// j = 0..7 (element j in bits [16 j, 16 j + 16) of the 128-bit operand); C/D: col = l % 32,
// row = (r & 3) + 8 * (r >> 2) + 4 * (l / 32).  Checks C = A B^T for small-integer matrices.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* C) {   // A [32][16], B [32][16] (B^T form), C [32][32]
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  u32x4 ua, ub;
  for (int w = 0; w < 4; ++w) {
    const uint32_t a0 = __float_as_uint(A[i * 16 + 8 * kb + 2 * w]) >> 16, a1 = __float_as_uint(A[i * 16 + 8 * kb + 2 * w + 1]) >> 16;
    const uint32_t b0 = __float_as_uint(B[i * 16 + 8 * kb + 2 * w]) >> 16, b1 = __float_as_uint(B[i * 16 + 8 * kb + 2 * w + 1]) >> 16;
    ua[w] = a0 | (a1 << 16); ub[w] = b0 | (b1 << 16);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = acc[r];
}

int main() {
  float hA[512], hB[512], hC[1024], ref[1024];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 9 - 4); hB[i] = (float)(rand() % 7 - 3); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[j * 16 + k]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
  printf("mfma_f32_32x32x16_bf16 layout hypothesis: %s (%d of 1024 mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  return bad != 0;
}
