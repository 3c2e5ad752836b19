// MFMA issue-rate microbenchmark (diagnostic): cycles per v_mfma_f32_16x16x4_f32 per SIMD for
// 1 / 2 waves per SIMD, 2..12 independent accumulator chains, with / without LDS operand reads.
// hipcc --offload-arch=gfx950 -O3 -x hip tools/csrc/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ void k16(float* out, int iters, unsigned long long* cyc) {
  __shared__ float sh[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sh[i] = 1e-3f * i;
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-6f, b = 1.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (LDS) { a = sh[(threadIdx.x + it) & 1023]; b = sh[(threadIdx.x * 3 + it) & 1023]; }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// dst != srcC: ping-pong between two accumulator sets
template <int NACC>
__global__ void k16pp(float* out, int iters, unsigned long long* cyc) {
  f32x4 accA[NACC], accB[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { accA[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[i] = accA[i]; }
  float a = threadIdx.x * 1e-6f, b = 1.0f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      accB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accA[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      accA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accB[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += accA[i][0] + accB[i][1];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
static void runpp(int threads, const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k16pp<NACC><<<256, threads>>>(out, iters, cyc);
  hipEventRecord(e0);
  k16pp<NACC><<<256, threads>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)iters * NACC * (threads / 256.0);
  printf("%-28s threads %4d  NACC %2d  dst!=srcC : %.1f ns/MFMA/SIMD (kernel %.3f ms)\n",
         name, threads, NACC, ms * 1e6 / per_simd, ms);
}

template <int NACC, bool LDS>
static void run(int threads, const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k16<NACC, LDS><<<256, threads>>>(out, iters, cyc);
  hipEventRecord(e0);
  k16<NACC, LDS><<<256, threads>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_simd = (double)iters * NACC * (threads / 256.0);   // MFMAs per SIMD
  printf("%-28s threads %4d  NACC %2d  lds %d : %.1f ns/MFMA/SIMD  (%.1f shader-clock ticks / MFMA / SIMD; kernel %.3f ms)\n",
         name, threads, NACC, (int)LDS, ms * 1e6 / per_simd, (double)c / per_simd, ms);
}

int main() {
  run<2, false>(256, "1 wave/SIMD");  run<4, false>(256, "1 wave/SIMD");
  run<8, false>(256, "1 wave/SIMD");  run<12, false>(256, "1 wave/SIMD");
  run<2, false>(512, "2 waves/SIMD"); run<4, false>(512, "2 waves/SIMD");
  run<8, false>(512, "2 waves/SIMD"); run<12, false>(512, "2 waves/SIMD");
  run<12, true>(256, "1 wave/SIMD");  run<12, true>(512, "2 waves/SIMD");
  run<4, true>(512, "2 waves/SIMD");
  runpp<6>(256, "1 wave/SIMD"); runpp<6>(512, "2 waves/SIMD"); runpp<12>(512, "2 waves/SIMD");
  return 0;
}
