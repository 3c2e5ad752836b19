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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The GEMM's inner loop in isolation: a 64x64 wave tile, fragments from LDS (prefetched one
// k-group ahead), SHAPE 0: 2x2 v_mfma_f32_32x32x2_f32 (4 ds_read_b32 per 4 MFMAs),
// SHAPE 1: 4x4 v_mfma_f32_16x16x4_f32 (8 ds_read_b32 per 16 MFMAs).  LDS = 0: operands stay
// in registers.  Nonzero data (zero operands run at a higher clock).
template <int SHAPE, bool LDS>
__global__ void kloop(float* out, int iters) {
  __shared__ float sh[16 * 132 * 2];
  for (int i = threadIdx.x; i < 16 * 132 * 2; i += blockDim.x) sh[i] = 1e-3f * ((i * 2654435761u) >> 20) - 2.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
  const float* as = sh + (wave >> 1) * 64;
  const float* bs = sh + 16 * 132 + (wave & 1) * 64;
  float s = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int f = lane & 31, fk = lane >> 5;
    float a0 = as[fk * 132 + f], a1 = as[fk * 132 + f + 32], b0 = bs[fk * 132 + f], b1 = bs[fk * 132 + f + 32];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float a0n = a0, a1n = a1, b0n = b0, b1n = b1;
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        if (LDS) {
          const int k = ((kk + 1) & 7) * 2 + fk;
          a0n = as[k * 132 + f]; a1n = as[k * 132 + f + 32];
          b0n = bs[k * 132 + f]; b1n = bs[k * 132 + f + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0];
  } else {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int f = lane & 15, fk = lane >> 4;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = as[fk * 132 + f + 16 * i]; b[i] = bs[fk * 132 + f + 16 * i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float an[4], bn[4];
        for (int i = 0; i < 4; ++i) { an[i] = a[i]; bn[i] = b[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[0], acc[i][0], 0, 0, 0);
          if (LDS) {
            const int k = ((kk + 1) & 3) * 4 + fk;
            an[i] = as[k * 132 + f + 16 * i]; bn[i] = bs[k * 132 + f + 16 * i];
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[1], acc[i][1], 0, 0, 0);
          acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[2], acc[i][2], 0, 0, 0);
          acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[3], acc[i][3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
  }
  if (s == 123.456f) out[0] = s;
}

template <int SHAPE, bool LDS>
static void runloop(int threads, const char* name) {
  float* out; hipMalloc(&out, 4);
  const int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kloop<SHAPE, LDS><<<256, threads>>>(out, iters);
  hipEventRecord(e0);
  kloop<SHAPE, LDS><<<256, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // flops per SIMD: iters * 16 k * 64 * 64 * 2 per wave
  const double fl = (double)iters * 16 * 64 * 64 * 2 * (threads / 256.0);
  printf("%-14s %-22s lds %d : %.1f GFLOP/s per SIMD = %.1f TFLOP/s per chip (kernel %.3f ms)\n", name,
         SHAPE == 0 ? "2x2 of 32x32x2" : "4x4 of 16x16x4", (int)LDS, fl / (ms * 1e6), fl / (ms * 1e6) * 1024 / 1e3, ms);
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
  runloop<0, false>(256, "1 wave/SIMD"); runloop<0, true>(256, "1 wave/SIMD");
  runloop<1, false>(256, "1 wave/SIMD"); runloop<1, true>(256, "1 wave/SIMD");
  runloop<0, false>(512, "2 waves/SIMD"); runloop<0, true>(512, "2 waves/SIMD");
  runloop<1, false>(512, "2 waves/SIMD"); runloop<1, true>(512, "2 waves/SIMD");
  runloop<0, true>(768, "3 waves/SIMD"); runloop<1, true>(768, "3 waves/SIMD");
  return 0;
}
