// Permutation-invariant MSE loss + SNR (gfx950) -- reference
// app/ops.py:374-431 (pit_mse_loss), app/ops.py:191-222 (batch_snr) and their
// call sites main.py:281-337.  HBM-bound: one pass over the complex truth
// [B][C][N], the estimated magnitudes [B][C][N] and the mixture phasor [B][N];
// the C x C cross-error matrix is accumulated in registers, reduced
// wave-shuffle -> LDS -> per-chunk partials, and a one-block finalize kernel
// does the C! permutation search (itertools.permutations order, first index
// on ties), the batch mean and the SNR.  The phase re-attach of main.py:281-284
// is fused: the complex estimate phasor*sep_pwr is never written to HBM here.
#include "common.h"

#include "pit_common.h"
#define LOSS_CHUNK 4096

__host__ __device__ static inline int loss_chunks(int64_t N) {
  return (int)((N + LOSS_CHUNK - 1) / LOSS_CHUNK);
}

template <int CP>
__global__ __launch_bounds__(256) void pit_cross_kernel(
    int mode, int64_t N, const float2* __restrict__ src, const float* __restrict__ sep_pwr,
    const float2* __restrict__ phasor, float* __restrict__ partial) {
  constexpr int C = CP;
  __shared__ float red[4 * REC];
  const int b = blockIdx.y, ch = blockIdx.x, nch = gridDim.x;
  const int64_t n0 = (int64_t)ch * LOSS_CHUNK, n1 = min(N, n0 + LOSS_CHUNK);
  float acc[REC];
#pragma unroll
  for (int i = 0; i < REC; ++i) acc[i] = 0.f;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
    const float2 ph = phasor[(int64_t)b * N + n];
    float2 s[C];
    float p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      s[c] = src[((int64_t)b * C + c) * N + n];
      p[c] = sep_pwr[((int64_t)b * C + c) * N + n];
    }
    pit_accumulate<C>(mode, s, p, ph, acc);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < REC; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) red[wave * REC + i] = v;
  }
  __syncthreads();
  float* out = partial + ((int64_t)b * nch + ch) * REC;
  for (int i = threadIdx.x; i < REC; i += 256)
    out[i] = red[i] + red[REC + i] + red[2 * REC + i] + red[3 * REC + i];
}

template <int CP>
__global__ __launch_bounds__(256) void pit_bwd_kernel(
    int mode, int B, int64_t N, const float2* __restrict__ src,
    const float* __restrict__ sep_pwr, const float2* __restrict__ phasor,
    const int32_t* __restrict__ perm_idx, float dloss, const float* __restrict__ dloss_dev,
    float* __restrict__ dsep) {
  constexpr int C = CP;
  const int b = blockIdx.y;
  int perm[MAXC], inv[MAXC];
  nth_perm(C, perm_idx[b], perm);
  for (int i = 0; i < C; ++i) inv[perm[i]] = i;   // estimate j is paired with truth inv[j]
  // the upstream gradient may live on the device (autograd): no host sync, no extra pass
  const float scale = dloss * (dloss_dev ? *dloss_dev : 1.f) * 2.f / ((float)B * (float)N);
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N;
       n += (int64_t)gridDim.x * 256) {
    const float2 ph = phasor[(int64_t)b * N + n];
    float2 s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = src[((int64_t)b * C + c) * N + n];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      float2 t = s[0];
#pragma unroll
      for (int q = 1; q < C; ++q) t = (inv[j] == q) ? s[q] : t;
      const float p = sep_pwr[((int64_t)b * C + j) * N + n];
      float g;
      if (mode == 1) g = p - hypotf(t.x, t.y);
      else g = p * (ph.x * ph.x + ph.y * ph.y) - (ph.x * t.x + ph.y * t.y);
      dsep[((int64_t)b * C + j) * N + n] = scale * g;
    }
  }
}

size_t dn_ws_pit_mse(int B, int C, int64_t N) {
  (void)C;
  return (size_t)B * loss_chunks(N) * REC * sizeof(float);
}

#define DISPATCH_C(CV, ...)                          \
  switch (CV) {                                      \
    case 1: { constexpr int CP = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int CP = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int CP = 3; __VA_ARGS__; } break; \
    case 4: { constexpr int CP = 4; __VA_ARGS__; } break; \
    default: break;                                  \
  }

extern "C" int danet_pit_mse_fwd(danet_stream_t stream_, int mode, int B, int C, int64_t N,
                                 const float* src_c64, const float* sep_pwr,
                                 const float* phasor, float eps, float* loss, float* snr,
                                 int32_t* perm_idx, void* ws, size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && C <= MAXC && N > 0, "pit_mse_fwd: bad shape");
  DANET_CHECK_ARG(mode == 0 || mode == 1, "pit_mse_fwd: mode");
  DANET_CHECK_ARG(src_c64 && sep_pwr && phasor && loss && perm_idx, "pit_mse_fwd: null pointer");
  if (!ws || ws_bytes < dn_ws_pit_mse(B, C, N)) {
    danet_set_error("pit_mse_fwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  const int nch = loss_chunks(N);
  dim3 grid(nch, B);
  DISPATCH_C(C, (pit_cross_kernel<CP><<<grid, 256, 0, stream>>>(
                    mode, N, (const float2*)src_c64, sep_pwr, (const float2*)phasor,
                    (float*)ws)));
  DANET_CHECK_LAUNCH();
  pit_final_kernel<<<1, PIT_FINAL_THREADS, 0, stream>>>(B, C, N, nch, eps, (const float*)ws, loss, snr,
                                          perm_idx);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_pit_mse_bwd(danet_stream_t stream_, int mode, int B, int C, int64_t N,
                                 const float* src_c64, const float* sep_pwr,
                                 const float* phasor, const int32_t* perm_idx, float dloss,
                                 const float* dloss_dev, float* dsep_pwr) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && C <= MAXC && N > 0, "pit_mse_bwd: bad shape");
  DANET_CHECK_ARG(src_c64 && sep_pwr && phasor && perm_idx && dsep_pwr, "pit_mse_bwd: null pointer");
  dim3 grid((unsigned)min((int64_t)64, cdiv64(N, 256)), B);
  DISPATCH_C(C, (pit_bwd_kernel<CP><<<grid, 256, 0, stream>>>(
                    mode, B, N, (const float2*)src_c64, sep_pwr, (const float2*)phasor,
                    perm_idx, dloss, dloss_dev, dsep_pwr)));
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}
