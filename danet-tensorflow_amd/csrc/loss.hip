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

#define MAXC 4
#define LOSS_CHUNK 4096

__host__ __device__ static inline int loss_chunks(int64_t N) {
  return (int)((N + LOSS_CHUNK - 1) / LOSS_CHUNK);
}

// per-chunk record: crossL[C*C] | crossS[C*C] | sig  (padded to 2*16+1)
#define REC 33

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
      acc[32] += s[c].x * s[c].x + s[c].y * s[c].y;          // |src|^2 (ops.py:209,213)
    }
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const float mag = (mode == 1) ? hypotf(s[i].x, s[i].y) : 0.f;
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const float dr = s[i].x - ph.x * p[j], di = s[i].y - ph.y * p[j];
        const float cs = dr * dr + di * di;                  // ops.py:415-418
        acc[16 + i * C + j] += cs;
        if (mode == 1) {
          const float d = mag - p[j];                        // ops.py:420-421
          acc[i * C + j] += d * d;
        } else {
          acc[i * C + j] += cs;
        }
      }
    }
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

__device__ __forceinline__ void nth_perm(int C, int p, int* out) {
  int avail[MAXC] = {0, 1, 2, 3};
  int fact = 1;
  for (int i = 2; i < C; ++i) fact *= i;
  int n = C;
  for (int i = 0; i < C; ++i) {
    const int q = p / fact;
    p -= q * fact;
    out[i] = avail[q];
    for (int j = q; j < n - 1; ++j) avail[j] = avail[j + 1];
    --n;
    if (n > 1) fact /= n;
  }
}

// single block: per-utterance permutation search, batch means
__global__ void pit_final_kernel(int B, int C, int64_t N, int nch, float eps,
                                 const float* __restrict__ partial, float* __restrict__ loss,
                                 float* __restrict__ snr, int32_t* __restrict__ perm_idx) {
  __shared__ float red[16];
  int nperm = 1;
  for (int i = 2; i <= C; ++i) nperm *= i;
  float my_loss = 0.f, my_snr = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float rec[REC];
    for (int i = 0; i < REC; ++i) rec[i] = 0.f;
    for (int ch = 0; ch < nch; ++ch)
      for (int i = 0; i < REC; ++i) rec[i] += partial[((int64_t)b * nch + ch) * REC + i];
    const float invN = 1.f / (float)N;                       // reduce_mean over T*F
    int best = 0;
    float best_v = 0.f, best_s = 0.f;
    for (int p = 0; p < nperm; ++p) {
      int perm[MAXC];
      nth_perm(C, p, perm);
      float v = 0.f, sv = 0.f;
      for (int i = 0; i < C; ++i) {                          // ops.py:422-423
        v += rec[i * C + perm[i]] * invN;
        sv += rec[16 + i * C + perm[i]] * invN;
      }
      if (p == 0 || v < best_v) { best = p; best_v = v; best_s = sv; }   // ops.py:424
    }
    perm_idx[b] = best;
    my_loss += best_v;
    const float sig_pwr = rec[32] * invN / (float)C;         // mean over (C,T,F)
    const float noise_pwr = best_s / (float)C;
    my_snr += 4.342944819f * (logf(sig_pwr + eps) - logf(noise_pwr + eps));   // ops.py:221-222
  }
  const float tl = block_sum(my_loss, red);
  const float ts = block_sum(my_snr, red);
  if (threadIdx.x == 0) {
    loss[0] = tl / (float)B;                                 // ops.py:430
    if (snr) snr[0] = ts / (float)B;                         // main.py:308-309
  }
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

extern "C" size_t danet_pit_mse_workspace_bytes(int B, int C, int64_t N) {
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
  if (!ws || ws_bytes < danet_pit_mse_workspace_bytes(B, C, N)) {
    danet_set_error("pit_mse_fwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  const int nch = loss_chunks(N);
  dim3 grid(nch, B);
  DISPATCH_C(C, (pit_cross_kernel<CP><<<grid, 256, 0, stream>>>(
                    mode, N, (const float2*)src_c64, sep_pwr, (const float2*)phasor,
                    (float*)ws)));
  DANET_CHECK_LAUNCH();
  pit_final_kernel<<<1, 256, 0, stream>>>(B, C, N, nch, eps, (const float*)ws, loss, snr,
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
