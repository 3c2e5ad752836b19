// Shared pieces of the PIT-MSE loss (app/ops.py:374-431, :191-222; main.py:281-337) used by
// loss.hip (stand-alone loss) and attractor.hip (separator + loss fused in one pass).
#pragma once
#include "common.h"

#ifndef MAXC
#define MAXC 4
#endif
// per-chunk record: crossL[C*C] | crossS[C*C] | sig  (padded to 2*16+1)
#define REC 33

// one time-frequency bin's contribution: truth s[c] (complex), estimated magnitudes p[c],
// mixture phasor ph.  acc[i*C+j] = loss cross term (mode 0: complex, ops.py:415-418;
// mode 1: magnitude, ops.py:420-421), acc[16+i*C+j] = complex cross term (SNR), acc[32] = |s|^2
template <int C>
__device__ __forceinline__ void pit_accumulate(int mode, const float2 (&s)[C], const float (&p)[C],
                                               float2 ph, float (&acc)[REC]) {
#pragma unroll
  for (int c = 0; c < C; ++c) acc[32] += s[c].x * s[c].x + s[c].y * s[c].y;   // ops.py:209,213
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

// sum_ch pp[ch * stride] in ascending order, the loads of 16 chunks in flight together: the same
// additions in the same order (hence the same bits) as the plain serial loop, which waited one
// memory latency per chunk (9 chunks at cfg 2: ~7 us at the head of every kernel that derives an
// utterance's permutation from the forward's records).  Lanes past nch re-load the last chunk and
// add +0 (exact: the running sum starts at +0 and can never be -0).
__device__ __forceinline__ float ordered_chunk_sum(const float* __restrict__ pp, int nch, int64_t stride) {
  float s = 0.f;
  for (int c0 = 0; c0 < nch; c0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = pp[(int64_t)min(c0 + u, nch - 1) * stride];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (c0 + u < nch) ? v[u] : 0.f;
  }
  return s;
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

// single block of 1024 threads: per-utterance permutation search, batch means.  The chunk
// partials of an utterance are summed by ONE WAVE (lane i < REC owns record element i and walks
// the chunks in ascending order -- the same order, hence the same bits, as a serial sum), 16
// utterances at a time; the permutation search of those utterances is then one thread each.
// (The serial form -- one thread per utterance, 9 x 33 dependent loads -- took 9.7 us at cfg 2;
// ordered_chunk_sum keeps the loads of a lane in flight together.)
#define PIT_FINAL_THREADS 1024
static __global__ __launch_bounds__(PIT_FINAL_THREADS) void pit_final_kernel(
    int B, int C, int64_t N, int nch, float eps, const float* __restrict__ partial,
    float* __restrict__ loss, float* __restrict__ snr, int32_t* __restrict__ perm_idx) {
  constexpr int NWV = PIT_FINAL_THREADS / 64;
  __shared__ float recs[NWV][REC + 1];
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nperm = 1;
  for (int i = 2; i <= C; ++i) nperm *= i;
  float my_loss = 0.f, my_snr = 0.f;
  for (int b0 = 0; b0 < B; b0 += NWV) {
    const int b = b0 + wave;
    if (b < B && lane < REC) {
      recs[wave][lane] = ordered_chunk_sum(partial + (int64_t)b * nch * REC + lane, nch, REC);
    }
    __syncthreads();
    if (threadIdx.x < NWV && b0 + (int)threadIdx.x < B) {
      const float* rec = recs[threadIdx.x];
      const int bb = b0 + threadIdx.x;
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
      perm_idx[bb] = best;
      my_loss += best_v;
      const float sig_pwr = rec[32] * invN / (float)C;         // mean over (C,T,F)
      const float noise_pwr = best_s / (float)C;
      my_snr += 4.342944819f * (logf(sig_pwr + eps) - logf(noise_pwr + eps));   // ops.py:221-222
    }
    __syncthreads();
  }
  const float tl = block_sum(my_loss, red);
  const float ts = block_sum(my_snr, red);
  if (threadIdx.x == 0) {
    loss[0] = tl / (float)B;                                 // ops.py:430
    if (snr) snr[0] = ts / (float)B;                         // main.py:308-309
  }
}
