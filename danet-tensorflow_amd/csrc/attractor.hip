// Attractor estimators and dot-product separators of the DANet hot path
// (gfx950) -- reference app/modules.py:382-603 + app/ops.py:273-292.
//
// All of these stream the [B][N][E] embedding (42 MB at B=32,T=128,F=129,E=20)
// once per call and are HBM-bound; one thread owns one time-frequency bin and
// keeps its E-vector in registers (16-B loads when E % 4 == 0), per-utterance
// [C][E] tables live in LDS, and cross-bin sums are reduced
// wave-shuffle -> LDS -> per-chunk partials in `ws` -> a small deterministic
// finalize kernel (no float atomics: results are run-to-run identical).
// The anchor estimator never materialises the reference's [B][P][T][F][C]
// assignment tensor (63 MB x2, app/modules.py:513-516): soft assignments for a
// 128-bin tile are formed in LDS and contracted against the tile immediately.
#include "common.h"
#include "options.h"
#include "pit_common.h"
#include <stdlib.h>

#ifndef CHUNK_N
#define CHUNK_N 2048       // bins per workgroup (reduction granularity; 512 / 1024 measured: the
                           // backward kernels gain 4 us, the chunk sums lose as much)
#endif
#define MAXA 8
// threads per workgroup of the streaming separator / loss / estimator-backward kernels (512 measured
// slower: sep_pit_fwd 13.6 -> 17.4 us, anchor_sep_bwd 40 -> 52 us at cfg 2)
#ifndef SEP_NT
#define SEP_NT 256
#endif
#define SEP_NW (SEP_NT / 64)
#define MAXP 70            // C(8,4)

__host__ __device__ static inline int n_chunks(int64_t N) { return (int)((N + CHUNK_N - 1) / CHUNK_N); }
// the anchor forward kernel does much more work per bin than the others: smaller chunks =
// 4x the workgroups hide its latencies (73 -> 51 us at cfg 2; its finalize 16 -> 22 us)
#ifndef ANCH_CHUNK_N
#define ANCH_CHUNK_N 512
#endif
__host__ __device__ static inline int n_chunks_anchor_fwd(int64_t N) { return (int)((N + ANCH_CHUNK_N - 1) / ANCH_CHUNK_N); }

// a wave-uniform value as a scalar register
__device__ __forceinline__ float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// ---- per-thread embedding row -------------------------------------------------
template <int EP>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int E, float (&x)[EP]) {
  if ((E & 3) == 0) {
#pragma unroll
    for (int q = 0; q < EP / 4; ++q) {
      if (q * 4 < E) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + q * 4);
        x[q * 4 + 0] = v.x; x[q * 4 + 1] = v.y; x[q * 4 + 2] = v.z; x[q * 4 + 3] = v.w;
      } else {
        x[q * 4 + 0] = x[q * 4 + 1] = x[q * 4 + 2] = x[q * 4 + 3] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < EP; ++e) x[e] = (e < E) ? p[e] : 0.f;
  }
}

template <int EP>
__device__ __forceinline__ void store_row(float* __restrict__ p, int E, const float (&x)[EP]) {
  if ((E & 3) == 0) {
#pragma unroll
    for (int q = 0; q < EP / 4; ++q)
      if (q * 4 < E)
        *reinterpret_cast<f32x4*>(p + q * 4) =
            (f32x4){x[q * 4 + 0], x[q * 4 + 1], x[q * 4 + 2], x[q * 4 + 3]};
  } else {
#pragma unroll
    for (int e = 0; e < EP; ++e)
      if (e < E) p[e] = x[e];
  }
}

// block-reduce `cnt` per-thread values (static-indexed array) into dst[cnt]
// (global), using LDS scratch red[NW][CNT].  blockDim.x == 64 * NW; fixed summation order.
template <int CNT, int NW = 4>
__device__ __forceinline__ void block_reduce_store(float (&v)[CNT], int cnt, float* red,
                                                   float* __restrict__ dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sum_n<CNT>(v);                 // (all CNT values: the ones past cnt are never stored)
#pragma unroll
  for (int i = 0; i < CNT; ++i)
    if (i < cnt && lane == 0) red[wave * CNT + i] = v[i];
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += 64 * NW) {
    float t = red[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w * CNT + i];
    dst[i] = t;
  }
  __syncthreads();
}

// tf.argmax over the speaker axis, first index on ties (K9)
__device__ __forceinline__ int argmax_src(const float* __restrict__ src_pwr, int C, int64_t N,
                                          int64_t n) {
  int best = 0;
  float bv = src_pwr[n];
  for (int c = 1; c < C; ++c) {
    const float v = src_pwr[(int64_t)c * N + n];
    if (v > bv) { bv = v; best = c; }
  }
  return best;
}

__device__ __forceinline__ float truth_weight(int mode, float mix) {
  if (mode == 0) return 1.f;                       // app/modules.py:404-406
  if (mode == 1) return (5.f < mix) ? 1.f : 0.f;   // app/modules.py:433-434
  return mix;                                      // app/modules.py:468-469
}

// (utterance, chunk) of this workgroup of a (nch, B) grid, XCD-aware.  Workgroups are dealt to the 8 XCDs
// round-robin by linear id, each XCD has its own L2, and clean lines survive from one kernel to the next.  The
// output projection writes utterance b's embedding from XCD b / (B / 8) (gemm_x6.hip x6_xcd_item: contiguous
// row tiles per XCD), so every head kernel runs utterance b's workgroups on THAT XCD: the estimator finds part
// of the embedding in L2, the separator and the backward kernels find what the kernel before them read
// (measured, cfg 2: the three-launch chain 81.8 -> 67.7 us alone, the train step -18 us;
// profiles/EXPERIMENTS.md round 6).  B not a multiple of 8: the plain mapping.
__device__ __forceinline__ void utterance_chunk(int& b, int& ch) {
  const int nch = gridDim.x, B = gridDim.y;
  b = blockIdx.y; ch = blockIdx.x;
  if ((B & 7) == 0) {
    const int i = blockIdx.x + nch * blockIdx.y, per = B >> 3, x = i & 7, j = i >> 3;
    b = x * per + j % per; ch = j / per;
  }
}

// =========================================================================
// truth family forward (app/modules.py:390-487)
// =========================================================================
template <int EP, bool EF>
__global__ __launch_bounds__(256) void truth_fwd_kernel(
    int mode, int C, int64_t N, int E_, const float* __restrict__ embed,
    const float* __restrict__ src_pwr, const float* __restrict__ mix_pwr,
    float* __restrict__ partial /* [B][chunks][C][EP+1] */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  __shared__ float red[4 * (EP + 1)];
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  const float* sp = src_pwr + (int64_t)b * C * N;
  const float* mp = mix_pwr + (int64_t)b * N;
  float* out = partial + ((int64_t)b * nch + ch) * C * (EP + 1);
  // one speaker at a time keeps the accumulator count at EP+1
  for (int c = 0; c < C; ++c) {
    float acc[EP + 1];
#pragma unroll
    for (int e = 0; e <= EP; ++e) acc[e] = 0.f;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
      if (argmax_src(sp, C, N, n) != c) continue;
      const float w = truth_weight(mode, mp[n]);
      float x[EP];
      load_row<EP>(eb + n * E, E, x);
#pragma unroll
      for (int e = 0; e < EP; ++e) acc[e] += w * x[e];
      acc[EP] += w;
    }
    block_reduce_store<EP + 1>(acc, EP + 1, red, out + c * (EP + 1));
  }
}

// Single pass over the bins with one accumulator set PER SPEAKER (CP * (EP + 1) <= 128 registers):
// the multi-pass kernel above walks the chunk once per speaker (3 x the src_pwr reads and the loop
// overhead at C = 3: 121 us at cfg 4).  A bin adds w * x to its own speaker's set and +0 to the
// others, so every sum sees the same addends in the same order: bit-identical to the multi-pass form.
template <int EP, int CP, bool EF>
__global__ __launch_bounds__(SEP_NT) void truth_fwd1_kernel(
    int mode, int64_t N, int E_, const float* __restrict__ embed,
    const float* __restrict__ src_pwr, const float* __restrict__ mix_pwr,
    float* __restrict__ partial /* [B][chunks][C][EP+1] */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  constexpr int C = CP;
  __shared__ float red[SEP_NW * (EP + 1)];
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  const float* sp = src_pwr + (int64_t)b * C * N;
  const float* mp = mix_pwr + (int64_t)b * N;
  float* out = partial + ((int64_t)b * nch + ch) * C * (EP + 1);
  float acc[CP][EP + 1];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e <= EP; ++e) acc[c][e] = 0.f;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += SEP_NT) {
    const int idx = argmax_src(sp, C, N, n);
    const float w = truth_weight(mode, mp[n]);
    float x[EP];
    load_row<EP>(eb + n * E, E, x);
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float wc = (idx == c) ? w : 0.f;
      if (idx == c) {
#pragma unroll
        for (int e = 0; e < EP; ++e) acc[c][e] += wc * x[e];
        acc[c][EP] += wc;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CP; ++c) block_reduce_store<EP + 1, SEP_NW>(acc[c], EP + 1, red, out + c * (EP + 1));
}

__global__ void truth_final_kernel(int mode, int C, int E, int EP, int nch, float eps,
                                   const float* __restrict__ partial,
                                   float* __restrict__ attr, float* __restrict__ denom) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < C * E; i += blockDim.x) {
    const int c = i / E, e = i % E;
    // (chunk loads in flight together, sums in ascending chunk order: pit_common.h)
    const float* p = partial + ((int64_t)b * nch * C + c) * (EP + 1);
    const float s = ordered_chunk_sum(p + e, nch, (int64_t)C * (EP + 1));
    const float w = ordered_chunk_sum(p + EP, nch, (int64_t)C * (EP + 1));
    const float add = (mode == 0) ? 1.f : eps;     // modules.py:407 vs :447,:482
    attr[((int64_t)b * C + c) * E + e] = s / (w + add);
    if (e == 0) denom[b * C + c] = w;
  }
}

// truth family backward: dembed[n][:] += w(n) * dattr[idx(n)][:] / (denom + add)
template <int EP>
__global__ __launch_bounds__(256) void truth_bwd_kernel(
    int mode, int C, int64_t N, int E, const float* __restrict__ dattr,
    const float* __restrict__ src_pwr, const float* __restrict__ mix_pwr,
    const float* __restrict__ denom, float eps, float* __restrict__ dembed) {
  __shared__ float tab[MAXC * EP];
  int b, ch; utterance_chunk(b, ch);
  for (int i = threadIdx.x; i < C * EP; i += 256) {
    const int c = i / EP, e = i % EP;
    const float add = (mode == 0) ? 1.f : eps;
    tab[i] = (e < E) ? dattr[((int64_t)b * C + c) * E + e] / (denom[b * C + c] + add) : 0.f;
  }
  __syncthreads();
  const float* sp = src_pwr + (int64_t)b * C * N;
  const float* mp = mix_pwr + (int64_t)b * N;
  float* db = dembed + (int64_t)b * N * E;
  for (int64_t n = (int64_t)ch * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    const int c = argmax_src(sp, C, N, n);
    const float w = truth_weight(mode, mp[n]);
    float x[EP];
    load_row<EP>(db + n * E, E, x);
#pragma unroll
    for (int e = 0; e < EP; ++e) x[e] += w * tab[c * EP + e];
    store_row<EP>(db + n * E, E, x);
  }
}

// =========================================================================
// separators (app/modules.py:556-603)
// =========================================================================
template <int EP, bool EF>
__global__ __launch_bounds__(256) void separate_fwd_kernel(
    int act, int C, int64_t N, int E_, const float* __restrict__ mix_pwr,
    const float* __restrict__ attr, const float* __restrict__ embed,
    float* __restrict__ out, float* __restrict__ masks) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  __shared__ float tab[MAXC * EP];
  int b, ch; utterance_chunk(b, ch);
  for (int i = threadIdx.x; i < C * EP; i += 256) {
    const int c = i / EP, e = i % EP;
    tab[i] = (e < E) ? attr[((int64_t)b * C + c) * E + e] : 0.f;
  }
  __syncthreads();
  const float* eb = embed + (int64_t)b * N * E;
  for (int64_t n = (int64_t)ch * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    float x[EP];
    load_row<EP>(eb + n * E, E, x);
    float lg[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      float s = 0.f;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < EP; ++e) s += x[e] * tab[c * EP + e];   // modules.py:558-560
      }
      lg[c] = s;
    }
    if (act == 0) {                                   // softmax, modules.py:595
      float mx = lg[0];
#pragma unroll
      for (int c = 1; c < MAXC; ++c) if (c < C) mx = fmaxf(mx, lg[c]);
      float den = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) if (c < C) { lg[c] = expf(lg[c] - mx); den += lg[c]; }
#pragma unroll
      for (int c = 0; c < MAXC; ++c) if (c < C) lg[c] = lg[c] / den;
    } else {                                          // sigmoid, modules.py:566
#pragma unroll
      for (int c = 0; c < MAXC; ++c) if (c < C) lg[c] = sigmoid_acc(lg[c]);
    }
    const float mp = mix_pwr[(int64_t)b * N + n];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        out[((int64_t)b * C + c) * N + n] = mp * lg[c];   // modules.py:567-574
        if (masks) masks[((int64_t)b * N + n) * C + c] = lg[c];
      }
  }
}

template <int EP, int CP>
__global__ __launch_bounds__(256) void separate_bwd_kernel(
    int act, int C_, int64_t N, int E, const float* __restrict__ mix_pwr,
    const float* __restrict__ attr, const float* __restrict__ embed,
    const float* __restrict__ dout, float* __restrict__ dembed,
    float* __restrict__ partial /* [B][chunks][C][EP] */) {
  constexpr int C = CP;
  (void)C_;
  __shared__ float tab[CP * EP];
  __shared__ float red[4 * EP];
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  for (int i = threadIdx.x; i < C * EP; i += 256) {
    const int c = i / EP, e = i % EP;
    tab[i] = (e < E) ? attr[((int64_t)b * C + c) * E + e] : 0.f;
  }
  __syncthreads();
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  float* db = dembed + (int64_t)b * N * E;
  // the attractor table is uniform: in scalar registers it costs no VGPRs and no LDS reads
  // (hoisted LDS broadcasts had pushed the kernel to 176 VGPRs = 2 waves per SIMD, and with 8
  // dependent load -> compute -> store rounds per thread the kernel is latency-bound)
  float st[CP][EP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e < EP; ++e) st[c][e] = uniform(tab[c * EP + e]);
  float accs[CP][EP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e < EP; ++e) accs[c][e] = 0.f;

  for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
    float x[EP];
    load_row<EP>(eb + n * E, E, x);
    float m[CP], dl[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      float s = 0.f;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < EP; ++e) s += x[e] * st[c][e];
      }
      m[c] = s;
    }
    const float mp = mix_pwr[(int64_t)b * N + n];
    if (act == 0) {
      float mx = m[0];
#pragma unroll
      for (int c = 1; c < CP; ++c) if (c < C) mx = fmaxf(mx, m[c]);
      float den = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) if (c < C) { m[c] = expf(m[c] - mx); den += m[c]; }
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c)
        if (c < C) {
          m[c] = m[c] / den;
          dl[c] = dout[((int64_t)b * C + c) * N + n] * mp;   // dL/dmask
          dot += m[c] * dl[c];
        }
#pragma unroll
      for (int c = 0; c < CP; ++c) if (c < C) dl[c] = m[c] * (dl[c] - dot);
    } else {
#pragma unroll
      for (int c = 0; c < CP; ++c)
        if (c < C) {
          m[c] = sigmoid_acc(m[c]);
          dl[c] = dout[((int64_t)b * C + c) * N + n] * mp * m[c] * (1.f - m[c]);
        }
    }
    float dx[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) dx[e] = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
      if (c < C) {
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          dx[e] += dl[c] * st[c][e];
          accs[c][e] += dl[c] * x[e];
        }
      }
    store_row<EP>(db + n * E, E, dx);
  }
  float* out = partial + ((int64_t)b * nch + ch) * C * EP;
#pragma unroll
  for (int c = 0; c < CP; ++c)
    if (c < C) block_reduce_store<EP>(accs[c], EP, red, out + c * EP);
}

// sum_ch pp[ch * stride]: 4 independent partial sums -- the chunk loads are in flight together instead of
// one dependent load per add (fixed combination order: deterministic)
__device__ __forceinline__ float chunk_sum4(const float* __restrict__ pp, int nch, int64_t stride) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int ch = 0;
  for (; ch + 4 <= nch; ch += 4)
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += pp[(int64_t)(ch + u) * stride];
  for (; ch < nch; ++ch) s[0] += pp[(int64_t)ch * stride];
  return (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ void sum_chunks_kernel(int nch, int C, int E, int EP,
                                  const float* __restrict__ partial, float* __restrict__ out) {
  // out[b][c][e] = sum_ch partial[b][ch][c][e(EP)]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < C * E; i += blockDim.x) {
    const int c = i / E, e = i % E;
    out[((int64_t)b * C + c) * E + e] = chunk_sum4(partial + ((int64_t)b * nch * C + c) * EP + e, nch, (int64_t)C * EP);
  }
}

// =========================================================================
// separator + PIT loss in one pass (training): app/modules.py:548-603 -> main.py:281-290,
// 308-309 -> app/ops.py:374-431.  The masks, the separated magnitudes and (in backward) the
// loss gradient w.r.t. them exist only in registers: logits -> softmax / sigmoid -> x |mix| ->
// cross-error partials (forward); the same chain recomputed + dL/dsep -> dL/dlogit -> dembed
// and dattr partials (backward).  `out` (separated magnitudes) is written only when asked for.
// =========================================================================
template <int EP, int CP>
__device__ __forceinline__ void sep_masks(int act, const float (&x)[EP], const float (&st)[CP][EP],
                                          float (&m)[CP]) {
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EP; ++e) s += x[e] * st[c][e];          // modules.py:558-560
    m[c] = s;
  }
  if (act == 0) {                                               // softmax, modules.py:595
    float mx = m[0];
#pragma unroll
    for (int c = 1; c < CP; ++c) mx = fmaxf(mx, m[c]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { m[c] = expf(m[c] - mx); den += m[c]; }
#pragma unroll
    for (int c = 0; c < CP; ++c) m[c] = m[c] / den;
  } else {                                                      // sigmoid, modules.py:566
#pragma unroll
    for (int c = 0; c < CP; ++c) m[c] = sigmoid_acc(m[c]);
  }
}

// the same with the attractor table read from LDS (broadcast reads): in the forward kernel the
// scalar-register copy spills (CP*EP > ~40 SGPRs) and costs more than it saves
template <int EP, int CP>
__device__ __forceinline__ void sep_masks_lds(int act, const float (&x)[EP], const float* tab,
                                              float (&m)[CP]) {
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EP; ++e) s += x[e] * tab[c * EP + e];
    m[c] = s;
  }
  if (act == 0) {
    float mx = m[0];
#pragma unroll
    for (int c = 1; c < CP; ++c) mx = fmaxf(mx, m[c]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { m[c] = expf(m[c] - mx); den += m[c]; }
#pragma unroll
    for (int c = 0; c < CP; ++c) m[c] = m[c] / den;
  } else {
#pragma unroll
    for (int c = 0; c < CP; ++c) m[c] = sigmoid_acc(m[c]);
  }
}


// GRAD (round 6; CP == 2, the two-speaker configurations): the same pass ALSO accumulates the attractor-gradient partials of the fused
// separator + loss backward -- sep_pit_bwd_kernel's accs[c][e] += dL/dlogit_c x_e -- for EVERY permutation
// (C! = 2 sets of C x EP sums per thread), with the upstream gradient left out (scale = 2 / (B N); the
// consumer multiplies by dloss).  Which permutation an utterance takes is only known once all of its
// chunks' cross errors are summed; the estimator's backward (anchor_sep_bwd_kernel / truth_sep_bwd_kernel)
// derives it from the records as before and then adds up the partials of THAT permutation itself -- so on
// the training path the launches danet_separate_pit_bwd (one more read of the 42 MB embedding, 21.5 us at
// cfg 2) and sum_chunks (5.6 us) do not exist.  Same products and sums in the same order as the backward
// kernel's (equal to an ulp per chunk partial: the compiler contracts the shared expressions differently).
template <int CP>
__device__ __forceinline__ void sep_pit_dlogit(int act, int mode, const float (&m)[CP], float mp,
                                               float2 ph, const float2 (&s)[CP],
                                               const int (&inv)[MAXC], float scale, float (&dl)[CP]);
template <int CP> struct NPerm { static constexpr int value = CP * NPerm<CP - 1>::value; };
template <> struct NPerm<1> { static constexpr int value = 1; };
#define SEP_GRAD_MAXC 2

template <int EP, int CP, bool EF, bool GRAD>
__global__ __launch_bounds__(SEP_NT) void sep_pit_fwd_kernel(
    int act, int mode, int B, int64_t N, int E_, const float* __restrict__ mix_pwr,
    const float* __restrict__ attr, const float* __restrict__ embed,
    const float2* __restrict__ src, const float2* __restrict__ phasor,
    float* __restrict__ out /* optional [B][C][N] */, float* __restrict__ partial /* [B][nch][REC] */,
    float* __restrict__ gpartial /* GRAD: [B][nch][C!][C][EP] */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  constexpr int C = CP;
  constexpr int NP = GRAD ? NPerm<CP>::value : 1;
  __shared__ float tab[CP * EP];
  __shared__ float red[SEP_NW * REC];
  __shared__ float gred[GRAD ? SEP_NW * EP : 1];
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  for (int i = threadIdx.x; i < C * EP; i += SEP_NT) {
    const int c = i / EP, e = i % EP;
    tab[i] = (e < E) ? attr[((int64_t)b * C + c) * E + e] : 0.f;
  }
  __syncthreads();
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  float acc[REC];
#pragma unroll
  for (int i = 0; i < REC; ++i) acc[i] = 0.f;
  float gacc[NP][CP][GRAD ? EP : 1];
  int inv[NP][MAXC];
  if constexpr (GRAD) {
#pragma unroll
    for (int pq = 0; pq < NP; ++pq) {
      int perm[MAXC];
      nth_perm(C, pq, perm);
      for (int i = 0; i < C; ++i) inv[pq][perm[i]] = i;          // estimate j is paired with truth inv[j]
#pragma unroll
      for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int e = 0; e < EP; ++e) gacc[pq][c][e] = 0.f;
    }
  }
  const float gscale = 2.f / ((float)B * (float)N);              // (sep_pit_bwd_kernel's scale at dloss = 1)
  for (int64_t n = n0 + threadIdx.x; n < n1; n += SEP_NT) {
    float x[EP];
    load_row<EP>(eb + n * E, E, x);
    const float mp = mix_pwr[(int64_t)b * N + n];
    const float2 ph = phasor[(int64_t)b * N + n];
    float2 s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = src[((int64_t)b * C + c) * N + n];
    float m[CP], p[CP];
    sep_masks_lds<EP, CP>(act, x, tab, m);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      p[c] = mp * m[c];                                          // modules.py:567-574
      if (out) out[((int64_t)b * C + c) * N + n] = p[c];
    }
    pit_accumulate<C>(mode, s, p, ph, acc);
    if constexpr (GRAD) {
#pragma unroll
      for (int pq = 0; pq < NP; ++pq) {
        float dl[CP];
        sep_pit_dlogit<CP>(act, mode, m, mp, ph, s, inv[pq], gscale, dl);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int e = 0; e < EP; ++e) gacc[pq][c][e] += dl[c] * x[e];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sum_n<REC>(acc);
#pragma unroll
  for (int i = 0; i < REC; ++i)
    if (lane == 0) red[wave * REC + i] = acc[i];
  __syncthreads();
  float* po = partial + ((int64_t)b * nch + ch) * REC;
  for (int i = threadIdx.x; i < REC; i += SEP_NT) {
    float t = red[i];
#pragma unroll
    for (int w = 1; w < SEP_NW; ++w) t += red[w * REC + i];
    po[i] = t;
  }
  if constexpr (GRAD) {
    __syncthreads();
    float* go = gpartial + ((int64_t)b * nch + ch) * NP * C * EP;
#pragma unroll
    for (int pq = 0; pq < NP; ++pq)
#pragma unroll
      for (int c = 0; c < CP; ++c) block_reduce_store<EP, SEP_NW>(gacc[pq][c], EP, gred, go + (pq * C + c) * EP);
  }
}

// dattr[c][e] of utterance b into LDS from the forward's gradient partials of permutation `perm`
// (sum_chunks_kernel's sum, times the upstream gradient); the caller's next __syncthreads() publishes it
template <int EP, int CP>
__device__ __forceinline__ void dattr_from_partials(const float* __restrict__ gpartial, int b, int nch,
                                                    int perm, float dscale, float* Dr) {
  constexpr int NP = NPerm<CP>::value;
  for (int i = threadIdx.x; i < CP * EP; i += blockDim.x)
    Dr[i] = dscale * chunk_sum4(gpartial + (((int64_t)b * nch) * NP + perm) * CP * EP + i, nch, (int64_t)NP * CP * EP);
}

// dL/dlogit_c of the fused separator + PIT loss for one bin (masks m, mixture magnitude mp,
// phasor ph, truth s, inverse permutation inv, scale = dloss * 2 / (B N)); also used by the
// estimator backward that recomputes the separator's embedding gradient
template <int CP>
__device__ __forceinline__ void sep_pit_dlogit(int act, int mode, const float (&m)[CP], float mp,
                                               float2 ph, const float2 (&s)[CP],
                                               const int (&inv)[MAXC], float scale, float (&dl)[CP]) {
  constexpr int C = CP;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    float2 t = s[0];
#pragma unroll
    for (int q = 1; q < C; ++q) t = (inv[j] == q) ? s[q] : t;
    const float p = mp * m[j];
    float g;
    if (mode == 1) g = p - hypotf(t.x, t.y);
    else g = p * (ph.x * ph.x + ph.y * ph.y) - (ph.x * t.x + ph.y * t.y);
    dl[j] = (scale * g) * mp;                                  // dL/dmask
  }
  if (act == 0) {
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) dot += m[c] * dl[c];
#pragma unroll
    for (int c = 0; c < C; ++c) dl[c] = m[c] * (dl[c] - dot);
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) dl[c] = dl[c] * m[c] * (1.f - m[c]);
  }
}

// this utterance's permutation from the forward's chunk records (the sums and the search of
// pit_final_kernel) or from perm_idx; published by the caller's next __syncthreads()
template <int CP>
__device__ __forceinline__ void sep_pit_perm(const float* __restrict__ records,
                                             const int32_t* __restrict__ perm_idx, int b, int nch,
                                             int64_t N, float* rec_s, int* perm_s) {
  constexpr int C = CP;
  if (records != nullptr) {
    if (threadIdx.x < REC)
      rec_s[threadIdx.x] = ordered_chunk_sum(records + (int64_t)b * nch * REC + threadIdx.x, nch, REC);
    __syncthreads();
    if (threadIdx.x == 0) {
      int nperm = 1;
      for (int i = 2; i <= C; ++i) nperm *= i;
      const float invN = 1.f / (float)N;
      int best = 0;
      float best_v = 0.f;
      for (int pq = 0; pq < nperm; ++pq) {
        int pm[MAXC];
        nth_perm(C, pq, pm);
        float v = 0.f;
        for (int i = 0; i < C; ++i) v += rec_s[i * C + pm[i]] * invN;
        if (pq == 0 || v < best_v) { best = pq; best_v = v; }
      }
      *perm_s = best;
    }
  } else if (threadIdx.x == 0) {
    *perm_s = perm_idx[b];
  }
}

template <int EP, int CP, bool EF>
__global__ __launch_bounds__(SEP_NT) void sep_pit_bwd_kernel(
    int act, int mode, int B, int64_t N, int E_, const float* __restrict__ mix_pwr,
    const float* __restrict__ attr, const float* __restrict__ embed,
    const float2* __restrict__ src, const float2* __restrict__ phasor,
    const int32_t* __restrict__ perm_idx, const float* __restrict__ records,
    float dloss, const float* __restrict__ dloss_dev,
    float* __restrict__ dembed, float* __restrict__ partial /* [B][nch][C][EP] */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  constexpr int C = CP;
  __shared__ float tab[CP * EP];
  __shared__ float red[SEP_NW * EP];
  __shared__ float rec_s[REC + 1];
  __shared__ int perm_s;
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  sep_pit_perm<CP>(records, perm_idx, b, nch, N, rec_s, &perm_s);
  for (int i = threadIdx.x; i < C * EP; i += SEP_NT) {
    const int c = i / EP, e = i % EP;
    tab[i] = (e < E) ? attr[((int64_t)b * C + c) * E + e] : 0.f;
  }
  __syncthreads();
  // (at C = 3, E = 40 the 120 values overflow the scalar register file and the compiler parks the excess in
  // vector-register lanes -- 332 v_readlane in this kernel; reading the table from LDS instead was measured in
  // round 6: cfg 4 3.45 ms either way)
  float st[CP][EP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e < EP; ++e) st[c][e] = uniform(tab[c * EP + e]);
  int perm[MAXC], inv[MAXC];
  nth_perm(C, perm_s, perm);       // (published by the barrier behind the table load above)
  for (int i = 0; i < C; ++i) inv[perm[i]] = i;   // estimate j is paired with truth inv[j]
  const float scale = dloss * (dloss_dev ? *dloss_dev : 1.f) * 2.f / ((float)B * (float)N);
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  float* db = dembed + (int64_t)b * N * E;
  float accs[CP][EP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e < EP; ++e) accs[c][e] = 0.f;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += SEP_NT) {
    float x[EP];
    load_row<EP>(eb + n * E, E, x);
    const float mp = mix_pwr[(int64_t)b * N + n];
    const float2 ph = phasor[(int64_t)b * N + n];
    float2 s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = src[((int64_t)b * C + c) * N + n];
    float m[CP], dl[CP];
    sep_masks<EP, CP>(act, x, st, m);
    // dL/dsep (ops.py:412-430 differentiated; the estimate j is compared with truth inv[j])
    sep_pit_dlogit<CP>(act, mode, m, mp, ph, s, inv, scale, dl);
    if (dembed != nullptr) {
      float dx[EP];
#pragma unroll
      for (int e = 0; e < EP; ++e) dx[e] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          dx[e] += dl[c] * st[c][e];
          accs[c][e] += dl[c] * x[e];
        }
      }
      store_row<EP>(db + n * E, E, dx);
    } else {          // dattr partials only: the estimator's backward recomputes the dx term
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < EP; ++e) accs[c][e] += dl[c] * x[e];
    }
  }
  float* po = partial + ((int64_t)b * nch + ch) * C * EP;
#pragma unroll
  for (int c = 0; c < CP; ++c) block_reduce_store<EP, SEP_NW>(accs[c], EP, red, po + c * EP);
}

// The truth-family estimator backward WITH the fused separator + loss backward's dembed term
// recomputed in the same pass (see anchor_sep_bwd_kernel): dembed = dL/dembed|separator +
// w(n) * dattr[idx(n)] / (denom + add), written once.
template <int EP, int CP, bool EF>
__global__ __launch_bounds__(SEP_NT) void truth_sep_bwd_kernel(
    int tmode, int64_t N, int E_, const float* __restrict__ dattr,
    const float* __restrict__ src_pwr, const float* __restrict__ mix_pwr,
    const float* __restrict__ denom, float eps,
    const float* __restrict__ embed, const float* __restrict__ attr,
    int act, int mode, int B, const float2* __restrict__ src, const float2* __restrict__ phasor,
    const int32_t* __restrict__ perm_idx, const float* __restrict__ records,
    float dloss, const float* __restrict__ dloss_dev, float* __restrict__ dembed,
    const float* __restrict__ gpartial /* or null: dattr = the forward's partials of the utterance's permutation */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  constexpr int C = CP;
  __shared__ float dtab[CP * EP];   // dattr / (denom + add)
  __shared__ float tab[CP * EP];    // attractors (the separator's table)
  __shared__ float rec_s[REC + 1];
  __shared__ int perm_s;
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  sep_pit_perm<CP>(records, perm_idx, b, nch, N, rec_s, &perm_s);
  const float dscale = dloss * (dloss_dev ? *dloss_dev : 1.f);
  if (gpartial != nullptr) {
    if constexpr (CP == SEP_GRAD_MAXC) {
      __syncthreads();                // perm_s
      dattr_from_partials<EP, CP>(gpartial, b, nch, perm_s, dscale, dtab);
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < C * EP; i += SEP_NT) {
    const int c = i / EP, e = i % EP;
    const float add = (tmode == 0) ? 1.f : eps;
    const float d = gpartial ? dtab[i] : ((e < E) ? dattr[((int64_t)b * C + c) * E + e] : 0.f);
    dtab[i] = (e < E) ? d / (denom[b * C + c] + add) : 0.f;
    tab[i] = (e < E) ? attr[((int64_t)b * C + c) * E + e] : 0.f;
  }
  __syncthreads();
  int perm[MAXC], inv[MAXC];
  nth_perm(C, perm_s, perm);
  for (int i = 0; i < C; ++i) inv[perm[i]] = i;
  const float scale = dscale * 2.f / ((float)B * (float)N);
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  const float* sp = src_pwr + (int64_t)b * C * N;
  float* db = dembed + (int64_t)b * N * E;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += SEP_NT) {
    float x[EP], dx[EP];
    load_row<EP>(eb + n * E, E, x);
    const float mp = mix_pwr[(int64_t)b * N + n];
    const float2 ph = phasor[(int64_t)b * N + n];
    float2 sv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sv[c] = src[((int64_t)b * C + c) * N + n];
    float m[CP], dl[CP];
    sep_masks_lds<EP, CP>(act, x, tab, m);
    sep_pit_dlogit<CP>(act, mode, m, mp, ph, sv, inv, scale, dl);
#pragma unroll
    for (int e = 0; e < EP; ++e) dx[e] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int e = 0; e < EP; ++e) dx[e] += dl[c] * tab[c * EP + e];
    const int cs = argmax_src(sp, C, N, n);
    const float w = truth_weight(tmode, mp);
#pragma unroll
    for (int e = 0; e < EP; ++e) dx[e] += w * dtab[cs * EP + e];
    store_row<EP>(db + n * E, E, dx);
  }
}

// =========================================================================
// anchor estimator (app/modules.py:501-545)
// =========================================================================
// threads per workgroup of the anchor forward kernel (a multiple of 64, >= ANCH_TN; 128 threads with
// 128-bin tiles measured 41.9 us against 38.9 us for 256 at cfg 2)
#ifndef ANCH_NT
#define ANCH_NT 256
#endif
#define ANCH_NW (ANCH_NT / 64)
#define ANCH_ITEMS (1024 / ANCH_NT)
#ifndef ANCH_TN
#define ANCH_TN 128   // bins per LDS tile (one per thread of the first two waves in phase 1; a multiple
                      // of 8, <= 256).  128: 28 KB of LDS, five workgroups per CU overlap each other's load /
                      // assignment / MFMA phases (256: 57 KB, two per CU; cfg 2 42.4 -> 38.6 us, E = 40 C = 3
                      // 140 -> 109 us; fetching the next tile's rows ahead made both slower)
#endif

struct AnchorCombos { int P; int idx[MAXP][MAXC]; };

static void make_combos(int A, int C, AnchorCombos& cb) {
  // itertools.combinations(range(A), C): lexicographic (app/ops.py:287-292)
  int cur[MAXC];
  for (int i = 0; i < C; ++i) cur[i] = i;
  cb.P = 0;
  for (;;) {
    for (int i = 0; i < C; ++i) cb.idx[cb.P][i] = cur[i];
    ++cb.P;
    int i = C - 1;
    while (i >= 0 && cur[i] == A - C + i) --i;
    if (i < 0) break;
    ++cur[i];
    for (int j = i + 1; j < C; ++j) cur[j] = cur[j - 1] + 1;
  }
}

// itertools.combinations(range(A), C) at compile time: with (A, C) known the
// subset loop indexes the exponential table with constants instead of two
// 7-deep select chains per member (it was ~3/4 of the kernel's instructions).
template <int A, int C>
struct CombosCE {
  int P;
  int idx[MAXP][MAXC];
  constexpr CombosCE() : P(0), idx() {
    int cur[MAXC] = {0, 1, 2, 3};
    bool more = true;
    while (more) {
      for (int i = 0; i < C; ++i) idx[P][i] = cur[i];
      ++P;
      int i = C - 1;
      while (i >= 0 && cur[i] == A - C + i) --i;
      if (i < 0) { more = false; }
      else {
        ++cur[i];
        for (int j = i + 1; j < C; ++j) cur[j] = cur[j - 1] + 1;
      }
    }
  }
};

// LDS: Xs[ANCH_TN][EPA] (embedding tile + ones column), Ss[ANCH_TN][PC+1]
// AT, CTT > 0: (A, C) specialisation; AT = 0: generic (run-time table in `cb`)
// T11: the contraction is ONE 32 x 32 MFMA tile (PC <= 32, EPA <= 32: cfg 2) known at compile time --
// one accumulator tile instead of four behind run-time conditions
template <int EP, int AT, int CTT, bool T11, bool EF>
__global__ __launch_bounds__(ANCH_NT) void anchor_fwd_kernel(
    int C, int64_t N, int E_, int A, AnchorCombos cb, const float* __restrict__ embed,
    const float* __restrict__ anchors, float* __restrict__ partial /* [B][chunks][PC][EPA] */,
    int RT_, int CT_ /* MFMA tiling of the [PC][EPA] contraction; RT = 0: scalar path */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  constexpr int NM = T11 ? 1 : 2;
  const int RT = T11 ? 1 : RT_, CT = T11 ? 1 : CT_;
  constexpr int EPA = EP + 4;           // + ones column, padded to a float4
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (A, C) specialisation: subset count, row stride and loop bounds are compile-time constants
  constexpr int AMAX = AT > 0 ? AT : MAXA;
  constexpr int PC_CE = AT > 0 ? CombosCE<(AT > 0 ? AT : 2), (AT > 0 ? CTT : 1)>().P * CTT : 0;
  const int PC = AT > 0 ? PC_CE : cb.P * C;
  const int lds = PC + 1;               // odd-ish stride for the assignment tile
  float* Xs = smem;                     // [ANCH_TN][EPA]
  float* Ss = Xs + ANCH_TN * EPA;       // [ANCH_TN][lds]
  float* An = Ss + ANCH_TN * lds;       // [A][EP]
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < A * EP; i += ANCH_NT) {
    const int a = i / EP, e = i % EP;
    An[i] = (e < E) ? anchors[a * E + e] : 0.f;
  }
  const int64_t n0 = (int64_t)ch * ANCH_CHUNK_N, n1 = min(N, n0 + ANCH_CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;

  // work items: (pc, quad) -> 4 accumulators; up to ANCH_ITEMS items per thread
  constexpr int EQ = EPA / 4;
  const int nitems = PC * EQ;
  f32x4 acc[ANCH_ITEMS];
#pragma unroll
  for (int i = 0; i < ANCH_ITEMS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // MFMA path: out[pc][e] = sum_bins Ss[bin][pc] * Xs[bin][e] as RT x CT tiles of
  // v_mfma_f32_32x32x2_f32, the tile's bins split over the waves.  Rows /
  // columns past PC / EPA read whatever follows in LDS: they only ever reach
  // accumulator entries that are never stored.
  f32x16 macc[NM][NM];
#pragma unroll
  for (int i = 0; i < NM; ++i)
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) macc[i][j][r] = 0.f;
  const int mlane = tid & 63, mwave = tid >> 6;
  __syncthreads();

  for (int64_t base = n0; base < n1; base += ANCH_TN) {
    // phase 1: one thread per bin builds its row and its soft assignments
    if (tid < ANCH_TN) {
      const int64_t n = base + tid;
      float x[EP];
      float* xr = Xs + tid * EPA;
      if (n < n1) {
        load_row<EP>(eb + n * E, E, x);
#pragma unroll
        for (int e = 0; e < EP; ++e) xr[e] = x[e];
        xr[EP] = 1.f; xr[EP + 1] = 0.f; xr[EP + 2] = 0.f; xr[EP + 3] = 0.f;
        float d[MAXA];
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
          float s = 0.f;
          if (a < AMAX && (AT > 0 || a < A)) {
#pragma unroll
            for (int e = 0; e < EP; ++e) s += x[e] * An[a * EP + e];   // modules.py:513-515
          }
          d[a] = s;
        }
        // softmax over a subset is invariant to a common shift, so the A
        // exponentials are taken once against the global max instead of C per
        // subset (P*C = 30 -> A = 6 expf per bin at A=6, C=2)
        float dmax = d[0];
#pragma unroll
        for (int a = 1; a < MAXA; ++a) if (a < AMAX && (AT > 0 || a < A)) dmax = fmaxf(dmax, d[a]);
        float ea[MAXA];
#pragma unroll
        // (hardware exponential: arguments are <= 0 after the max subtraction; ~1e-7 relative in a
        // soft assignment, far inside the 1e-4 bar, and 6 x ~20 fewer vector instructions per bin)
        for (int a = 0; a < MAXA; ++a) ea[a] = (a < AMAX && (AT > 0 || a < A)) ? __expf(d[a] - dmax) : 0.f;
        if constexpr (AT > 0) {
          constexpr CombosCE<AT, CTT> tb{};
          float* srow = Ss + tid * lds;
          // branch-free over the subsets; ONE (rare) branch afterwards redoes the subsets all of
          // whose members underflowed against the global max (15 per-subset branches cost more
          // exec-mask bookkeeping than the arithmetic they guard)
          bool under = false;
#pragma unroll
          for (int p = 0; p < tb.P; ++p) {
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < CTT; ++c) den += ea[tb.idx[p][c]];
            under |= (den < 1e-30f);
            const float inv = __builtin_amdgcn_rcpf(den);          // v_rcp_f32 (__frcp_rn is the 11-instruction IEEE division)
#pragma unroll
            for (int c = 0; c < CTT; ++c) srow[p * CTT + c] = ea[tb.idx[p][c]] * inv;   // modules.py:516
          }
          if (under) {
#pragma unroll
            for (int p = 0; p < tb.P; ++p) {
              float den = 0.f;
#pragma unroll
              for (int c = 0; c < CTT; ++c) den += ea[tb.idx[p][c]];
              if (den < 1e-30f) {      // redo against the subset's own max (tf.nn.softmax's formulation)
                float lg[CTT];
                float mx = -INFINITY;
#pragma unroll
                for (int c = 0; c < CTT; ++c) mx = fmaxf(mx, d[tb.idx[p][c]]);
                den = 0.f;
#pragma unroll
                for (int c = 0; c < CTT; ++c) { lg[c] = expf(d[tb.idx[p][c]] - mx); den += lg[c]; }
#pragma unroll
                for (int c = 0; c < CTT; ++c) srow[p * CTT + c] = lg[c] / den;
              }
            }
          }
        } else
        for (int p = 0; p < cb.P; ++p) {
          float lg[MAXC], dl[MAXC];
          float den = 0.f;
#pragma unroll
          for (int c = 0; c < MAXC; ++c)
            if (c < C) {
              const int a = cb.idx[p][c];
              float v = ea[0], w = d[0];
#pragma unroll
              for (int q = 1; q < MAXA; ++q) { v = (a == q) ? ea[q] : v; w = (a == q) ? d[q] : w; }
              lg[c] = v; dl[c] = w;
              den += v;
            }
          if (den < 1e-30f) {
            // every member underflowed against the global max: redo this subset
            // against its own max (exactly tf.nn.softmax's formulation)
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) mx = fmaxf(mx, dl[c]);
            den = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) { lg[c] = expf(dl[c] - mx); den += lg[c]; }
          }
#pragma unroll
          for (int c = 0; c < MAXC; ++c)
            if (c < C) Ss[tid * lds + p * C + c] = lg[c] / den;        // modules.py:516
        }
      } else {
        for (int e = 0; e < EPA; ++e) xr[e] = 0.f;
        for (int i = 0; i < PC; ++i) Ss[tid * lds + i] = 0.f;
      }
    }
    __syncthreads();
    // phase 2: contract assignments against [x | 1]  (modules.py:519-523)
    if (RT > 0) {
      const int il = mlane & 31, kl = mlane >> 5;
#pragma unroll 4
      for (int ks = 0; ks < ANCH_TN / (2 * ANCH_NW); ++ks) {
        const int r = mwave * (ANCH_TN / ANCH_NW) + ks * 2 + kl;
        float av[NM], bv[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          av[i] = (i < RT) ? Ss[r * lds + i * 32 + il] : 0.f;
          bv[i] = (i < CT) ? Xs[r * EPA + i * 32 + il] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
          for (int j = 0; j < NM; ++j)
            if (i < RT && j < CT)
              macc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], macc[i][j], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int it = 0; it < ANCH_ITEMS; ++it) {
      const int item = tid + it * ANCH_NT;
      if (item < nitems) {
        const int pc = item / EQ, q = item % EQ;
        f32x4 a4 = acc[it];
        for (int r = 0; r < ANCH_TN; ++r) {
          const float sv = Ss[r * lds + pc];
          const f32x4 xv = *reinterpret_cast<const f32x4*>(&Xs[r * EPA + q * 4]);
          a4 += sv * xv;
        }
        acc[it] = a4;
      }
    }
    __syncthreads();
  }
  float* out = partial + ((int64_t)b * nch + ch) * PC * EPA;
  if (RT > 0) {
    // cross-wave reduction through LDS (tile buffers are dead now).
    // D layout 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* red = smem;   // [ANCH_NW waves][RT*32][CT*32]
    const int ldr = CT * 32;
#pragma unroll
    for (int i = 0; i < NM; ++i)
#pragma unroll
      for (int j = 0; j < NM; ++j)
        if (i < RT && j < CT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (mlane >> 5);
            red[(mwave * RT * 32 + row) * ldr + j * 32 + (mlane & 31)] = macc[i][j][r];
          }
        }
    __syncthreads();
    for (int idx = tid; idx < PC * EPA; idx += ANCH_NT) {
      const int pc = idx / EPA, e = idx % EPA;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < ANCH_NW; ++w) v += red[(w * RT * 32 + pc) * ldr + e];
      out[idx] = v;
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < ANCH_ITEMS; ++it) {
    const int item = tid + it * ANCH_NT;
    if (item < nitems) {
      const int pc = item / EQ, q = item % EQ;
      *reinterpret_cast<f32x4*>(&out[pc * EPA + q * 4]) = acc[it];
    }
  }
}

// one block per utterance: sum chunks, normalise, Gram max incl. diagonal,
// argmin (first index on ties), gather (modules.py:522-537)
__global__ void anchor_final_kernel(int C, int E, int EPA, int P, int nch,
                                    const float* __restrict__ partial,
                                    float* __restrict__ attr, float* __restrict__ asets,
                                    float* __restrict__ asum, int32_t* __restrict__ choice) {
  extern __shared__ float sm[];
  const int PC = P * C;
  float* S = sm;                 // [PC][EPA] summed
  float* sim = S + PC * EPA;     // [P]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < PC * EPA; i += blockDim.x) {
    // 8 independent partial sums: the chunk loads of a thread are then in flight
    // together instead of one dependent load per add (24 -> see profiles/README.md)
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* pp = partial + (int64_t)b * nch * PC * EPA + i;
    int ch = 0;
    for (; ch + 8 <= nch; ch += 8)
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += pp[(int64_t)(ch + u) * PC * EPA];
    for (; ch < nch; ++ch) s[0] += pp[(int64_t)ch * PC * EPA];
    S[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  __syncthreads();
  const int EP = EPA - 4;
  // normalise in place (columns e < E are rewritten, the denominator column EP is only read)
  for (int i = threadIdx.x; i < PC * E; i += blockDim.x) {
    const int pc = i / E, e = i % E;
    const float v = S[pc * EPA + e] / S[pc * EPA + EP];      // modules.py:522-523
    asets[((int64_t)b * PC + pc) * E + e] = v;
    S[pc * EPA + e] = v;
  }
  for (int i = threadIdx.x; i < PC; i += blockDim.x) asum[(int64_t)b * PC + i] = S[i * EPA + EP];
  __syncthreads();
  // Gram max over the full C x C matrix incl. the diagonal (K7) of the NORMALISED sets (as the
  // reference: asets @ asets^T, modules.py:526-530): one thread per (p, c1, c2) entry -- P * C * C
  // short dot products in parallel instead of P threads walking C * C * E divisions each
  {
    float* gram = S + PC * EPA;    // (sim lives there; the Gram entries go behind it)
    float* ge = gram + P;
    const int CC = C * C;
    for (int i = threadIdx.x; i < P * CC; i += blockDim.x) {
      const int p = i / CC, c1 = (i % CC) / C, c2 = i % C;
      float dot = 0.f;
      for (int e = 0; e < E; ++e) dot += S[(p * C + c1) * EPA + e] * S[(p * C + c2) * EPA + e];
      ge[i] = dot;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      float mx = -INFINITY;
      for (int q = 0; q < CC; ++q) mx = fmaxf(mx, ge[p * CC + q]);
      sim[p] = mx;                                            // modules.py:526-530
    }
  }
  __syncthreads();
  __shared__ int best_s;
  if (threadIdx.x == 0) {
    int best = 0;
    for (int p = 1; p < P; ++p) if (sim[p] < sim[best]) best = p;   // modules.py:533
    choice[b] = best;
    best_s = best;
  }
  __syncthreads();
  const int best = best_s;
  for (int i = threadIdx.x; i < C * E; i += blockDim.x) {
    const int c = i / E, e = i % E;
    attr[((int64_t)b * C + c) * E + e] = S[(best * C + c) * EPA + e];   // modules.py:534-537 (normalised above)
  }
}

// Per-utterance tables of the anchor estimator's backward (chosen subset p*): anchors An, dL/dSnum
// G = dattr / den, and the raw dattr / attractor / den values dL/dSden is formed from.  dL/dSden[c] =
// -sum_e dattr[c][e] attr[c][e] / den[c] used to be a serial loop over E behind two dependent global
// loads per term on one thread per c (~10 us at the head of a 40-us kernel); every thread now forms
// it from the LDS tables (same terms, same order; the zero padding adds exact zeros).
template <int EP, int CP>
__device__ __forceinline__ void anchor_bwd_tables(int b, int E, const AnchorCombos& cb, int pstar,
                                                  const float* __restrict__ dattr,
                                                  const float* __restrict__ anchors,
                                                  const float* __restrict__ attr,
                                                  const float* __restrict__ asum,
                                                  float* An, float* G, float* Dr, float* At, float* dn) {
  constexpr int C = CP;
  for (int i = threadIdx.x; i < C * EP; i += blockDim.x) {
    const int c = i / EP, e = i % EP;
    const int a = cb.idx[pstar][c];
    const float den = asum[((int64_t)b * cb.P + pstar) * C + c];
    // (dattr == nullptr: the caller has put it into Dr -- dattr_from_partials -- and synchronised)
    const float d = dattr ? ((e < E) ? dattr[((int64_t)b * C + c) * E + e] : 0.f) : Dr[i];
    An[i] = (e < E) ? anchors[a * E + e] : 0.f;
    G[i] = (e < E) ? d / den : 0.f;
    Dr[i] = d;
    At[i] = (e < E) ? attr[((int64_t)b * C + c) * E + e] : 0.f;
    if (e == 0) dn[c] = den;
  }
}

template <int EP>
__device__ __forceinline__ float anchor_bwd_g0(const float* Dr, const float* At, const float* dn, int c) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EP; ++e) s += Dr[c * EP + e] * At[c * EP + e];
  return -s / dn[c];
}

// backward through the chosen subset only (argmin has no gradient)
template <int EP, int CP>
__global__ __launch_bounds__(256) void anchor_bwd_kernel(
    int C_, int64_t N, int E, int A, AnchorCombos cb, const float* __restrict__ dattr,
    const float* __restrict__ embed, const float* __restrict__ anchors,
    const float* __restrict__ attr, const float* __restrict__ asum,
    const int32_t* __restrict__ choice, float* __restrict__ dembed,
    float* __restrict__ partial /* [B][chunks][C][EP] */) {
  constexpr int C = CP;
  (void)C_;
  __shared__ float An[MAXC * EP];   // chosen anchors
  __shared__ float G[MAXC * EP];    // dL/dSnum[c][e]
  __shared__ float Dr[MAXC * EP];   // dattr
  __shared__ float At[MAXC * EP];   // attractors
  __shared__ float dn[MAXC];        // soft-assignment sums of the chosen subset
  __shared__ float red[4 * EP];
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  const int pstar = choice[b];
  anchor_bwd_tables<EP, CP>(b, E, cb, pstar, dattr, anchors, attr, asum, An, G, Dr, At, dn);
  __syncthreads();
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  float* db = dembed + (int64_t)b * N * E;
  // the per-utterance tables are uniform: as scalar registers they cost the vector ALU nothing
  // (as LDS broadcasts the compiler hoisted 80 of them into VGPRs: 264 VGPRs, one wave per SIMD)
  float sAn[CP][EP], sG[CP][EP], sg0[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    sg0[c] = uniform(anchor_bwd_g0<EP>(Dr, At, dn, c));
#pragma unroll
    for (int e = 0; e < EP; ++e) { sAn[c][e] = uniform(An[c * EP + e]); sG[c][e] = uniform(G[c * EP + e]); }
  }
  float accs[CP][EP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e < EP; ++e) accs[c][e] = 0.f;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
    float x[EP], dx[EP];
    load_row<EP>(eb + n * E, E, x);
    load_row<EP>(db + n * E, E, dx);
    float s[CP], ds[CP];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      float v = 0.f, w = 0.f;
#pragma unroll
      for (int e = 0; e < EP; ++e) { v += x[e] * sAn[c][e]; w += x[e] * sG[c][e]; }
      s[c] = v; ds[c] = w + sg0[c];
      mx = fmaxf(mx, v);
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { s[c] = expf(s[c] - mx); den += s[c]; }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { s[c] /= den; dot += s[c] * ds[c]; }
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float dl = s[c] * (ds[c] - dot);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        dx[e] += s[c] * sG[c][e] + dl * sAn[c][e];
        accs[c][e] += dl * x[e];
      }
    }
    store_row<EP>(db + n * E, E, dx);
  }
  float* out = partial + ((int64_t)b * nch + ch) * C * EP;
#pragma unroll
  for (int c = 0; c < CP; ++c)
    if (c < C) block_reduce_store<EP>(accs[c], EP, red, out + c * EP);
}

// The estimator backward WITH the separator's embedding-gradient term recomputed in place: the
// fused separator + loss backward (sep_pit_bwd_kernel with dembed == NULL) only produces dattr, and
// this kernel forms dembed = dL/dembed|separator + dL/dembed|estimator in ONE pass -- the same
// additions in the same order as sep_pit_bwd_kernel followed by anchor_bwd_kernel, without
// writing the first term to HBM (42 MB at cfg 2) and reading it back.
template <int EP, int CP, bool EF>
__global__ __launch_bounds__(SEP_NT) void anchor_sep_bwd_kernel(
    int64_t N, int E_, int A, AnchorCombos cb, const float* __restrict__ dattr,
    const float* __restrict__ embed, const float* __restrict__ anchors,
    const float* __restrict__ attr, const float* __restrict__ asum,
    const int32_t* __restrict__ choice,
    int act, int mode, int B, const float* __restrict__ mix_pwr,
    const float2* __restrict__ src, const float2* __restrict__ phasor,
    const int32_t* __restrict__ perm_idx, const float* __restrict__ records,
    float dloss, const float* __restrict__ dloss_dev,
    float* __restrict__ dembed, float* __restrict__ partial /* [B][chunks][C][EP] */,
    const float* __restrict__ gpartial /* or null: dattr = the forward's partials of the utterance's permutation */) {
  const int E = EF ? EP : E_;   // EF: E == EP known at compile time (guard-free 16-byte row accesses)
  constexpr int C = CP;
  __shared__ float An[MAXC * EP];   // chosen anchors
  __shared__ float G[MAXC * EP];    // dL/dSnum[c][e]
  __shared__ float Dr[MAXC * EP];   // dattr
  __shared__ float tab[MAXC * EP];  // attractors (the separator's table)
  __shared__ float dn[MAXC];        // soft-assignment sums of the chosen subset
  __shared__ float red[SEP_NW * EP];
  __shared__ float rec_s[REC + 1];
  __shared__ int perm_s;
  int b, ch; utterance_chunk(b, ch);
  const int nch = gridDim.x;
  const int pstar = choice[b];      // (issued before the record sums: its latency hides behind them)
  sep_pit_perm<CP>(records, perm_idx, b, nch, N, rec_s, &perm_s);
  const float dscale = dloss * (dloss_dev ? *dloss_dev : 1.f);
  if (gpartial != nullptr) {
    if constexpr (CP == SEP_GRAD_MAXC) {
      __syncthreads();                // perm_s
      dattr_from_partials<EP, CP>(gpartial, b, nch, perm_s, dscale, Dr);
      __syncthreads();
      anchor_bwd_tables<EP, CP>(b, E, cb, pstar, nullptr, anchors, attr, asum, An, G, Dr, tab, dn);
    }
  } else {
    anchor_bwd_tables<EP, CP>(b, E, cb, pstar, dattr, anchors, attr, asum, An, G, Dr, tab, dn);
  }
  __syncthreads();
  int perm[MAXC], inv[MAXC];
  nth_perm(C, perm_s, perm);
  for (int i = 0; i < C; ++i) inv[perm[i]] = i;
  const float scale = dscale * 2.f / ((float)B * (float)N);
  const int64_t n0 = (int64_t)ch * CHUNK_N, n1 = min(N, n0 + CHUNK_N);
  const float* eb = embed + (int64_t)b * N * E;
  float* db = dembed + (int64_t)b * N * E;
  float sAn[CP][EP], sG[CP][EP], sg0[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    sg0[c] = uniform(anchor_bwd_g0<EP>(Dr, tab, dn, c));
#pragma unroll
    // (only ONE of the two tables as wave-uniform scalars: both -- 80 values at E = 20 -- overflow the scalar
    // register file, and the compiler parked the excess in the lanes of a vector register: 50 v_readlane +
    // 29 s_nop per bin; the anchors stay per-lane copies in vector registers, round 6)
    for (int e = 0; e < EP; ++e) { sAn[c][e] = An[c * EP + e]; sG[c][e] = uniform(G[c * EP + e]); }
  }
  float accs[CP][EP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int e = 0; e < EP; ++e) accs[c][e] = 0.f;
  for (int64_t n = n0 + threadIdx.x; n < n1; n += SEP_NT) {
    float x[EP], dx[EP];
    load_row<EP>(eb + n * E, E, x);
    // ---- the separator's term (sep_pit_bwd_kernel's arithmetic, attractor table from LDS)
    {
      const float mp = mix_pwr[(int64_t)b * N + n];
      const float2 ph = phasor[(int64_t)b * N + n];
      float2 sv[C];
#pragma unroll
      for (int c = 0; c < C; ++c) sv[c] = src[((int64_t)b * C + c) * N + n];
      float m[CP], dl[CP];
      sep_masks_lds<EP, CP>(act, x, tab, m);
      sep_pit_dlogit<CP>(act, mode, m, mp, ph, sv, inv, scale, dl);
#pragma unroll
      for (int e = 0; e < EP; ++e) dx[e] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < EP; ++e) dx[e] += dl[c] * tab[c * EP + e];
    }
    // ---- the estimator's term (anchor_bwd_kernel's arithmetic)
    float s[CP], ds[CP];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      float v = 0.f, w = 0.f;
#pragma unroll
      for (int e = 0; e < EP; ++e) { v += x[e] * sAn[c][e]; w += x[e] * sG[c][e]; }
      s[c] = v; ds[c] = w + sg0[c];
      mx = fmaxf(mx, v);
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { s[c] = expf(s[c] - mx); den += s[c]; }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { s[c] /= den; dot += s[c] * ds[c]; }
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float dl = s[c] * (ds[c] - dot);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        dx[e] += s[c] * sG[c][e] + dl * sAn[c][e];
        accs[c][e] += dl * x[e];
      }
    }
    store_row<EP>(db + n * E, E, dx);
  }
  float* out = partial + ((int64_t)b * nch + ch) * C * EP;
#pragma unroll
  for (int c = 0; c < CP; ++c)
    if (c < C) block_reduce_store<EP, SEP_NW>(accs[c], EP, red, out + c * EP);
}

__global__ __launch_bounds__(256) void anchor_bwd_final_kernel(
    int B, int C, int E, int EP, int A, int nch, AnchorCombos cb,
    const float* __restrict__ partial, const int32_t* __restrict__ choice,
    float* __restrict__ danchors, float beta) {
  // one block per anchor.  Phase 1: the (utterance, e) pairs of a tile of 64 utterances are
  // spread over all 256 threads (each sums the chunk partials of its pair with the loads in
  // flight together); phase 2: thread e adds the tile's utterances in ascending order --
  // fixed summation order => deterministic scatter over utterances.  (One thread per e walking
  // 8 utterances serially, each behind a dependent `choice` load, took 12.7 us at cfg 2.)
  __shared__ float vals[64][65];
  const int a = blockIdx.x;
  float acc = 0.f;
  for (int b0 = 0; b0 < B; b0 += 64) {
    for (int idx = threadIdx.x; idx < 64 * E; idx += 256) {
      const int bb = idx / E, e = idx % E, b = b0 + bb;
      float v = 0.f;
      if (b < B) {
        const int p = choice[b];
        for (int c = 0; c < C; ++c) {
          if (cb.idx[p][c] != a) continue;
          float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // loads in flight together
          const float* pp = partial + ((int64_t)b * nch * C + c) * EP + e;
          int ch = 0;
          for (; ch + 8 <= nch; ch += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] += pp[(int64_t)(ch + u) * C * EP];
          for (; ch < nch; ++ch) t[0] += pp[(int64_t)ch * C * EP];
          v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
      }
      vals[bb][e] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < E) {
      const int nb = min(64, B - b0);
      for (int bb = 0; bb < nb; ++bb) acc += vals[bb][threadIdx.x];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < E) {
    const int e = threadIdx.x;
    danchors[a * E + e] = (beta != 0.f) ? danchors[a * E + e] + acc : acc;
  }
}

// =========================================================================
// host entry points
// =========================================================================
static int pick_ep(int E) {
  if (E <= 4) return 4;
  if (E <= 8) return 8;
  if (E <= 20) return 20;
  if (E <= 40) return 40;
  if (E <= 64) return 64;
  return 0;
}

#define DISPATCH_EP(EPV, ...)                     \
  switch (EPV) {                                   \
    case 4: { constexpr int EP = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int EP = 8; __VA_ARGS__; } break;   \
    case 20: { constexpr int EP = 20; __VA_ARGS__; } break; \
    case 40: { constexpr int EP = 40; __VA_ARGS__; } break; \
    case 64: { constexpr int EP = 64; __VA_ARGS__; } break; \
    default: break;                                \
  }

// E == EP known at compile time (the usual case: E = 20 / 40): the kernels' row accesses lose their
// per-vector guards (E = 40, C = 3: sep_pit_fwd 44 -> 28 us, sep_pit_bwd 63 -> 52, estimator backward
// 163 -> 131; E = 20: 114 -> 111 us over the four head kernels)
#define DISPATCH_EF(COND, ...)                     \
  if (COND) { constexpr bool EF = true; __VA_ARGS__; } else { constexpr bool EF = false; __VA_ARGS__; }

#define DISPATCH_CP(CV, ...)                      \
  switch (CV) {                                    \
    case 1: { constexpr int CP = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int CP = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int CP = 3; __VA_ARGS__; } break;   \
    case 4: { constexpr int CP = 4; __VA_ARGS__; } break;   \
    default: break;                                \
  }

static int check_common(const char* who, int B, int C, int64_t N, int E) {
  if (!(B > 0 && C > 0 && C <= MAXC && N > 0 && E > 0)) {
    danet_set_error("%s: bad shape B=%d C=%d N=%lld E=%d", who, B, C, (long long)N, E);
    return DANET_ERR_ARG;
  }
  if (pick_ep(E) == 0) {
    danet_set_error("%s: E=%d > 64 unsupported", who, E);
    return DANET_ERR_UNSUPPORTED;
  }
  if (B > 65535) {
    danet_set_error("%s: B > 65535", who);
    return DANET_ERR_UNSUPPORTED;
  }
  return DANET_OK;
}

size_t dn_ws_attractor_truth(int B, int C, int64_t N, int E) {
  const int EP = pick_ep(E);
  return (size_t)B * n_chunks(N) * C * (EP + 1) * sizeof(float);
}

extern "C" int danet_attractor_truth_fwd(danet_stream_t stream_, int mode, int B, int C,
                                         int64_t N, int E, const float* embed,
                                         const float* src_pwr, const float* mix_pwr, float eps,
                                         float* attr, float* denom, void* ws, size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("attractor_truth_fwd", B, C, N, E);
  if (rc) return rc;
  DANET_CHECK_ARG(mode >= 0 && mode <= 2, "attractor_truth_fwd: mode");
  DANET_CHECK_ARG(embed && src_pwr && attr && denom && (mode == 0 || mix_pwr),
                  "attractor_truth_fwd: null pointer");
  if (!ws || ws_bytes < dn_ws_attractor_truth(B, C, N, E)) {
    danet_set_error("attractor_truth_fwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  if (mode == 0 && !mix_pwr) mix_pwr = src_pwr;  // never read for its value
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  if (C * (EPV + 1) <= 128) {
    DISPATCH_EP(EPV, DISPATCH_CP(C, DISPATCH_EF(E == EPV, truth_fwd1_kernel<EP, CP, EF><<<grid, SEP_NT, 0, stream>>>(
                         mode, N, E, embed, src_pwr, mix_pwr, (float*)ws))));
  } else {
    DISPATCH_EP(EPV, DISPATCH_EF(E == EPV, truth_fwd_kernel<EP, EF><<<grid, 256, 0, stream>>>(
                         mode, C, N, E, embed, src_pwr, mix_pwr, (float*)ws)));
  }
  DANET_CHECK_LAUNCH();
  truth_final_kernel<<<B, 128, 0, stream>>>(mode, C, E, EPV, nch, eps, (const float*)ws, attr,
                                            denom);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_attractor_truth_bwd(danet_stream_t stream_, int mode, int B, int C,
                                         int64_t N, int E, const float* dattr,
                                         const float* src_pwr, const float* mix_pwr,
                                         const float* denom, float eps, float* dembed) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("attractor_truth_bwd", B, C, N, E);
  if (rc) return rc;
  DANET_CHECK_ARG(dattr && src_pwr && denom && dembed && (mode == 0 || mix_pwr),
                  "attractor_truth_bwd: null pointer");
  if (mode == 0 && !mix_pwr) mix_pwr = src_pwr;
  const int EPV = pick_ep(E);
  dim3 grid((unsigned)min((int64_t)64, cdiv64(N, 256)), B);
  DISPATCH_EP(EPV, (truth_bwd_kernel<EP><<<grid, 256, 0, stream>>>(
                       mode, C, N, E, dattr, src_pwr, mix_pwr, denom, eps, dembed)));
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// danet_attractor_truth_bwd + the separator's dembed term recomputed (truth_sep_bwd_kernel):
// dembed is WRITTEN; pair it with danet_separate_pit_bwd(dembed = NULL).
extern "C" int danet_attractor_truth_bwd_sep(danet_stream_t stream_, int tmode, int B, int C,
                                             int64_t N, int E, const float* dattr,
                                             const float* src_pwr, const float* mix_pwr,
                                             const float* denom, float eps, const float* embed,
                                             const float* attr, int act, int mode,
                                             const float* src_c64, const float* phasor,
                                             const int32_t* perm_idx, const float* records,
                                             float dloss, const float* dloss_dev, float* dembed,
                                             const float* dattr_partials) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("attractor_truth_bwd_sep", B, C, N, E);
  if (rc) return rc;
  if (dattr_partials && (C != SEP_GRAD_MAXC || !records)) {
    danet_set_error("attractor_truth_bwd_sep: dattr_partials needs C == %d and records", SEP_GRAD_MAXC);
    return DANET_ERR_UNSUPPORTED;
  }
  DANET_CHECK_ARG((dattr || dattr_partials) && src_pwr && mix_pwr && denom && dembed && embed && attr && src_c64 && phasor &&
                  (perm_idx || records), "attractor_truth_bwd_sep: null pointer");
  DANET_CHECK_ARG(tmode >= 0 && tmode <= 2 && (act == 0 || act == 1) && (mode == 0 || mode == 1),
                  "attractor_truth_bwd_sep: mode");
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  DISPATCH_EP(EPV, DISPATCH_CP(C, DISPATCH_EF(E == EPV, truth_sep_bwd_kernel<EP, CP, EF><<<grid, SEP_NT, 0, stream>>>(
                       tmode, N, E, dattr, src_pwr, mix_pwr, denom, eps, embed, attr, act, mode, B,
                       (const float2*)src_c64, (const float2*)phasor, perm_idx, records, dloss,
                       dloss_dev, dembed, dattr_partials))));
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_separate_fwd(danet_stream_t stream_, int act, int B, int C, int64_t N,
                                  int E, const float* mix_pwr, const float* attr,
                                  const float* embed, float* out, float* masks) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("separate_fwd", B, C, N, E);
  if (rc) return rc;
  DANET_CHECK_ARG(act == 0 || act == 1, "separate_fwd: act");
  DANET_CHECK_ARG(mix_pwr && attr && embed && out, "separate_fwd: null pointer");
  const int EPV = pick_ep(E);
  dim3 grid((unsigned)min((int64_t)64, cdiv64(N, 256)), B);
  DISPATCH_EP(EPV, DISPATCH_EF(E == EPV, separate_fwd_kernel<EP, EF><<<grid, 256, 0, stream>>>(
                       act, C, N, E, mix_pwr, attr, embed, out, masks)));
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

size_t dn_ws_separate_bwd(int B, int C, int64_t N, int E) {
  return (size_t)B * n_chunks(N) * C * pick_ep(E) * sizeof(float);
}

extern "C" int danet_separate_bwd(danet_stream_t stream_, int act, int B, int C, int64_t N,
                                  int E, const float* mix_pwr, const float* attr,
                                  const float* embed, const float* dout, float* dembed,
                                  float* dattr, void* ws, size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("separate_bwd", B, C, N, E);
  if (rc) return rc;
  DANET_CHECK_ARG(act == 0 || act == 1, "separate_bwd: act");
  DANET_CHECK_ARG(mix_pwr && attr && embed && dout && dembed && dattr, "separate_bwd: null pointer");
  if (!ws || ws_bytes < dn_ws_separate_bwd(B, C, N, E)) {
    danet_set_error("separate_bwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  DISPATCH_EP(EPV, DISPATCH_CP(C, (separate_bwd_kernel<EP, CP><<<grid, 256, 0, stream>>>(
                       act, C, N, E, mix_pwr, attr, embed, dout, dembed, (float*)ws))));
  DANET_CHECK_LAUNCH();
  sum_chunks_kernel<<<B, 128, 0, stream>>>(nch, C, E, EPV, (const float*)ws, dattr);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

size_t dn_ws_separate_pit(int B, int C, int64_t N, int E) {
  const size_t fwd = (size_t)B * n_chunks(N) * REC * sizeof(float);
  const size_t bwd = (size_t)B * n_chunks(N) * C * pick_ep(E) * sizeof(float);
  return fwd > bwd ? fwd : bwd;
}

// The forward in two stream-ordered parts: part 1 writes the per-chunk cross-error records
// (`records`, dn_ws_separate_pit_records, caller-owned: the backward reads them again),
// part 2 turns them into loss / SNR / permutation index.  The backward derives each utterance's
// permutation from the records itself (same sums, same order), so part 2 is not on the
// critical path between forward and backward: a host may issue it on another stream.
size_t dn_ws_separate_pit_records(int B, int64_t N) {
  return (size_t)B * n_chunks(N) * REC * sizeof(float);
}

extern "C" int danet_separate_pit_fwd_records(danet_stream_t stream_, int act, int mode, int B,
                                              int C, int64_t N, int E, const float* mix_pwr,
                                              const float* attr, const float* embed,
                                              const float* src_c64, const float* phasor,
                                              float* sep_pwr_out, float* records, float* dattr_partials) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("separate_pit_fwd", B, C, N, E);
  if (rc) return rc;
  if (dattr_partials && C != SEP_GRAD_MAXC) {
    danet_set_error("separate_pit_fwd: dattr_partials is offered for C == %d (C! x C x E sums per thread)", SEP_GRAD_MAXC);
    return DANET_ERR_UNSUPPORTED;
  }
  DANET_CHECK_ARG((act == 0 || act == 1) && (mode == 0 || mode == 1), "separate_pit_fwd: act / mode");
  DANET_CHECK_ARG(mix_pwr && attr && embed && src_c64 && phasor && records,
                  "separate_pit_fwd: null pointer");
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  if (dattr_partials) {
    DISPATCH_EP(EPV, DISPATCH_EF(E == EPV, sep_pit_fwd_kernel<EP, SEP_GRAD_MAXC, EF, true><<<grid, SEP_NT, 0, stream>>>(
                         act, mode, B, N, E, mix_pwr, attr, embed, (const float2*)src_c64, (const float2*)phasor,
                         sep_pwr_out, records, dattr_partials)));
  } else {
    DISPATCH_EP(EPV, DISPATCH_CP(C, DISPATCH_EF(E == EPV, sep_pit_fwd_kernel<EP, CP, EF, false><<<grid, SEP_NT, 0, stream>>>(
                         act, mode, B, N, E, mix_pwr, attr, embed, (const float2*)src_c64,
                         (const float2*)phasor, sep_pwr_out, records, nullptr))));
  }
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

size_t dn_ws_separate_pit_grad(int B, int C, int64_t N, int E) {
  if (C != SEP_GRAD_MAXC) return 0;                      // (not offered: the caller keeps danet_separate_pit_bwd)
  int np = 1;
  for (int i = 2; i <= C; ++i) np *= i;
  return (size_t)B * n_chunks(N) * np * C * pick_ep(E) * sizeof(float);
}

extern "C" int danet_separate_pit_final(danet_stream_t stream_, int B, int C, int64_t N, float eps,
                                        const float* records, float* loss, float* snr,
                                        int32_t* perm_idx) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(B > 0 && C > 0 && C <= MAXC && N > 0 && records && loss && perm_idx,
                  "separate_pit_final: bad argument");
  pit_final_kernel<<<1, PIT_FINAL_THREADS, 0, stream>>>(B, C, N, n_chunks(N), eps, records, loss, snr,
                                                        perm_idx);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_separate_pit_bwd(danet_stream_t stream_, int act, int mode, int B, int C,
                                      int64_t N, int E, const float* mix_pwr, const float* attr,
                                      const float* embed, const float* src_c64,
                                      const float* phasor, const int32_t* perm_idx,
                                      const float* records, float dloss,
                                      const float* dloss_dev, float* dembed, float* dattr,
                                      void* ws, size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("separate_pit_bwd", B, C, N, E);
  if (rc) return rc;
  DANET_CHECK_ARG((act == 0 || act == 1) && (mode == 0 || mode == 1), "separate_pit_bwd: act / mode");
  // dembed == NULL: only dattr is produced (the caller's estimator backward recomputes the
  // separator's contribution to dembed: danet_attractor_anchor_bwd_embed_sep)
  DANET_CHECK_ARG(mix_pwr && attr && embed && src_c64 && phasor && (perm_idx || records) && dattr,
                  "separate_pit_bwd: null pointer (one of perm_idx / records is required)");
  if (!ws || ws_bytes < dn_ws_separate_pit(B, C, N, E)) {
    danet_set_error("separate_pit_bwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  DISPATCH_EP(EPV, DISPATCH_CP(C, DISPATCH_EF(E == EPV, sep_pit_bwd_kernel<EP, CP, EF><<<grid, SEP_NT, 0, stream>>>(
                       act, mode, B, N, E, mix_pwr, attr, embed, (const float2*)src_c64,
                       (const float2*)phasor, perm_idx, records, dloss, dloss_dev, dembed,
                       (float*)ws))));
  DANET_CHECK_LAUNCH();
  sum_chunks_kernel<<<B, 128, 0, stream>>>(nch, C, E, EPV, (const float*)ws, dattr);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

static int anchor_check(const char* who, int B, int C, int64_t N, int E, int A) {
  int rc = check_common(who, B, C, N, E);
  if (rc) return rc;
  if (!(A >= C && A <= MAXA)) {
    danet_set_error("%s: need C <= A <= %d (A=%d)", who, MAXA, A);
    return DANET_ERR_UNSUPPORTED;
  }
  return DANET_OK;
}

static int n_combos(int A, int C) {
  long r = 1;
  for (int i = 1; i <= C; ++i) r = r * (A - C + i) / i;
  return (int)r;
}

size_t dn_ws_attractor_anchor(int B, int C, int64_t N, int E, int A) {
  const int EP = pick_ep(E), P = n_combos(A, C);
  const size_t fwd = (size_t)B * n_chunks_anchor_fwd(N) * P * C * (EP + 4) * sizeof(float);
  const size_t bwd = (size_t)B * n_chunks(N) * C * EP * sizeof(float);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int danet_attractor_anchor_fwd(danet_stream_t stream_, int B, int C, int64_t N, int E,
                                          int A, const float* embed, const float* anchors,
                                          float* attr, float* asets, float* asum,
                                          int32_t* choice, void* ws, size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = anchor_check("attractor_anchor_fwd", B, C, N, E, A);
  if (rc) return rc;
  DANET_CHECK_ARG(embed && anchors && attr && asets && asum && choice,
                  "attractor_anchor_fwd: null pointer");
  if (!ws || ws_bytes < dn_ws_attractor_anchor(B, C, N, E, A)) {
    danet_set_error("attractor_anchor_fwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  AnchorCombos cb;
  make_combos(A, C, cb);
  const int PC = cb.P * C, EPV = pick_ep(E), EPA = EPV + 4;
  if (PC * (EPA / 4) > 1024) {
    danet_set_error("attractor_anchor_fwd: P*C*E too large");
    return DANET_ERR_UNSUPPORTED;
  }
  const int nch = n_chunks_anchor_fwd(N);
  // the [PC][EPA] contraction runs on the matrix cores when it fits 2 x 2 tiles of 32 x 32
  int RT = 0, CT = 0;
  if (PC <= 64 && EPA <= 64) { RT = cdiv(PC, 32); CT = cdiv(EPA, 32); }
  size_t lds = ((size_t)ANCH_TN * EPA + (size_t)ANCH_TN * (PC + 1) + (size_t)A * EPV + 64) * sizeof(float);
  const size_t lds_red = (size_t)ANCH_NW * RT * 32 * CT * 32 * sizeof(float);
  if (lds_red > lds) lds = lds_red;
  dim3 grid(nch, B);
#define LAUNCH_ANCHOR(AT_, CT_, T11_)                                                        \
  DISPATCH_EP(EPV, DISPATCH_EF(E == EPV, {                                                    \
    DANET_CHECK_HIP(hipFuncSetAttribute((const void*)anchor_fwd_kernel<EP, AT_, CT_, T11_, EF>, \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
    anchor_fwd_kernel<EP, AT_, CT_, T11_, EF><<<grid, ANCH_NT, lds, stream>>>(C, N, E, A, cb, embed, \
                                                                anchors, (float*)ws, RT, CT); \
  }))
  const bool t11 = (RT == 1 && CT == 1);
  if (A == 6 && C == 2) { if (t11) { LAUNCH_ANCHOR(6, 2, true); } else { LAUNCH_ANCHOR(6, 2, false); } }   // default.json: NUM_ANCHOR 6, 2 speakers
  else if (A == 6 && C == 3) { LAUNCH_ANCHOR(6, 3, false); }     // 3-speaker configs
  else { LAUNCH_ANCHOR(0, 0, false); }
  DANET_CHECK_LAUNCH();
  const size_t lds2 = ((size_t)PC * EPA + cb.P + (size_t)cb.P * C * C) * sizeof(float);
  anchor_final_kernel<<<B, 512, lds2, stream>>>(C, E, EPA, cb.P, nch, (const float*)ws, attr,
                                                asets, asum, choice);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// Backward in two stream-ordered parts so that a host may issue the second one elsewhere: part 1
// (everything the rest of backward waits for: dembed += ...; per-chunk anchor-gradient partials
// into `ws`) and part 2 (danchors from the partials: only the optimiser needs it).
extern "C" int danet_attractor_anchor_bwd_embed(danet_stream_t stream_, int B, int C, int64_t N,
                                                int E, int A, const float* dattr,
                                                const float* embed, const float* anchors,
                                                const float* attr, const float* asum,
                                                const int32_t* choice, float* dembed, void* ws,
                                                size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = anchor_check("attractor_anchor_bwd", B, C, N, E, A);
  if (rc) return rc;
  DANET_CHECK_ARG(dattr && embed && anchors && attr && asum && choice && dembed,
                  "attractor_anchor_bwd: null pointer");
  if (!ws || ws_bytes < dn_ws_attractor_anchor(B, C, N, E, A)) {
    danet_set_error("attractor_anchor_bwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  AnchorCombos cb;
  make_combos(A, C, cb);
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  DISPATCH_EP(EPV, DISPATCH_CP(C, (anchor_bwd_kernel<EP, CP><<<grid, 256, 0, stream>>>(
                       C, N, E, A, cb, dattr, embed, anchors, attr, asum, choice, dembed,
                       (float*)ws))));
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// danet_attractor_anchor_bwd_embed + the separator's dembed term recomputed (see
// anchor_sep_bwd_kernel): dembed is WRITTEN (not accumulated); pair it with
// danet_separate_pit_bwd(..., dembed = NULL, ...) which then only produces dattr.
extern "C" int danet_attractor_anchor_bwd_embed_sep(
    danet_stream_t stream_, int B, int C, int64_t N, int E, int A, const float* dattr,
    const float* embed, const float* anchors, const float* attr, const float* asum,
    const int32_t* choice, int act, int mode, const float* mix_pwr, const float* src_c64,
    const float* phasor, const int32_t* perm_idx, const float* records, float dloss,
    const float* dloss_dev, float* dembed, void* ws, size_t ws_bytes, const float* dattr_partials) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = anchor_check("attractor_anchor_bwd", B, C, N, E, A);
  if (rc) return rc;
  if (dattr_partials && (C != SEP_GRAD_MAXC || !records)) {
    danet_set_error("attractor_anchor_bwd_embed_sep: dattr_partials needs C == %d and records", SEP_GRAD_MAXC);
    return DANET_ERR_UNSUPPORTED;
  }
  DANET_CHECK_ARG((dattr || dattr_partials) && embed && anchors && attr && asum && choice && dembed && mix_pwr && src_c64 &&
                  phasor && (perm_idx || records), "attractor_anchor_bwd_embed_sep: null pointer");
  DANET_CHECK_ARG((act == 0 || act == 1) && (mode == 0 || mode == 1), "attractor_anchor_bwd_embed_sep: act / mode");
  if (!ws || ws_bytes < dn_ws_attractor_anchor(B, C, N, E, A)) {
    danet_set_error("attractor_anchor_bwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  AnchorCombos cb;
  make_combos(A, C, cb);
  const int nch = n_chunks(N), EPV = pick_ep(E);
  dim3 grid(nch, B);
  DISPATCH_EP(EPV, DISPATCH_CP(C, DISPATCH_EF(E == EPV, anchor_sep_bwd_kernel<EP, CP, EF><<<grid, SEP_NT, 0, stream>>>(
                       N, E, A, cb, dattr, embed, anchors, attr, asum, choice, act, mode, B, mix_pwr,
                       (const float2*)src_c64, (const float2*)phasor, perm_idx, records, dloss,
                       dloss_dev, dembed, (float*)ws, dattr_partials))));
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_attractor_anchor_bwd_anchors(danet_stream_t stream_, int B, int C, int64_t N,
                                                  int E, int A, const int32_t* choice,
                                                  float* danchors, const void* ws,
                                                  size_t ws_bytes, float danchors_beta) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = anchor_check("attractor_anchor_bwd", B, C, N, E, A);
  if (rc) return rc;
  DANET_CHECK_ARG(choice && danchors, "attractor_anchor_bwd: null pointer");
  DANET_CHECK_ARG(danchors_beta == 0.f || danchors_beta == 1.f, "attractor_anchor_bwd: beta must be 0 or 1");
  if (!ws || ws_bytes < dn_ws_attractor_anchor(B, C, N, E, A)) {
    danet_set_error("attractor_anchor_bwd: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  AnchorCombos cb;
  make_combos(A, C, cb);
  const int nch = n_chunks(N), EPV = pick_ep(E);
  anchor_bwd_final_kernel<<<A, 256, 0, stream>>>(B, C, E, EPV, A, nch, cb, (const float*)ws,
                                                 choice, danchors, danchors_beta);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}
