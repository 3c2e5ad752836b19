// HBM-bound elementwise / reduction kernels of the DANet hot path (gfx950):
// in-graph front-end (reference main.py:233-240), phase re-attach
// (main.py:281-284,330-335), per-utterance mean-centre (app/modules.py:209-210,
// 244-245), column sums (bias gradients) and the clip + TF1-Adam update
// (main.py:359-363, app/ozers.py:15-18).  All are coalesced grid-stride streams;
// the complex spectra are read as interleaved (re,im) float2.
#include "common.h"
#include "options.h"
#include <atomic>

// ------------------------------------------------------------------ front-end
__global__ void frontend_kernel(int B, int C, int64_t N, const float2* __restrict__ src,
                                float* __restrict__ mix_pwr, float* __restrict__ mix_log,
                                float2* __restrict__ phasor, float* __restrict__ phase,
                                float* __restrict__ src_pwr, float2* __restrict__ mix) {
  const int64_t total = (int64_t)B * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N, n = i % N;
    float re = 0.f, im = 0.f;
    for (int c = 0; c < C; ++c) {                      // main.py:233-234
      const float2 s = src[((int64_t)b * C + c) * N + n];
      re += s.x; im += s.y;
      if (src_pwr) src_pwr[((int64_t)b * C + c) * N + n] = hypotf(s.x, s.y);  // main.py:236
    }
    const float mag = hypotf(re, im);                  // main.py:239
    if (mix) mix[i] = make_float2(re, im);
    if (mix_pwr) mix_pwr[i] = mag;
    if (mix_log) mix_log[i] = log1pf(mag);             // main.py:240
    const float ph = atan2f(im, re);                   // main.py:237-238
    if (phase) phase[i] = ph;
    if (phasor) phasor[i] = make_float2(cosf(ph), sinf(ph));   // main.py:283-284
  }
}

extern "C" int danet_frontend_fwd(danet_stream_t stream, int B, int C, int64_t N,
                                  const float* src_c64, float* mix_pwr, float* mix_log,
                                  float* phasor, float* phase, float* src_pwr,
                                  float* mix_c64) {
  DANET_CHECK_ARG(B > 0 && C > 0 && N > 0 && src_c64, "frontend: bad args");
  const int64_t total = (int64_t)B * N;
  const int grid = (int)min((int64_t)2048, cdiv64(total, 256));
  frontend_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      B, C, N, (const float2*)src_c64, mix_pwr, mix_log, (float2*)phasor, phase, src_pwr,
      (float2*)mix_c64);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// ------------------------------------------------------------- phase re-attach
// permutation p of C elements in itertools.permutations (lexicographic) order
__device__ __forceinline__ void nth_permutation(int C, int p, int* out) {
  int avail[4] = {0, 1, 2, 3};
  int fact = 1;
  for (int i = 2; i < C; ++i) fact *= i;  // (C-1)!
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

__global__ void reattach_kernel(int B, int C, int64_t N, const float* __restrict__ sep_pwr,
                                const float2* __restrict__ phasor,
                                const int32_t* __restrict__ perm_idx,
                                float2* __restrict__ out) {
  const int64_t total = (int64_t)B * C * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i % N;
    const int c = (int)((i / N) % C);
    const int b = (int)(i / (N * C));
    int src_c = c;
    if (perm_idx) {
      int perm[4];
      nth_permutation(C, perm_idx[b], perm);
      src_c = perm[c];                                 // main.py:293-306
    }
    const float pw = sep_pwr[((int64_t)b * C + src_c) * N + n];
    const float2 ph = phasor[(int64_t)b * N + n];
    out[i] = make_float2(ph.x * pw, ph.y * pw);        // main.py:282-284
  }
}

extern "C" int danet_reattach_phase(danet_stream_t stream, int B, int C, int64_t N,
                                    const float* sep_pwr, const float* phasor,
                                    const int32_t* perm_idx, float* out_c64) {
  DANET_CHECK_ARG(B > 0 && C > 0 && C <= 4 && N > 0 && sep_pwr && phasor && out_c64,
                  "reattach: bad args");
  const int64_t total = (int64_t)B * C * N;
  const int grid = (int)min((int64_t)2048, cdiv64(total, 256));
  reattach_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      B, C, N, sep_pwr, (const float2*)phasor, perm_idx, (float2*)out_c64);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// ---------------------------------------------------------------- mean-centre
// One workgroup per (utterance, T-slice) pair would need a second pass; the
// tensors here are a few MB, so: kernel 1 = per-utterance sums with CH chunks
// per utterance (deterministic two-level reduce), kernel 2 = subtract + layout
// change + zero padding.
#define CENTER_CHUNKS 32

__device__ __forceinline__ int64_t center_index(int layout, int B, int T, int ld, int b, int t) {
  return layout == 0 ? ((int64_t)b * T + t) * ld : ((int64_t)t * B + b) * ld;
}

// Both kernels walk whole rows (t) with the threads striding over d: no per-element
// 64-bit division, coalesced row segments.
// The per-utterance sum is accumulated in DOUBLE: the mean is subtracted from every element, so
// its rounding error is a common-mode error of all T*D inputs of the stack (coherent through the
// first layer's 129-term products), and a float32 tree sum of 16512 values of magnitude ~3 is
// off by 3e-7 .. 5e-7 -- more than the rounding of any single input.  Measured at trained cfg-2
// parameters (tools/parity_decompose.py): centred-input error 3.2e-7 rms with the float32 sum vs
// 1.9e-7 for numpy's float32 pairwise mean; with the double sum the mean is the correctly
// rounded float32 value.  16512 adds per utterance: free.
// One thread's share of the chunk sum (rows t0 .. t1 - 1 of utterance b), the SAME terms in the
// same order in the two-launch and the one-launch form.  `vec` (host: D, ld and the base address
// are multiples of four floats): the chunk is walked as 16-byte groups, group tid + 256 k in step
// k, all of a thread's loads in flight together; with KEEP the first CENTER_NV groups stay in
// registers for the caller (the one-launch form subtracts the mean from them: no second read).
#define CENTER_NV 8
template <bool KEEP>
__device__ __forceinline__ double center_thread_sum(int B, int T, int D, const float* __restrict__ in,
                                                    int layout, int ld, int b, int t0, int t1,
                                                    int vec, f32x4* regs) {
  if (vec) {
    const int D4 = D >> 2, n = (t1 - t0) * D4;
    double acc = 0.0;
    int k0 = 0;
    if (KEEP) {
#pragma unroll
      for (int k = 0; k < CENTER_NV; ++k) {
        const int idx = threadIdx.x + k * 256;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < n) {
          const int tr = idx / D4, d4 = idx - tr * D4;
          v = *reinterpret_cast<const f32x4*>(in + center_index(layout, B, T, ld, b, t0 + tr) + 4 * d4);
        }
        regs[k] = v;
      }
#pragma unroll
      for (int k = 0; k < CENTER_NV; ++k)       // (groups beyond n add exact zeros)
        acc += ((double)regs[k][0] + (double)regs[k][1]) + ((double)regs[k][2] + (double)regs[k][3]);
      k0 = CENTER_NV;
    }
    for (int idx = threadIdx.x + k0 * 256; idx < n; idx += 256) {
      const int tr = idx / D4, d4 = idx - tr * D4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + center_index(layout, B, T, ld, b, t0 + tr) + 4 * d4);
      acc += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    }
    return acc;
  }
  double s0 = 0.0, s1 = 0.0;
  for (int t = t0; t < t1; ++t) {
    const float* row = in + center_index(layout, B, T, ld, b, t);
    int d = threadIdx.x;
    for (; d + (int)blockDim.x < D; d += 2 * blockDim.x) { s0 += (double)row[d]; s1 += (double)row[d + blockDim.x]; }
    if (d < D) s0 += (double)row[d];
  }
  return s0 + s1;
}

__global__ __launch_bounds__(256) void center_sum_kernel(int B, int T, int D, const float* __restrict__ in,
                                                         int layout, int ld, int vec,
                                                         double* __restrict__ partial) {
  __shared__ double redd[16];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int tper = cdiv(T, CENTER_CHUNKS);
  const int t0 = ch * tper, t1 = min(T, t0 + tper);
  double v = wave_sum_d(center_thread_sum<false>(B, T, D, in, layout, ld, b, t0, t1, vec, nullptr));
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) redd[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += redd[i];
    partial[b * CENTER_CHUNKS + ch] = s;
  }
}

__global__ void center_apply_kernel(int B, int T, int D, const float* __restrict__ in,
                                    int in_layout, int ld_in, float* __restrict__ out,
                                    int out_layout, int ld_out,
                                    const double* __restrict__ partial,
                                    float* __restrict__ mean_out) {
  const int b = blockIdx.y;
  double s = 0.0;
  for (int i = 0; i < CENTER_CHUNKS; ++i) s += partial[b * CENTER_CHUNKS + i];
  const float mean = (float)(s / (double)((int64_t)T * D));
  if (mean_out && blockIdx.x == 0 && threadIdx.x == 0) mean_out[b] = mean;
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    const float* src = in + center_index(in_layout, B, T, ld_in, b, t);
    float* dst = out + center_index(out_layout, B, T, ld_out, b, t);
    for (int d = threadIdx.x; d < ld_out; d += blockDim.x)
      dst[d] = (d < D) ? src[d] - mean : 0.f;
  }
}

// Both phases in ONE launch (round 6): the grid is the sum kernel's -- CENTER_CHUNKS workgroups per
// utterance --, every workgroup sums its rows, publishes the partial, waits until the other chunks
// of ITS utterance have published theirs, and then subtracts the mean from the rows it has just
// read (they come back from L1 / L2).  The wait needs no zeroed memory and no fence: a partial is
// ONE 16-byte write-through (sc1) store of {sum, tag ^ bits(sum)} with `tag` a process-wide launch
// counter that no earlier launch has used, and a consumer polls the 32 slots of its utterance with
// 16-byte L1-bypassing (sc1) loads until every slot satisfies word1 ^ word0 == tag -- a slot that
// is stale, uninitialised or torn between two stores does not (2^-64) -- and then adds the sums in
// chunk order: same partials, same order, same mean as the two-launch form, bit for bit.  (An
// agent-scope release / acquire pair instead of the self-validating slot costs a write-back of
// the whole L2 behind a kernel that has just filled it: 23-27 us per launch against 13-15 for
// the two launches, `profiles/r06_f_center_one_launch.txt`.)  All workgroups of an utterance must be
// co-resident, so the host takes this form only when B * CENTER_CHUNKS workgroups fit the GPU four
// to a CU (B <= 32 on 256 CUs); kernels of other streams that hold CUs end, so waiting for a slot
// cannot deadlock.  A wait that exceeds its bound (a GPU fault elsewhere) yields NaN means: loud,
// not a hang.
typedef unsigned cv4u __attribute__((__vector_size__(16)));
#define CENTER_SPIN_LIMIT (1u << 16)

__global__ __launch_bounds__(256) void center_fused_kernel(
    int B, int T, int D, const float* __restrict__ in, int in_layout, int ld_in,
    float* __restrict__ out, int out_layout, int ld_out,
    void* slots_ /* NOT restrict: other workgroups write it while this one polls */,
    unsigned long long tag, float* __restrict__ mean_out, int vec, int keep, FillArgs rider) {
  __shared__ double redd[16];
  __shared__ float mean_s;
  const int b = blockIdx.y, ch = blockIdx.x;
  const int tper = cdiv(T, CENTER_CHUNKS);
  const int t0 = ch * tper, t1 = min(T, t0 + tper);
  f32x4 regs[CENTER_NV];
  // (keep: the host has checked that the chunk fits CENTER_NV groups per thread and that `out`
  // takes 16-byte stores with ld_out == D)
  const double ts = keep ? center_thread_sum<true>(B, T, D, in, in_layout, ld_in, b, t0, t1, vec, regs)
                         : center_thread_sum<false>(B, T, D, in, in_layout, ld_in, b, t0, t1, vec, nullptr);
  double v = wave_sum_d(ts);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) redd[w] = v;
  __syncthreads();
  // the 32 slots of this utterance: 512 bytes behind one buffer descriptor
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      static_cast<char*>(slots_) + (size_t)b * CENTER_CHUNKS * 16, 0, CENTER_CHUNKS * 16, 0x00020000);
  if (rider.nseg > 0 && threadIdx.x >= 64) {
    // the rider (danet_encoder_prologue: the recurrent launches' prefill): waves 1-3 stream their share
    // of the fill while wave 0 waits for the utterance's other chunks
    const unsigned long long nb = (unsigned long long)gridDim.x * gridDim.y;
    fill_share(rider, ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * 192 + (threadIdx.x - 64), nb * 192);
  }
  if (threadIdx.x < 64) {          // wave 0
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int i = 0; i < nw; ++i) s += redd[i];
      const unsigned long long sb = __builtin_bit_cast(unsigned long long, s), x = sb ^ tag;
      const cv4u word = {(unsigned)sb, (unsigned)(sb >> 32), (unsigned)x, (unsigned)(x >> 32)};
      __builtin_amdgcn_raw_buffer_store_b128(word, rs, ch * 16, 0, 16 /*sc1: write-through*/);
    }
    // lane i polls slot i mod CENTER_CHUNKS
    const int i = threadIdx.x & (CENTER_CHUNKS - 1);
    unsigned spins = 0;
    bool all_ok;                   // wave-uniform
    unsigned long long sb;
    for (;;) {
      const cv4u g = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16 /*sc1*/);
      sb = (unsigned long long)g[0] | ((unsigned long long)g[1] << 32);
      const unsigned long long x = (unsigned long long)g[2] | ((unsigned long long)g[3] << 32);
      all_ok = __all((x ^ sb) == tag);
      if (all_ok || ++spins > CENTER_SPIN_LIMIT) break;
      __builtin_amdgcn_s_sleep(1);
      asm volatile("" :: "v"(slots_) : "memory");     // the next poll is a NEW load (the builtin is not volatile)
    }
    // the sums in chunk order (every lane of the wave holds slot (lane mod 32)'s)
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < CENTER_CHUNKS; ++k) {
      const unsigned lo = __builtin_amdgcn_readlane((unsigned)sb, k);
      const unsigned hi = __builtin_amdgcn_readlane((unsigned)(sb >> 32), k);
      s += __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
    }
    if (threadIdx.x == 0) mean_s = all_ok ? (float)(s / (double)((int64_t)T * D)) : __builtin_nanf("");
  }
  __syncthreads();
  const float mean = mean_s;
  if (mean_out && ch == 0 && threadIdx.x == 0) mean_out[b] = mean;
  if (keep) {
    const int D4 = D >> 2, n = (t1 - t0) * D4;
#pragma unroll
    for (int k = 0; k < CENTER_NV; ++k) {
      const int idx = threadIdx.x + k * 256;
      if (idx < n) {
        const int tr = idx / D4, d4 = idx - tr * D4;
        *reinterpret_cast<f32x4*>(out + center_index(out_layout, B, T, ld_out, b, t0 + tr) + 4 * d4) =
            regs[k] - mean;
      }
    }
    return;
  }
  for (int t = t0; t < t1; ++t) {
    const float* src = in + center_index(in_layout, B, T, ld_in, b, t);
    float* dst = out + center_index(out_layout, B, T, ld_out, b, t);
    for (int d = threadIdx.x; d < ld_out; d += blockDim.x)
      dst[d] = (d < D) ? src[d] - mean : 0.f;
  }
}

// `mean` doubles as scratch: [B] means (padded to a multiple of four) followed by
// [B][CENTER_CHUNKS] 16-byte slots {double partial sum, launch tag ^ its bits} (keeps the ABI allocation-free and
// re-entrant; nothing in it needs initialising).
int dn_center_mean_elems(int B) { return ((B + 3) & ~3) + 4 * B * CENTER_CHUNKS; }

static std::atomic<unsigned long long> g_center_tag{0x9E3779B97F4A7C15ull};

int dn_center(hipStream_t stream, int B, int T, int D, const float* in, int in_layout, int ld_in, float* out,
              int out_layout, int ld_out, float* mean, const FillArgs* rider, bool* rider_taken) {
  if (rider_taken) *rider_taken = false;
  DANET_CHECK_ARG(B > 0 && T > 0 && D > 0 && in && out && mean, "center: bad args (mean scratch is required)");
  DANET_CHECK_ARG(ld_in >= D && ld_out >= D, "center: ld < D");
  DANET_CHECK_ARG((in_layout | 1) == 1 && (out_layout | 1) == 1, "center: layout must be 0/1");
  DANET_CHECK_ARG(((uintptr_t)mean & 15) == 0, "center: mean scratch must be 16-byte aligned");
  dim3 g1(CENTER_CHUNKS, B);
  // 16-byte groups: the same decision in both forms (it fixes the order of the sum)
  const int vec = D % 4 == 0 && ld_in % 4 == 0 && ((uintptr_t)in & 15) == 0;
#ifndef DANET_CENTER_TWO_LAUNCHES
  if ((int64_t)B * CENTER_CHUNKS <= 4ll * dn_num_cus() && in != out) {
    void* slots = mean + ((B + 3) & ~3);
    const unsigned long long tag = g_center_tag.fetch_add(1, std::memory_order_relaxed);
    const int keep = vec && ld_out == D && ((uintptr_t)out & 15) == 0 &&
                     (int64_t)cdiv(T, CENTER_CHUNKS) * (D / 4) <= CENTER_NV * 256;
    FillArgs none; none.nseg = 0;
    center_fused_kernel<<<g1, 256, 0, stream>>>(B, T, D, in, in_layout, ld_in, out, out_layout, ld_out, slots,
                                                 tag, mean, vec, keep, rider ? *rider : none);
    DANET_CHECK_LAUNCH();
    if (rider && rider_taken) *rider_taken = true;
    return DANET_OK;
  }
#endif
  double* partial = reinterpret_cast<double*>(mean + ((B + 3) & ~3));
  center_sum_kernel<<<g1, 256, 0, stream>>>(B, T, D, in, in_layout, ld_in, vec, partial);
  DANET_CHECK_LAUNCH();
  const int gx = T < 64 ? T : 64;   // row-strided
  dim3 g2(gx, B);
  center_apply_kernel<<<g2, 256, 0, stream>>>(
      B, T, D, in, in_layout, ld_in, out, out_layout, ld_out, partial, mean);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_center(danet_stream_t stream, int B, int T, int D, const float* in,
                            int in_layout, int ld_in, float* out, int out_layout,
                            int ld_out, float* mean) {
  return dn_center((hipStream_t)stream, B, T, D, in, in_layout, ld_in, out, out_layout, ld_out, mean, nullptr,
                   nullptr);
}

// -------------------------------------------------------------------- colsum
// Two-level deterministic reduction.  Level 1: workgroup (64 columns x 4 row
// lanes) sums a band of COLSUM_ROWS rows -- 64 consecutive floats per row lane
// keep the loads coalesced while 4 independent row lanes (and many bands) keep
// enough loads in flight; level 2 sums the <= M/COLSUM_ROWS band partials.
#define COLSUM_ROWS 128

__global__ __launch_bounds__(256) void colsum_partial_kernel(
    int M, int N, const float* __restrict__ A, int lda, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * COLSUM_ROWS, r1 = min(M, r0 + COLSUM_ROWS);
  float s0 = 0.f, s1 = 0.f;
  if (col < N) {
    int r = r0 + rl;
    for (; r + 4 < r1; r += 8) {
      s0 += A[(size_t)r * lda + col];
      s1 += A[(size_t)(r + 4) * lda + col];
    }
    if (r < r1) s0 += A[(size_t)r * lda + col];
  }
  red[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && col < N)
    partial[(size_t)blockIdx.y * N + col] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ void colsum_final_kernel(int nparts, int N, const float* __restrict__ partial,
                                    float* __restrict__ out, float beta) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= N) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * N + col];
  out[col] = (beta != 0.f) ? out[col] + s : s;
}

size_t dn_ws_colsum(int M, int N) {
  return (size_t)cdiv(M, COLSUM_ROWS) * N * sizeof(float);
}

extern "C" int danet_colsum_f32(danet_stream_t stream, int M, int N, const float* A, int lda,
                                float* out, float beta, void* ws, size_t ws_bytes) {
  DANET_CHECK_ARG(M > 0 && N > 0 && A && out && lda >= N, "colsum: bad args");
  if (!ws || ws_bytes < dn_ws_colsum(M, N)) {
    danet_set_error("colsum: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  const int nparts = cdiv(M, COLSUM_ROWS);
  dim3 g(cdiv(N, 64), nparts);
  colsum_partial_kernel<<<g, 256, 0, (hipStream_t)stream>>>(M, N, A, lda, (float*)ws);
  DANET_CHECK_LAUNCH();
  colsum_final_kernel<<<cdiv(N, 64), 64, 0, (hipStream_t)stream>>>(nparts, N, (const float*)ws,
                                                                   out, beta);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// ------------------------------------------------------------- leaky relu
// ops.relu (app/ops.py:93-107): y = max(x, 0) for alpha = 0, max(alpha*x, x) otherwise;
// backward dx = dy * (x > 0 ? 1 : alpha) for 0 <= alpha < 1 (tf.maximum routes the gradient
// to the larger argument; at x = 0 both are equal and TF sends it to the FIRST operand,
// alpha*x, i.e. the slope there is alpha -- for alpha = 0 tf.nn.relu gives 0 at x = 0 too).
__global__ __launch_bounds__(256) void leaky_relu_kernel(int64_t n, const float* __restrict__ x,
                                                         const float* __restrict__ dy, float alpha,
                                                         float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (dy) out[i] = dy[i] * (v > 0.f ? 1.f : alpha);
    else out[i] = v > 0.f ? v : alpha * v;
  }
}

extern "C" int danet_leaky_relu(danet_stream_t stream, int64_t n, const float* x, const float* dy,
                                float alpha, float* out) {
  DANET_CHECK_ARG(n > 0 && x && out, "leaky_relu: bad args");
  DANET_CHECK_ARG(alpha >= 0.f && alpha < 1.f, "leaky_relu: alpha must be in [0, 1)");
  const int grid = (int)min((int64_t)4096, cdiv64(n, 256));
  leaky_relu_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, x, dy, alpha, out);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// ---------------------------------------------------------- clip + TF1 Adam
// One pass over the four flat buffers (16-B accesses).  A NaN gradient stays NaN through the
// clip like tf.clip_by_value (fminf/fmaxf alone would turn it into -clip).  zero_grad: the
// gradient is overwritten with 0 after use, so the next step's backward can accumulate into
// it without a separate fill kernel.
__device__ __forceinline__ void adam_one(float& th, float& gi, float& mi, float& vi, float lr_t,
                                         float b1, float b2, float eps, float clip, float gscale) {
  float g = gi * gscale;
  if (clip > 0.f) g = (g != g) ? g : fminf(fmaxf(g, -clip), clip);   // main.py:359-362
  mi = b1 * mi + (1.f - b1) * g;
  vi = b2 * vi + (1.f - b2) * g * g;
  th -= lr_t * mi / (sqrtf(vi) + eps);                              // eps outside the root (TF1)
}

__global__ __launch_bounds__(256) void adam_clip_kernel(
    int64_t n, float* __restrict__ theta, float* __restrict__ grad, float* __restrict__ m,
    float* __restrict__ v, float lr_t, float b1, float b2, float eps, float clip, float gscale,
    int zero_grad, int vec) {
  const int64_t n4 = vec ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < n4; i += stride) {
    f32x4 th = reinterpret_cast<f32x4*>(theta)[i], g = reinterpret_cast<f32x4*>(grad)[i];
    f32x4 mi = reinterpret_cast<f32x4*>(m)[i], vi = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t_ = th[c], g_ = g[c], m_ = mi[c], v_ = vi[c];
      adam_one(t_, g_, m_, v_, lr_t, b1, b2, eps, clip, gscale);
      th[c] = t_; mi[c] = m_; vi[c] = v_;
    }
    reinterpret_cast<f32x4*>(theta)[i] = th;
    reinterpret_cast<f32x4*>(m)[i] = mi;
    reinterpret_cast<f32x4*>(v)[i] = vi;
    if (zero_grad) reinterpret_cast<f32x4*>(grad)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t i = n4 * 4 + t0; i < n; i += stride) {
    float th = theta[i], g = grad[i], mi = m[i], vi = v[i];
    adam_one(th, g, mi, vi, lr_t, b1, b2, eps, clip, gscale);
    theta[i] = th; m[i] = mi; v[i] = vi;
    if (zero_grad) grad[i] = 0.f;
  }
}

extern "C" int danet_adam_clip_step(danet_stream_t stream, int64_t n, float* theta,
                                    float* grad, float* m, float* v, float lr_t,
                                    float beta1, float beta2, float eps, float clip,
                                    float grad_scale, int zero_grad) {
  DANET_CHECK_ARG(n > 0 && theta && grad && m && v, "adam: bad args");
  const int vec = (((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  const int grid = (int)min((int64_t)2048, cdiv64(vec ? cdiv64(n, 4) : n, 256));
  adam_clip_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, theta, grad, m, v, lr_t, beta1,
                                                          beta2, eps, clip, grad_scale,
                                                          zero_grad, vec);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}
