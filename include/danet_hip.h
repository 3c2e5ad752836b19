/*
 * libdanet_hip.so -- C ABI of the MI355X-native (gfx950) Deep Attractor
 * Network hot path.
 *
 * The reference (khaotik/DaNet-Tensorflow) has NO native boundary: its hot
 * path is TF1 graph ops built by pure-Python plugin classes
 * (app/modules.py:11-93, registries app/hparams.py:72-100).  This header is
 * therefore the boundary a maintainer would bind from those plugin classes
 * (ctypes stub shown in INTEGRATION.md); each entry point cites the reference
 * op sequence it replaces (file:line under the reference root).
 *
 * Conventions
 *  - every pointer is a caller-owned DEVICE pointer (the library never
 *    allocates or frees), dense row-major, fp32 unless the name says
 *    otherwise; `stream` is a hipStream_t passed as void*.
 *  - `ws` is caller-provided device scratch of at least
 *    `danet_workspace_bytes(DANET_WS_<op>, dims, n)` bytes (below); contents
 *    are clobbered.
 *  - return value: 0 = DANET_OK, negative = error (no exceptions cross the
 *    ABI); `danet_last_error()` returns a thread-local message.
 *  - all launches are asynchronous on `stream`; calls are re-entrant per stream
 *    and may come from several host threads (each concurrent call needs its own
 *    `ws`).  Process-wide state is limited to: (1) the tuning/diagnostic option
 *    table below (ints in atomics, changed only through danet_set_option --
 *    the library never reads the process environment), (2) an atomic launch
 *    counter that numbers stream-K launches, (3) one-time per-kernel
 *    attribute setup (thread-safe), (4) the thread-local error string.
 *  - symbols: B batch (mixtures), C speakers, T frames, F bins, E embedding,
 *    H hidden units per LSTM direction, N = T*F time-frequency bins,
 *    A anchors, P = C(A,C) anchor subsets.
 *  - "time-major" = [T][B][.], "batch-major" = [B][T][.].
 */
#ifndef DANET_HIP_H
#define DANET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden and linked against csrc/exports.map: exactly the
 * entry points declared between this push and the pop at the end of the file are exported
 * (`nm -D libdanet_hip.so` shows danet_* and nothing else of the library's own).              */
#pragma GCC visibility push(default)

#define DANET_OK 0
#define DANET_ERR_ARG (-1)          /* bad shape / null pointer / misalignment */
#define DANET_ERR_LAUNCH (-2)       /* hipLaunch / hipMemset failure           */
#define DANET_ERR_UNSUPPORTED (-3)  /* shape outside the compiled envelope     */
#define DANET_ERR_WORKSPACE (-4)    /* ws too small                            */

/* 6 (round 6): the workspace of danet_gemm_x6 starts with a 16 KB region of K-slice tickets (since the end
 * of round 5, then without a version bump) and must be DEDICATED to that entry point (per stream) and
 * zero-initialised once by the caller -- a v5 caller that handed it any shared scratch must change; the
 * option gemm_x6_plan gained the hybrid-schedule bits; danet_gemm_x6_tn_grouped masks odd row pads.     */
/* 7 (round 6, second half): danet_lstm_bwd_prefill is replaced by danet_lstm_train_prefill (the forward
 * launches' buffers and the BPTT rings in one fill launch); danet_encoder_prologue added (the input's
 * centring + that prefill in one launch); the `mean` scratch of danet_center is 16-byte aligned and
 * DANET_WS_CENTER_MEAN grew (16-byte slots of the one-launch form).                                      */
#define DANET_ABI_VERSION 7

typedef void* danet_stream_t;

int danet_abi_version(void);
const char* danet_last_error(void);

/* Options: the kernel-schedule switches a caller may legitimately need (a
 * different GPU partition, a co-scheduled workload, fault-injection tests).
 * Every launch reads the current value; defaults are the shipped configuration.
 * Names (value meaning in csrc/options.h): gemm_dma, splitk_target, gemm_wgs,
 * gemm_maxsplit, gemm_yield, gemm_mfma16, lstm_fwd_un, lstm_fwd_small,
 * lstm_fwd_fused, lstm_bwd_u, lstm_bwd_s, lstm_bwd_twin_xcd,
 * lstm_xmap, lstm_spin_limit, lstm_fault_inject, gemm_x6_plan.
 * danet_set_option / danet_get_option return DANET_ERR_ARG for an unknown
 * name; danet_option_name(i), 0 <= i < danet_option_count(), enumerates.   */
int danet_set_option(const char* name, int value);
int danet_get_option(const char* name, int* value);
void danet_reset_options(void);
int danet_option_count(void);
const char* danet_option_name(int index);

/* Scratch sizes: ONE query for every entry point that takes `ws`.  `dims` are
 * the shape arguments of that entry point in the order listed; returns the
 * bytes needed, or (size_t)-1 for a bad op / dim count (danet_last_error says
 * which).                                                                    */
enum {
  DANET_WS_ISTFT = 0,            /* n_sig, T, N, S          danet_istft                       */
  DANET_WS_GEMM,                 /* M, N, K                 danet_gemm_f32 (split-K slabs; may be 0) */
  DANET_WS_GEMM_STREAMK,         /* M, N, K                 danet_gemm_f32_streamk* (dims ignored)   */
  DANET_WS_COLSUM,               /* M, N                    danet_colsum_f32                  */
  DANET_WS_LSTM,                 /* T, B, H, ndir           danet_lstm_fwd* / danet_lstm_bwd  */
  DANET_WS_ATTRACTOR_TRUTH,      /* B, C, N, E              danet_attractor_truth_fwd         */
  DANET_WS_ATTRACTOR_ANCHOR,     /* B, C, N, E, A           danet_attractor_anchor_*          */
  DANET_WS_SEPARATE_BWD,         /* B, C, N, E              danet_separate_bwd                */
  DANET_WS_SEPARATE_PIT,         /* B, C, N, E              danet_separate_pit_bwd            */
  DANET_WS_SEPARATE_PIT_RECORDS, /* B, N                    `records` of danet_separate_pit_* */
  DANET_WS_PIT_MSE,              /* B, C, N                 danet_pit_mse_fwd                 */
  DANET_WS_CENTER_MEAN,          /* B                       `mean` of danet_center            */
  DANET_WS_GEMM_X6,              /* M, N, K1, K2            danet_gemm_x6 (split-K slabs; may be 0) */
  DANET_WS_GEMM_PACK,            /* N, K                    `out` of a danet_gemm_pack_weights job  */
  DANET_WS_GEMM_X6_TN,           /* sum M*N, 128x128 tiles, K   danet_gemm_x6_tn_grouped (may be 0)  */
  DANET_WS_SEPARATE_PIT_GRAD,    /* B, C, N, E              `dattr_partials` of danet_separate_pit_fwd_records;
                                                            0 = not offered for this shape (C != 2)         */
  DANET_WS_COUNT
};
size_t danet_workspace_bytes(int op, const int64_t* dims, int n_dims);

/* ---------------------------------------------------------------- a1 / a2
 * STFT: replaces scipy.signal.stft(x, window=FFT_WND, nperseg=N,
 * noverlap=N-S)[2].astype(complex64).T  (app/utils.py:117-122,
 * app/datasets/TIMIT/process.py:93-97, app/datasets/WSJ0/process.py:175-179).
 * boundary='zeros', padded=True, scaling 1/sum(window).  N in {64..1024},
 * power of two.  x[n_sig][Ls] -> out[n_sig][T][N/2+1] (re,im interleaved).
 * danet_stft_num_frames returns T = 1+ceil(Ls/S) or DANET_ERR_ARG if Ls < N
 * (scipy raises ValueError there).                                        */
int danet_stft_num_frames(int64_t Ls, int N, int S);
int danet_stft(danet_stream_t stream, int n_sig, int64_t Ls, int N, int S,
               const float* x, const float* window, float* out_c64);

/* iSTFT: replaces utils.istft (app/utils.py:53-75): overlap-add of
 * irfft(X[n])*w over frames n < len(range(0, T*S-N, S)), divided by the
 * overlap-added w^2 where non-zero; float64 accumulators and output
 * [n_sig][T*S]; does NOT undo the 1/sum(w) STFT scaling (neither does the
 * reference).                                                              */
int danet_istft(danet_stream_t stream, int n_sig, int T, int N, int S,
                const float* X_c64, const float* window, double* out,
                void* ws, size_t ws_bytes);

/* ---------------------------------------------------------------- a3 / a13
 * In-graph front-end (main.py:233-240): mix = sum_c src; |src|; atan2;
 * |mix|; log1p(|mix|).  src complex64 [B][C][N].  Optional outputs may be
 * NULL.  `phasor` [B][N][2] = (cos phi, sin phi) of the mixture phase
 * ((1,0) where mix == 0, matching atan2(0,0)=0).                            */
int danet_frontend_fwd(danet_stream_t stream, int B, int C, int64_t N,
                       const float* src_c64, float* mix_pwr, float* mix_log,
                       float* phasor, float* phase, float* src_pwr,
                       float* mix_c64);

/* Phase re-attach (main.py:281-284, :330-335): out[b][c] =
 * phasor[b] * sep_pwr[b][perm(c)], perm = perms[perm_idx[b]] in
 * itertools.permutations order, identity when perm_idx == NULL.            */
int danet_reattach_phase(danet_stream_t stream, int B, int C, int64_t N,
                         const float* sep_pwr, const float* phasor,
                         const int32_t* perm_idx, float* out_c64);

/* ---------------------------------------------------------------- a4
 * Per-utterance mean-centre (app/modules.py:209-210, :244-245):
 * out[b] = in[b] - mean_{t,d}(in[b]).  Layout 0 = batch-major [B][T][ld],
 * 1 = time-major [T][B][ld]; in/out layouts are independent (this is where
 * the encoder switches between the API's batch-major tensors and the LSTM
 * stack's time-major ones).  Columns D..ld_out-1 of `out` are zero-filled.
 * `mean` is REQUIRED scratch+output of DANET_WS_CENTER_MEAN bytes,
 * 16-byte aligned: the first B hold the per-utterance means, the rest per-chunk
 * partial sums (nothing in it needs initialising; one scratch per call in flight).  The sum is accumulated in double and the mean is its correctly
 * rounded float32 value (the mean's error is a common-mode error of every
 * element; a float32 tree sum is off by more than any single element's rounding).
 * The same call is its own backward.                                        */
int danet_center(danet_stream_t stream, int B, int T, int D,
                 const float* in, int in_layout, int ld_in,
                 float* out, int out_layout, int ld_out, float* mean);

/* ---------------------------------------------------------------- a4 / a7
 * fp32 GEMM on MFMA (v_mfma_f32_32x32x2_f32, exact fp32):
 *   C[M][N] = op(A)[M][K] * op(B)[K][N] (+ bias[N]) (+ beta * C), beta in {0,1}
 * transA=0: A stored [M][lda];  transA=1: A stored [K][lda] (op(A)=A^T)
 * transB=0: B stored [K][ldb];  transB=1: B stored [N][ldb] (op(B)=B^T)
 * Replaces the tf.matmul inside ops.lyr_linear (app/ops.py:66-68,72-78) for
 * the hoisted LSTM input projections and the encoder output projection, and
 * all their backward products.  `ws` is used for deterministic split-K.
 * Every GEMM entry point below: an operand (A or B as stored, first to last element)
 * must span less than 2 GiB (DANET_ERR_ARG otherwise); any 4-byte aligned pointer and
 * leading dimension are accepted -- 16-byte aligned operands with ld % 4 == 0 (and
 * K % 4 == 0 for an operand stored contiguous along K) take the faster DMA staging
 * path, with bit-identical results.                                            */
/* `max_workgroups` > 0 caps the launch at that many persistent workgroups
 * (0 = one per tile): lets a product that is overlapped with a latency-bound
 * kernel on another stream stay off the CUs that kernel occupies.             */
int danet_gemm_f32(danet_stream_t stream, int transA, int transB,
                   int M, int N, int K,
                   const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, const float* bias, float beta,
                   void* ws, size_t ws_bytes, int max_workgroups);

/* Same product, stream-K schedule: G persistent workgroups each take an equal
 * share of the (tile, k-iteration) space of their XCD band; cut tiles are finished
 * in-kernel in a fixed order (bit-reproducible, no second kernel).  Faster than
 * danet_gemm_f32 for a product that has the GPU to itself (critical-path dX / dYc),
 * slower when several products share the CUs.  `ws` (>= DANET_WS_GEMM_STREAMK bytes,
 * 16-B aligned) must be zero-initialised once and then only ever be used by this
 * function: it keeps the hand-off flags of earlier launches.  `ws` >=
 * DANET_WS_GEMM_STREAMK bytes.                                                 */
int danet_gemm_f32_streamk(danet_stream_t stream, int transA, int transB,
                           int M, int N, int K,
                           const float* A, int lda, const float* B, int ldb,
                           float* C, int ldc, const float* bias, float beta,
                           void* ws, size_t ws_bytes);
/* A GROUP of up to 6 products that share K and the transpose flags, as ONE stream-K
 * launch over the union of their tiles: the four weight-gradient products of a BiLSTM
 * layer (dWx, dWh per direction; K = T*B) then fill the GPU evenly in one kernel
 * instead of 4 split-K launches + 4 reduce launches.  `max_workgroups` > 0 caps the
 * persistent grid (256 = one per CU leaves room for a co-resident recurrent kernel).
 * Workspace rules as for danet_gemm_f32_streamk.                                */
typedef struct {
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  int M, N;
  const float* bias;   /* [N] or NULL */
  float beta;          /* 0 or 1 */
} danet_gemm_problem_t;
int danet_gemm_f32_streamk_grouped(danet_stream_t stream, int transA, int transB,
                                   int K, int nprob, const danet_gemm_problem_t* probs,
                                   int max_workgroups, void* ws, size_t ws_bytes);
/* K-concatenated product C = op(A1) op(B1) + op(A2) op(B2) (+bias) (+beta*C) on the stream-K
 * schedule (one launch, no slabs, no reduce kernel; K1 % 16 == 0): dX = da_fwd Wx_fwd^T +
 * da_bwd Wx_bwd^T of a BiLSTM layer without a second GEMM and a second read-modify-write of
 * dX.  Workspace: DANET_WS_GEMM_STREAMK.                                                 */
int danet_gemm_f32_streamk_kcat(danet_stream_t stream, int transA, int transB, int M, int N,
                                int K1, const float* A1, int lda1, const float* B1, int ldb1,
                                int K2, const float* A2, int lda2, const float* B2, int ldb2,
                                float* C, int ldc, const float* bias, float beta,
                                void* ws, size_t ws_bytes);

/* fp32 products on the bf16 matrix cores for products whose B operand is a WEIGHT:
 * C[M][N] = A1 B1^T (+ A2 B2^T), A* fp32 [M][lda*] K-contiguous activations, B* weights that
 * danet_gemm_pack_weights has split into three bf16 pieces each (hi + mid + lo == the fp32 value
 * exactly) and laid out in the matrix instruction's operand order.  Six of the nine piece products
 * are accumulated in fp32: the result is as close to the exact product as danet_gemm_f32's
 * (csrc/gemm_x6.hip), at 1.6-1.8 x its speed on the step's projection / dYc / dX shapes.
 * beta = 0; optional bias [N] added to every row.  K* and lda* multiples of 4, A* 16-byte aligned: DANET_ERR_UNSUPPORTED
 * otherwise (the caller falls back to danet_gemm_f32*).  A pack is valid until its weight changes:
 * the host re-packs after every optimizer step (one launch for the whole table).  An event armed
 * with danet_next_launch_events completes with this product, as for stream-K launches.
 * B(n, k) = src[n * stride_n + k * stride_k]; `out`: DANET_WS_GEMM_PACK(N, K) bytes, 16-B aligned.
 * `ws`: DANET_WS_GEMM_X6(M, N, K1, K2) bytes: 16 KB of K-slice tickets + the slabs.  A workspace
 * DEDICATED to this entry point (per stream), zero-initialised once by the caller; it may be grown
 * (zero the new one) and shared by calls of any shape.                                           */
typedef struct {
  const float* src; long long stride_n, stride_k;
  int N, K;
  void* out; size_t out_bytes;
} danet_gemm_pack_t;
int danet_gemm_pack_weights(danet_stream_t stream, int n, const danet_gemm_pack_t* jobs);
int danet_gemm_x6(danet_stream_t stream, int M, int N,
                  int K1, const float* A1, int lda1, const void* B1_packed,
                  int K2, const float* A2, int lda2, const void* B2_packed,
                  float* C, int ldc, const float* bias, void* ws, size_t ws_bytes);

/* The same arithmetic for up to 6 products C (+)= A^T B that share K, with BOTH operands
 * activations: A [K][lda] (M contiguous), B [K][ldb] (N contiguous) -- the weight gradients of a
 * layer, dW = x^T da with K = T*B.  Both operands are split inside the kernel; `bias` must be NULL,
 * beta 0 or 1.  lda / ldb multiples of 4, A / B 16-byte aligned: DANET_ERR_UNSUPPORTED otherwise.
 * Deterministic (K slices are summed in slice order).  `ws`: DANET_WS_GEMM_X6_TN(sum of M*N over
 * the products, sum of tile_rows(M)*ceil(N/128), K) bytes, tile_rows(M) = ceil(M/128) -- except that
 * 1..4 rows beyond a multiple of 128 (M > 128, N % 4 == 0: the 129 spectrogram bins of the bottom
 * layer's dWx) are not a tile row: tile_rows(M) = M/128, those rows are exact fp32 FMA chains.
 * Row pads: rows are read in whole 16-byte groups inside their pitch (lda / ldb).  Pad values never reach a
 * stored result: for an even M / N they only meet accumulator rows / columns that are not stored, for an
 * odd M / N (where the last valid value shares its split pair with the first pad value) the launch takes
 * a kernel variant that masks the pad per element -- the pad need not be initialised or finite.
 * Non-finite data: an operand value that is Inf or rounds to Inf in bf16 (|x| >= 3.39e38) turns the
 * result rows / columns of BOTH values of its pair (m and m ^ 1, resp. n and n ^ 1) into NaN -- the
 * split forms remainders as Inf * 0 for the partner (csrc/common.h); finite data are unaffected.  */
int danet_gemm_x6_tn_grouped(danet_stream_t stream, int K, int nprob, const danet_gemm_problem_t* probs,
                             void* ws, size_t ws_bytes);

/* Events that ride on a kernel's own dispatch packet instead of separate hipEventRecord calls:
 * `start` / `stop` (hipEvent_t; either may be NULL) are attached to the NEXT launch the calling host
 * thread makes through danet_gemm_f32_streamk*, danet_gemm_x6 (stop only; a start event is dropped)
 * or danet_lstm_fwd / danet_lstm_fwd_fused / danet_lstm_bwd (both), and are consumed by that call
 * even if it fails.  Uses: (1) fork without an event record -- another stream waits for `stop` with
 * danet_stream_wait_event; a hipEventRecord behind the launch costs the launching stream ~4 us
 * before its next kernel, the attached event ~1 us (tools/csrc/event_gap.hip); (2) timing a
 * recurrent kernel from its own start / stop time stamps (events created with timing enabled),
 * which does not delay the launch relative to work on other streams.                           */
int danet_next_launch_events(void* start, void* stop);
int danet_event_create(void** event);
int danet_event_destroy(void* event);
int danet_stream_wait_event(danet_stream_t stream, void* event);

/* out[N] = sum_m A[m][n] (+ beta*out): bias gradients.                     */
int danet_colsum_f32(danet_stream_t stream, int M, int N, const float* A,
                     int lda, float* out, float beta, void* ws,
                     size_t ws_bytes);

/* ---------------------------------------------------------------- a5-a7
 * Recurrent half of Model.lyr_lstm / _lyr_bilstm (main.py:76-132,
 * app/modules.py:120-137, app/ops.py:110-148) as ONE persistent launch for
 * all T steps and both directions.
 *   gx_d   [T][B][4H]  hoisted x_t*Wx + b (gate column blocks g|i|f|o)
 *   Wh_d   [H][ldw]    recurrent rows of the reference's W[D+H][4H]
 *   ypad   [T+2][B][ldy] layer output, time t in block t+1; blocks 0 and T+1
 *          are zeroed by the call (zero initial state, main.py:108-123).
 *          direction 0 (fwd) writes columns [0,H), direction 1 (bwd, the
 *          reversed scan of app/modules.py:132-136) writes [H,2H).
 *   gates_d[T][B][4H]  saved post-activation g,i,f,o (may alias gx_d)
 *   cell_d [T][B][H]   saved c_t
 * ndir = 1 (lstm-orig) or 2 (bilstm-orig); *_b pointers ignored if ndir=1.
 * Requirements: H % 4 == 0, H <= 608, ypad 16-byte aligned, ldy % 4 == 0,
 * all workgroups of the launch co-resident (checked: <= the device's CU count).
 * The call first fills ypad blocks 1..T with the bit pattern 0xFFFFFFFF ("not
 * yet published": the exchanged state is its own flag), so ypad must not be
 * read concurrently.
 * Hand-off status: `status` is an optional caller-owned DEVICE int32 that the
 * kernels set non-zero when a bounded inter-workgroup wait timed out (the
 * outputs of that launch are then invalid).  It is STICKY: the library never
 * clears it, so one word can serve every launch of a training run and be read
 * back once in a while (later launches that find it non-zero give up their
 * waits early instead of spinning to the bound).  status == NULL: ws word 0 is
 * used instead and is zeroed by the call.  The value stored on a timeout is
 * DANET_STATUS_TIMEOUT, the bit pattern of 1.0f: the word may therefore live
 * inside a float32 buffer that is SUM-all-reduced across ranks (a data-parallel
 * host puts it behind its gradient bucket, so every rank learns of a timeout on
 * any rank from the collective it issues anyway).  The word may also be pinned,
 * device-mapped HOST memory: the kernels touch it only on the timeout path.   */
#define DANET_STATUS_TIMEOUT 0x3F800000
int danet_lstm_fwd(danet_stream_t stream, int T, int B, int H, int ndir,
                   const float* gx_f, const float* gx_b,
                   const float* Wh_f, const float* Wh_b, int ldw,
                   float* ypad, int ldy,
                   float* gates_f, float* gates_b,
                   float* cell_f, float* cell_b,
                   void* ws, size_t ws_bytes, int32_t* status, int flags);

/* `flags` of danet_lstm_fwd / danet_lstm_fwd_fused / danet_lstm_bwd: DANET_LSTM_PREFILLED = the
 * caller has prefilled this launch's buffers with danet_lstm_fwd_prefill / danet_lstm_train_prefill
 * (one fill launch for the buffers of ALL layers instead of one per call); 0 = the call prefills
 * its own buffers.                                                                          */
#define DANET_LSTM_PREFILLED 1
/* danet_lstm_bwd with db only: leave the per-cluster bias-gradient partials in the workspace and do NOT
 * touch db; the caller finishes with danet_lstm_bwd_db_reduce on any stream ordered behind the
 * launch (the workspace must stay untouched until then).                                     */
#define DANET_LSTM_DB_DEFERRED 2
int danet_lstm_fwd_prefill(danet_stream_t stream, int T, int B, int ldy, int n,
                           float* const* ypads, void* const* wss /* NULL or n workspaces */);
/* (ABI 7, replaces danet_lstm_bwd_prefill) a TRAIN step prefills the n forward launches' buffers AND the
 * rings of the n BPTT launches that will follow (workspaces of their own, DANET_WS_LSTM bytes each) in
 * ONE fill launch at the head of the forward pass; danet_lstm_bwd then takes DANET_LSTM_PREFILLED too.
 * DANET_ERR_UNSUPPORTED (nothing launched) when (B, H, ndir) is outside the reduce-scatter BPTT
 * geometry (danet_lstm_bwd_db_supported): prefill the forward with danet_lstm_fwd_prefill and let
 * danet_lstm_bwd prefill its own medium.                                                        */
int danet_lstm_train_prefill(danet_stream_t stream, int T, int B, int H, int ndir, int ldy, int n,
                             float* const* ypads, void* const* fwd_wss /* NULL or n */,
                             void* const* bwd_wss /* n */);
/* The head of the encoder (app/modules.py:209-223) in ONE launch: danet_center of the input (same
 * arguments, same result bit for bit) with the prefill above as a rider of its kernel -- bwd_wss != NULL:
 * danet_lstm_train_prefill's lists, NULL: danet_lstm_fwd_prefill's (inference).  Where the one-launch
 * centring form is not taken (B * 32 workgroups do not fit the GPU four to a CU) it is danet_center
 * followed by one fill launch.                                                                    */
int danet_encoder_prologue(danet_stream_t stream, int B, int T, int D, const float* in, int in_layout,
                           int ld_in, float* out, int out_layout, int ld_out, float* mean,
                           int H, int ndir, int ldy, int n, float* const* ypads,
                           void* const* fwd_wss /* NULL or n */, void* const* bwd_wss /* NULL or n */);

/* The same layer forward with the INPUT projection fused (no hoisted GEMM, no gx
 * tensor): the kernel computes a_t = [x_t, h_{t-1}] W + b itself -- x_t*Wx of step t
 * runs on the matrix cores while the workgroup waits for h_{t-1} to arrive from the
 * other workgroups (that window is ~1 us per step and otherwise idle), with its 32
 * columns of Wx stationary in registers.  x [T][B][ldx] time-major (16-byte aligned,
 * ldx % 4 == 0, columns D..ldx-1 that fall into the last 4-float group must be finite,
 * e.g. zero padding); W_d [D+H][ldw] in the reference layout (rows 0..D-1 input half,
 * rows D.. recurrent half, app/ops.py:138-142); bias_d [4H].  Outputs as danet_lstm_fwd.
 * Envelope: H % 4 == 0, H <= 320, D <= 640, ndir*ceil(B/16)*ceil(H/8) <= CUs;
 * danet_lstm_fwd_fused_supported() returns 1 inside it when the path is expected to be the
 * faster one (B >= 24: below that the hoisted GEMM is cheaper than the extra MFMAs per step;
 * DANET_LSTM_FWD_FUSED=1 forces it, =0 turns it off), otherwise callers hoist the projection
 * (danet_gemm_f32*) and call danet_lstm_fwd.                                  */
int danet_lstm_fwd_fused_supported(int T, int B, int H, int ndir, int D);
int danet_lstm_fwd_fused(danet_stream_t stream, int T, int B, int H, int ndir,
                         const float* x, int ldx, int D,
                         const float* W_f, const float* W_b, int ldw,
                         const float* bias_f, const float* bias_b,
                         float* ypad, int ldy,
                         float* gates_f, float* gates_b,
                         float* cell_f, float* cell_b,
                         void* ws, size_t ws_bytes, int32_t* status, int flags);

/* BPTT of the above.  dy [T][B][lddy] (dir d uses columns [d*H,(d+1)*H)).
 * Outputs da_d [T][B][4H] = dL/d(pre-activation) (16-byte aligned); the caller
 * finishes with GEMMs: dWx = X^T da, dWh = Hprev^T da, dX = da Wx^T.  Same status
 * convention as danet_lstm_fwd.
 * db_d [4H] (optional, NULL = not wanted; 16-byte aligned): the bias gradients
 * sum_{t,b} da_d, overwritten for beta = 0, accumulated into for beta = 1 -- the
 * owner threads of the reduce-scatter kernel add their da to a register per step, so
 * no column-sum launches are needed (NULL: the caller may use danet_colsum_f32 on da).
 * Envelope = the reduce-scatter geometry (one workgroup per CU: at H = 300 up to B = 200,
 * at H = 600 up to B = 96, both directions); danet_lstm_bwd_db_supported() == 1 inside
 * it, DANET_ERR_UNSUPPORTED outside.  With DANET_LSTM_DB_DEFERRED in
 * `flags` db is not touched by this call: danet_lstm_bwd_db_reduce finishes it on any
 * stream ordered behind the launch.                                             */
int danet_lstm_bwd_db_supported(int T, int B, int H, int ndir);
int danet_lstm_bwd(danet_stream_t stream, int T, int B, int H, int ndir,
                   const float* dy, int lddy,
                   const float* Wh_f, const float* Wh_b, int ldw,
                   const float* gates_f, const float* gates_b,
                   const float* cell_f, const float* cell_b,
                   float* da_f, float* da_b, float* db_f, float* db_b, float beta,
                   void* ws, size_t ws_bytes, int32_t* status, int flags);
int danet_lstm_bwd_db_reduce(danet_stream_t stream, int T, int B, int H, int ndir,
                             float* db_f, float* db_b, float beta, const void* ws, size_t ws_bytes);

/* ---------------------------------------------------------------- a8-a10
 * Truth-family attractor estimators (app/modules.py:382-487).
 * mode 0 'truth' (w=1, denom count+1), 1 'truth-threshold' (w=[|mix|>5],
 * denom +eps), 2 'truth-weighted' (w=|mix|, denom +eps).
 * embed [B][N][E], src_pwr [B][C][N], mix_pwr [B][N] -> attr [B][C][E],
 * denom [B][C] (sum of weights, before the +1 / +eps; saved for bwd).      */
int danet_attractor_truth_fwd(danet_stream_t stream, int mode, int B, int C,
                              int64_t N, int E, const float* embed,
                              const float* src_pwr, const float* mix_pwr,
                              float eps, float* attr, float* denom,
                              void* ws, size_t ws_bytes);
/* dembed [B][N][E] += w[b][n] * dattr[b][idx(b,n)][:] / (denom + add)      */
int danet_attractor_truth_bwd(danet_stream_t stream, int mode, int B, int C,
                              int64_t N, int E, const float* dattr,
                              const float* src_pwr, const float* mix_pwr,
                              const float* denom, float eps, float* dembed);
/* The same with the fused separator + loss backward's dembed term RECOMPUTED in the same pass
 * (dembed is written, not accumulated; embed and attr = the forward's inputs / outputs): pair it
 * with danet_separate_pit_bwd(dembed = NULL), which then produces dattr only.                  */
int danet_attractor_truth_bwd_sep(danet_stream_t stream, int tmode, int B, int C, int64_t N, int E,
                                  const float* dattr, const float* src_pwr, const float* mix_pwr,
                                  const float* denom, float eps, const float* embed,
                                  const float* attr, int act, int mode, const float* src_c64,
                                  const float* phasor, const int32_t* perm_idx,
                                  const float* records, float dloss, const float* dloss_dev,
                                  float* dembed, const float* dattr_partials /* or NULL, see
                                  danet_separate_pit_fwd_records; then dattr may be NULL */);

/* ---------------------------------------------------------------- a11
 * Anchor estimator (app/modules.py:490-545, app/ops.py:273-292), fused: one
 * read of the embedding instead of the reference's [B][P][T][F][C] tensors.
 * anchors [A][E]; outputs attr [B][C][E], asets [B][P][C][E] (attractor per
 * subset), asum [B][P][C] (sum of soft assignments), choice int32 [B].     */
int danet_attractor_anchor_fwd(danet_stream_t stream, int B, int C, int64_t N,
                               int E, int A, const float* embed,
                               const float* anchors, float* attr,
                               float* asets, float* asum, int32_t* choice,
                               void* ws, size_t ws_bytes);
/* Backward (through the chosen subset) in two stream-ordered parts: `_embed` does everything the
 * rest of backward waits for (dembed [B][N][E] += ..., per-chunk anchor-gradient partials into ws),
 * `_anchors` reduces the partials in ws to danchors [A][E] (only the optimiser needs it; ws must
 * stay untouched in between; danchors_beta = 1 accumulates instead of overwriting).             */
int danet_attractor_anchor_bwd_embed(danet_stream_t stream, int B, int C, int64_t N, int E, int A,
                                     const float* dattr, const float* embed, const float* anchors,
                                     const float* attr, const float* asum, const int32_t* choice,
                                     float* dembed, void* ws, size_t ws_bytes);
/* `_embed` with the fused separator + loss backward's dembed term RECOMPUTED in the same pass
 * (dembed is written, not accumulated): pair it with danet_separate_pit_bwd(dembed = NULL), which
 * then produces dattr only -- the separator's term is never written to HBM and read back.      */
int danet_attractor_anchor_bwd_embed_sep(danet_stream_t stream, int B, int C, int64_t N, int E, int A,
                                         const float* dattr, const float* embed, const float* anchors,
                                         const float* attr, const float* asum, const int32_t* choice,
                                         int act, int mode, const float* mix_pwr, const float* src_c64,
                                         const float* phasor, const int32_t* perm_idx,
                                         const float* records, float dloss, const float* dloss_dev,
                                         float* dembed, void* ws, size_t ws_bytes,
                                         const float* dattr_partials /* or NULL, see
                                         danet_separate_pit_fwd_records; then dattr may be NULL */);
int danet_attractor_anchor_bwd_anchors(danet_stream_t stream, int B, int C, int64_t N, int E, int A,
                                       const int32_t* choice, float* danchors, const void* ws,
                                       size_t ws_bytes, float danchors_beta);

/* ---------------------------------------------------------------- a12
 * Dot-product separators (app/modules.py:548-603). act 0 = softmax over C
 * ('dot-softmax-orig'), 1 = sigmoid ('dot-sigmoid-orig').
 * out [B][C][N] = mix_pwr * act(embed . attr^T); masks [B][N][C] optional. */
int danet_separate_fwd(danet_stream_t stream, int act, int B, int C,
                       int64_t N, int E, const float* mix_pwr,
                       const float* attr, const float* embed, float* out,
                       float* masks);
/* dembed [B][N][E] = (overwritten), dattr [B][C][E] = (overwritten)        */
int danet_separate_bwd(danet_stream_t stream, int act, int B, int C,
                       int64_t N, int E, const float* mix_pwr,
                       const float* attr, const float* embed,
                       const float* dout, float* dembed, float* dattr,
                       void* ws, size_t ws_bytes);

/* ------------------------------------------------- a12 + a13 + a14 + a15 fused
 * Separator + phase re-attach + PIT-MSE + SNR in ONE pass over the embedding (the
 * training path: app/modules.py:548-603 -> main.py:281-290, 308-309 ->
 * app/ops.py:374-431, 191-222).  Same arithmetic as danet_separate_fwd followed
 * by danet_pit_mse_fwd, but the masks and separated magnitudes stay in registers
 * (sep_pwr_out may be NULL; if given, [B][C][N] is written as well).  `act` 0
 * softmax / 1 sigmoid; `mode` 0 complex MSE (train) / 1 magnitude MSE (valid).
 * The backward recomputes the masks and returns the gradient w.r.t. the
 * embedding (dembed [B][N][E], overwritten) and the attractors (dattr [B][C][E]):
 * danet_pit_mse_bwd + danet_separate_bwd without the dsep round trip.        */
/* The forward is two stream-ordered parts.  Part 1 writes the per-chunk cross-error `records`
 * (DANET_WS_SEPARATE_PIT_RECORDS bytes, caller-owned); part 2 reduces them to loss / SNR /
 * permutation index.  danet_separate_pit_bwd accepts `records` INSTEAD of perm_idx (pass
 * perm_idx = NULL) and derives each utterance's permutation from them itself -- the same sums
 * in the same order, hence the same index -- so part 2 is off the forward -> backward critical
 * path and a host may issue it on another stream.                                          */
/* `dattr_partials` (round 6; optional, DANET_WS_SEPARATE_PIT_GRAD bytes, offered for C == 2): the same pass
 * also leaves the per-chunk attractor-gradient sums of the backward -- for EVERY permutation, without the
 * upstream gradient -- so that the training path needs no danet_separate_pit_bwd at all: hand the buffer
 * (with `records`) to danet_attractor_anchor_bwd_embed_sep / danet_attractor_truth_bwd_sep, which derive each
 * utterance's permutation from the records, add up the partials of that permutation (x dloss) and use the
 * result as dattr.  One read of the embedding and two launches fewer per train step; the same products
 * and sums in the same order as danet_separate_pit_bwd's (dattr equal to an ulp per chunk partial).          */
int danet_separate_pit_fwd_records(danet_stream_t stream, int act, int mode, int B, int C,
                                   int64_t N, int E, const float* mix_pwr, const float* attr,
                                   const float* embed, const float* src_c64, const float* phasor,
                                   float* sep_pwr_out, float* records, float* dattr_partials);
int danet_separate_pit_final(danet_stream_t stream, int B, int C, int64_t N, float eps,
                             const float* records, float* loss, float* snr, int32_t* perm_idx);
int danet_separate_pit_bwd(danet_stream_t stream, int act, int mode, int B, int C, int64_t N,
                           int E, const float* mix_pwr, const float* attr, const float* embed,
                           const float* src_c64, const float* phasor, const int32_t* perm_idx,
                           const float* records, float dloss, const float* dloss_dev,
                           float* dembed, float* dattr, void* ws, size_t ws_bytes);

/* ---------------------------------------------------------------- a14/a15
 * PIT-MSE loss + SNR (app/ops.py:374-431, :191-222; main.py:289-337).
 * src is always the complex64 truth [B][C][N].
 * mode 0: MSE between src and phasor*sep_pwr (train loss, main.py:281-290);
 * mode 1: MSE between |src| and sep_pwr (valid loss, main.py:312-313).
 * Outputs: loss[1]; perm_idx[B] (index into itertools.permutations(range(C)),
 * first on ties); snr[1] (optional) = mean batch_snr of src vs the complex
 * estimate permuted by perm_idx (main.py:308-309, :336-337).  C <= 4.       */
int danet_pit_mse_fwd(danet_stream_t stream, int mode, int B, int C,
                      int64_t N, const float* src_c64, const float* sep_pwr,
                      const float* phasor, float eps, float* loss,
                      float* snr, int32_t* perm_idx, void* ws,
                      size_t ws_bytes);
/* dsep_pwr [B][C][N] = dloss * (dloss_dev ? *dloss_dev : 1) * dL/dsep_pwr
 * (dloss_dev: optional DEVICE scalar, e.g. the upstream gradient of an autograd
 * engine, folded in without a host sync)                                    */
int danet_pit_mse_bwd(danet_stream_t stream, int mode, int B, int C,
                      int64_t N, const float* src_c64, const float* sep_pwr,
                      const float* phasor, const int32_t* perm_idx,
                      float dloss, const float* dloss_dev, float* dsep_pwr);

/* ---------------------------------------------------------------- f-4 (toy encoder)
 * ops.relu (app/ops.py:93-107), the activation of the reference's default `toy` encoder
 * (app/modules.py:96-116).  dy == NULL: out = x > 0 ? x : alpha*x;  dy != NULL (backward):
 * out = dy * (x > 0 ? 1 : alpha).  0 <= alpha < 1.                                   */
int danet_leaky_relu(danet_stream_t stream, int64_t n, const float* x, const float* dy,
                     float alpha, float* out);

/* ---------------------------------------------------------------- a16
 * clip_by_value + tf.train.AdamOptimizer update (main.py:359-363,
 * app/ozers.py:15-18): g = clamp(grad*grad_scale, +-clip);
 * m,v EMA; theta -= lr_t * m / (sqrt(v) + eps), lr_t precomputed by host
 * as lr*sqrt(1-b2^t)/(1-b1^t).  clip <= 0 disables clipping.  A NaN gradient
 * stays NaN through the clip (tf.clip_by_value propagates NaN), so a diverged
 * step poisons the parameters and the caller's NaN-restore (main.py:462-476)
 * sees it.  zero_grad != 0: grad is overwritten with zeros after use (the
 * next backward accumulates into it; no separate fill kernel).             */
int danet_adam_clip_step(danet_stream_t stream, int64_t n, float* theta,
                         float* grad, float* m, float* v, float lr_t,
                         float beta1, float beta2, float eps, float clip,
                         float grad_scale, int zero_grad);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* DANET_HIP_H */
