// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// k-ordered fmaf chain) -- replaces the tf.matmul inside ops.lyr_linear
// (reference app/ops.py:66-68,72-78) for the hoisted LSTM input projections,
// the encoder output projection and every backward product of both.
//
// Tiling: 128x128x16 block tile, 256 threads = 4 waves in 2x2, each wave owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  Operands are
// staged global -> registers -> LDS as [k][mn] (ld 132: conflict-free
// ds_read_b32 fragment reads, 2-way-at-most transposing writes) through a ring of
// three LDS stages (see gemm_kloop: one barrier per k-tile, nothing but the barrier
// itself between the MFMAs of consecutive k-tiles).  Small-output / long-K products (weight gradients) use a
// deterministic split-K: per-slice slabs in `ws`, summed by a second kernel.
#include "common.h"
#include <hip/hip_ext.h>
#include "options.h"
#include <atomic>
#include <stdlib.h>

#define BM 128
#define BN 128
#ifndef BK
#define BK 16
#endif
static_assert(BK == 16, "gemm_kloop places its side work for BK = 16 (2 pieces per operand)");
#define NLD (BK / 8)   // 16-byte loads per thread and operand tile (128 x BK floats / 256 threads)
#define KQ (BK / 4)    // 4-float groups along k
#define LDT 132  // BM + 4
#define NSTAGE 3
#define STG (BK * LDT)      // floats per operand stage
#define GEMM_SMEM_BYTES (NSTAGE * 2 * STG * sizeof(float))

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* slab;  // split-K partials [splitk][M][N] (or null)
  int M, N, K;
  int lda, ldb, ldc;
  int kchunk;   // K range per z-slice (multiple of BK)
  int splitk;
  float beta;
};

typedef unsigned v4u __attribute__((__vector_size__(16)));   // see lstm.hip (b128 builtins)
#define GEMM_OOB 0xfffffff0u   // voffset beyond num_records: the buffer load returns 0, no access

// Buffer view of an operand from a tile's origin (element (mn0, kbeg)) to the operand's last
// element; the host checks that no operand spans 2 GiB or more.  Guards are turned into
// out-of-range OFFSETS (no branches: a conditionally executed load makes the compiler wait
// vmcnt(0) where the paths join, which would serialise the loads hidden behind the MFMAs).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* p, const float* end) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)((end - p) * 4), 0x00020000);
}

// Load piece i (one 16-byte vector, 4-byte aligned) of this thread's share of an operand tile.
// KCONTIG: element (mn, k) at P[mn*ld + k]; else at P[k*ld + mn].  adv: byte offset of the
// k-tile from the view's origin; mn_rem / k_rem: valid extent from the k-tile's origin.  A vector that STARTS out of range is not fetched (zeros); one
// that straddles the edge of a row is fetched whole (what lies beyond is the next row, or is
// cut off by the buffer range at the operand's end) and masked by store_piece.
template <bool KCONTIG>
__device__ __forceinline__ f32x4 load_piece(__amdgpu_buffer_rsrc_t rs, int ld, int mn_rem,
                                            int k_rem, int tid, int i, unsigned adv) {
  const int idx = tid + i * 256;
  int mn, k;
  if (KCONTIG) { mn = idx / KQ; k = (idx % KQ) * 4; }
  else         { k = idx >> 5;  mn = (idx & 31) * 4; }
  const unsigned off = (unsigned)(KCONTIG ? mn * ld + k : k * ld + mn) * 4u + adv;
  const bool ok = mn < mn_rem && k < k_rem;
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : GEMM_OOB, 0, 0));
}

template <bool KCONTIG>
__device__ __forceinline__ void store_piece(float* __restrict__ S, int tid, f32x4 r, int i,
                                            int mn_rem, int k_rem) {
  const int idx = tid + i * 256;
  int mn, k;
  if (KCONTIG) { mn = idx / KQ; k = (idx % KQ) * 4; }
  else         { k = idx >> 5;  mn = (idx & 31) * 4; }
  const int left = KCONTIG ? k_rem - k : mn_rem - mn;   // valid elements of this vector
  r.y = left > 1 ? r.y : 0.f;
  r.z = left > 2 ? r.z : 0.f;
  r.w = left > 3 ? r.w : 0.f;
  if (KCONTIG) {
    S[(k + 0) * LDT + mn] = r.x;
    S[(k + 1) * LDT + mn] = r.y;
    S[(k + 2) * LDT + mn] = r.z;
    S[(k + 3) * LDT + mn] = r.w;
  } else {
    *reinterpret_cast<f32x4*>(&S[k * LDT + mn]) = r;
  }
}

// One operand pair's k-range [kbeg, kend) of the 128x128 tile at (m0, n0), accumulated into
// the wave's 2x2 accumulators.  LDS: [A stage 0..2 | B stage 0..2].
//
// Pipeline (k-tile kt of BK): the global loads of tile kt+2 are issued during tile kt and
// written to the third LDS stage at the end of it; tile kt+1 is already in LDS (written
// during kt-1, published by the barrier that ended kt-1), so the first fragments of kt+1 are
// read BEFORE the barrier that ends kt.  After that barrier the wave therefore starts the
// next 32 MFMAs at once -- with the two-stage ring it had to store, wait for the barrier and
// then wait for its first ds_reads (~300 of 2350 clocks per k-tile with one wave per SIMD,
// which is what the last round of tiles and every few-tile product of the train step run at).
// All side work sits in the shadow of an MFMA: one piece after each MFMA (every piece is
// followed by a sched_barrier, or the machine scheduler gathers them again).
template <bool A_KCONTIG, bool B_KCONTIG>
__device__ __forceinline__ void gemm_kloop(const float* __restrict__ Ap, int lda,
                                           const float* __restrict__ Bp, int ldb,
                                           int M, int N, int m0, int n0, int kbeg, int kend,
                                           float* smem, f32x16 (&acc)[2][2], int yield) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  float* const sA = smem;
  float* const sB = smem + NSTAGE * STG;
  // views from the tile origin, per-k-tile strides in bytes (uniform)
  const float* pa = Ap + (A_KCONTIG ? (size_t)m0 * lda + kbeg : (size_t)kbeg * lda + m0);
  const float* pb = Bp + (B_KCONTIG ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
  const unsigned sta = (A_KCONTIG ? BK : BK * lda) * 4u;
  const unsigned stb = (B_KCONTIG ? BK : BK * ldb) * 4u;
  const int mrem = M - m0, nrem = N - n0;
  // one past the last element this segment may touch
  const float* enda = Ap + (A_KCONTIG ? (size_t)(M - 1) * lda + kend : (size_t)(kend - 1) * lda + M);
  const float* endb = Bp + (B_KCONTIG ? (size_t)(N - 1) * ldb + kend : (size_t)(kend - 1) * ldb + N);
  const __amdgpu_buffer_rsrc_t rsa = tile_rsrc(pa, enda), rsb = tile_rsrc(pb, endb);
  f32x4 ra[NLD], rb[NLD];
  {
    f32x4 ra1[NLD], rb1[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      ra[i] = load_piece<A_KCONTIG>(rsa, lda, mrem, kend - kbeg, tid, i, 0u);
      rb[i] = load_piece<B_KCONTIG>(rsb, ldb, nrem, kend - kbeg, tid, i, 0u);
    }
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        ra1[i] = load_piece<A_KCONTIG>(rsa, lda, mrem, kend - kbeg - BK, tid, i, sta);
        rb1[i] = load_piece<B_KCONTIG>(rsb, ldb, nrem, kend - kbeg - BK, tid, i, stb);
      }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      store_piece<A_KCONTIG>(sA, tid, ra[i], i, mrem, kend - kbeg);
      store_piece<B_KCONTIG>(sB, tid, rb[i], i, nrem, kend - kbeg);
    }
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        store_piece<A_KCONTIG>(sA + STG, tid, ra1[i], i, mrem, kend - kbeg - BK);
        store_piece<B_KCONTIG>(sB + STG, tid, rb1[i], i, nrem, kend - kbeg - BK);
      }
    }
  }
  unsigned adva = 2 * sta, advb = 2 * stb;     // k-tile kt + 2
  __syncthreads();

  const int fa = wm * 64 + (lane & 31);  // fragment column within the tile
  const int fb = wn * 64 + (lane & 31);
  const int fk = lane >> 5;
  int s_cur = 0, s_nxt = STG, s_fill = 2 * STG;
  float a0 = sA[fk * LDT + fa], a1 = sA[fk * LDT + fa + 32];
  float b0 = sB[fk * LDT + fb], b1 = sB[fk * LDT + fb + 32];

  for (int kt = 0; kt < nk; ++kt) {
    // The first instruction after the barrier is an MFMA (its fragments were read before the
    // barrier); everything else of the k-tile -- stage addresses, buffer descriptors, the
    // fragment reads of the next k-pair, the fetch of tile kt + 2 (past the last tile
    // k2rem <= 0 turns every load into an out-of-range one: zeros, stored to the stage nobody
    // reads, so there is no branch in here) -- follows in the MFMAs' shadows.
    float a0n, a1n, b0n, b1n;
    int k2rem;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      // fragments of the next k-pair (of the next k-tile after the last pair: that stage was
      // published by the previous barrier; past the last tile the values are never used)
      if (kk + 1 < BK / 2) {
        const int k = (kk + 1) * 2 + fk;
        a0n = sA[s_cur + k * LDT + fa]; a1n = sA[s_cur + k * LDT + fa + 32];
        b0n = sB[s_cur + k * LDT + fb]; b1n = sB[s_cur + k * LDT + fb + 32];
      } else {
        a0n = sA[s_nxt + fk * LDT + fa]; a1n = sA[s_nxt + fk * LDT + fa + 32];
        b0n = sB[s_nxt + fk * LDT + fb]; b1n = sB[s_nxt + fk * LDT + fb + 32];
      }
      if (kk == 0) {
        k2rem = kend - kbeg - (kt + 2) * BK;
        ra[0] = load_piece<A_KCONTIG>(rsa, lda, mrem, k2rem, tid, 0, adva);
      }
      if (kk == BK / 2 - 2) store_piece<A_KCONTIG>(sA + s_fill, tid, ra[0], 0, mrem, k2rem);
      if (kk == BK / 2 - 1) store_piece<B_KCONTIG>(sB + s_fill, tid, rb[0], 0, nrem, k2rem);
      __builtin_amdgcn_sched_barrier(0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      if (kk == 0) ra[1] = load_piece<A_KCONTIG>(rsa, lda, mrem, k2rem, tid, 1, adva);
      __builtin_amdgcn_sched_barrier(0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      if (kk == 0) rb[0] = load_piece<B_KCONTIG>(rsb, ldb, nrem, k2rem, tid, 0, advb);
      if (kk == BK / 2 - 2) store_piece<A_KCONTIG>(sA + s_fill, tid, ra[1], 1, mrem, k2rem);
      if (kk == BK / 2 - 1) store_piece<B_KCONTIG>(sB + s_fill, tid, rb[1], 1, nrem, k2rem);
      __builtin_amdgcn_sched_barrier(0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      if (kk == 0) rb[1] = load_piece<B_KCONTIG>(rsb, ldb, nrem, k2rem, tid, 1, advb);
      __builtin_amdgcn_sched_barrier(0);
      a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
    }
    if (yield == 1) __builtin_amdgcn_s_sleep(1);
    else if (yield == 2) __builtin_amdgcn_s_sleep(2);
    else if (yield == 4) __builtin_amdgcn_s_sleep(4);
    else for (int z = 0; z < (yield >> 3); ++z) __builtin_amdgcn_s_sleep(8);   // yield x 64 clocks
    __syncthreads();   // publishes stage s_fill; also makes the ring reusable by the next segment
    const int t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
    adva += sta; advb += stb;
  }
}

// ---------------------------------------------------------------------------------------------
// The same k-loop with the operands moved global -> LDS by the DMA path (buffer_load ... lds):
// no staging registers, no LDS store instructions, no edge masks in the loop.  Eligible (host:
// dma_ok) when both operands are 16-byte aligned with ld % 4 == 0 and every K-contiguous operand
// has K % 4 == 0, so that a 16-byte vector never straddles a row's end in the k direction (what
// a vector drags in past the mn edge only reaches output rows / columns that are never stored).
// Out-of-range lanes of a DMA load WRITE ZEROS to LDS (tools/csrc: dma probe), which is what the
// k edge needs.  The LDS image of a DMA load is lane-linear (wave-uniform base + lane * 16), so
// the layout is made by choosing which global vector each lane fetches:
//   mn-contiguous operand: [16 k][128 mn], odd k rows rotated by 32 floats -- the two half-waves
//     of a fragment read (k = 2 kk + lane / 32) then hit disjoint banks;
//   k-contiguous operand: [128 mn][16 k], the four 16-byte slots of a row XOR-ed with
//     (row / 4) % 4 -- 32 rows of one k spread over 16 bank groups (2-way instead of 8-way).
// Fragment order (k = 2 kk + lane / 32) and therefore every sum is identical to gemm_kloop.
template <bool KCONTIG>
__device__ __forceinline__ void dma_lane(int chunk, int lane, int ld, unsigned& off, int& mn, int& kpos) {
  if (KCONTIG) {
    const int row = 16 * chunk + (lane >> 2);
    const int kq = (lane & 3) ^ ((row >> 2) & 3);
    mn = row; kpos = kq * 4;
    off = (unsigned)(row * ld + kq * 4) * 4u;
  } else {
    const int k = 2 * chunk + (lane >> 5);
    const int col = (((lane & 31) * 4) + ((k & 1) ? 96 : 0)) & 127;
    mn = col; kpos = k;
    off = (unsigned)(k * ld + col) * 4u;
  }
}

// float offset of fragment element (tile row/col t + 32 f, k = 2 kk + fk) inside an operand image
template <bool KCONTIG>
__device__ __forceinline__ int frag_off(int t, int f, int kk, int fk) {
  if (KCONTIG) {
    const int sw = (t >> 2) & 3;                       // (t + 32 f) / 4 % 4 == t / 4 % 4
    return (t + 32 * f) * 16 + (((kk >> 1) ^ sw) << 2) + 2 * (kk & 1) + fk;
  }
  return (2 * kk + fk) * 128 + ((t + 32 * f + 32 * fk) & 127);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <bool A_KCONTIG, bool B_KCONTIG>
__device__ __forceinline__ void gemm_kloop_dma(const float* __restrict__ Ap, int lda,
                                               const float* __restrict__ Bp, int ldb,
                                               int M, int N, int m0, int n0, int kbeg, int kend,
                                               float* smem, f32x16 (&acc)[2][2], int yield) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  float* const sA = smem;
  float* const sB = smem + NSTAGE * STG;
  const float* pa = Ap + (A_KCONTIG ? (size_t)m0 * lda + kbeg : (size_t)kbeg * lda + m0);
  const float* pb = Bp + (B_KCONTIG ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
  const unsigned sta = (A_KCONTIG ? BK : BK * lda) * 4u;
  const unsigned stb = (B_KCONTIG ? BK : BK * ldb) * 4u;
  const int mrem = M - m0, nrem = N - n0;
  const float* enda = Ap + (A_KCONTIG ? (size_t)(M - 1) * lda + kend : (size_t)(kend - 1) * lda + M);
  const float* endb = Bp + (B_KCONTIG ? (size_t)(N - 1) * ldb + kend : (size_t)(kend - 1) * ldb + N);
  const __amdgpu_buffer_rsrc_t rsa = tile_rsrc(pa, enda), rsb = tile_rsrc(pb, endb);

  // this wave moves chunks {wave, wave + 4} (1 KB each) of both operand images
  unsigned oa[2], ob[2];
  int ka[2], kb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int mn;
    dma_lane<A_KCONTIG>(wave + 4 * j, lane, lda, oa[j], mn, ka[j]);
    if (mn >= mrem) oa[j] = GEMM_OOB;
    dma_lane<B_KCONTIG>(wave + 4 * j, lane, ldb, ob[j], mn, kb[j]);
    if (mn >= nrem) ob[j] = GEMM_OOB;
  }
#define DMA_A(j, stage_off, adv, krem)                                                              \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr_t)(sA + (stage_off) + (wave + 4 * (j)) * 256), 16, \
      (oa[j] != GEMM_OOB && ka[j] < (krem)) ? oa[j] + (adv) : GEMM_OOB, 0, 0, 0)
#define DMA_B(j, stage_off, adv, krem)                                                              \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr_t)(sB + (stage_off) + (wave + 4 * (j)) * 256), 16, \
      (ob[j] != GEMM_OOB && kb[j] < (krem)) ? ob[j] + (adv) : GEMM_OOB, 0, 0, 0)
  DMA_A(0, 0, 0u, kend - kbeg); DMA_A(1, 0, 0u, kend - kbeg);
  DMA_B(0, 0, 0u, kend - kbeg); DMA_B(1, 0, 0u, kend - kbeg);
  if (nk > 1) {
    DMA_A(0, STG, sta, kend - kbeg - BK); DMA_A(1, STG, sta, kend - kbeg - BK);
    DMA_B(0, STG, stb, kend - kbeg - BK); DMA_B(1, STG, stb, kend - kbeg - BK);
  }
  unsigned adva = 2 * sta, advb = 2 * stb;     // k-tile kt + 2
  __syncthreads();

  const int ta = wm * 64 + (lane & 31);   // fragment row / column within the tile
  const int tb = wn * 64 + (lane & 31);
  const int fk = lane >> 5;
  int s_cur = 0, s_nxt = STG, s_fill = 2 * STG;
  float a0 = sA[frag_off<A_KCONTIG>(ta, 0, 0, fk)], a1 = sA[frag_off<A_KCONTIG>(ta, 1, 0, fk)];
  float b0 = sB[frag_off<B_KCONTIG>(tb, 0, 0, fk)], b1 = sB[frag_off<B_KCONTIG>(tb, 1, 0, fk)];

  for (int kt = 0; kt < nk; ++kt) {
    float a0n, a1n, b0n, b1n;
    const int k2rem = kend - kbeg - (kt + 2) * BK;   // <= 0 past the last tile: every lane out of range
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      {
        const int s = (kk + 1 < BK / 2) ? s_cur : s_nxt, k1 = (kk + 1 < BK / 2) ? kk + 1 : 0;
        a0n = sA[s + frag_off<A_KCONTIG>(ta, 0, k1, fk)]; a1n = sA[s + frag_off<A_KCONTIG>(ta, 1, k1, fk)];
        b0n = sB[s + frag_off<B_KCONTIG>(tb, 0, k1, fk)]; b1n = sB[s + frag_off<B_KCONTIG>(tb, 1, k1, fk)];
      }
      if (kk == 0) DMA_A(0, s_fill, adva, k2rem);
      __builtin_amdgcn_sched_barrier(0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      if (kk == 0) DMA_A(1, s_fill, adva, k2rem);
      __builtin_amdgcn_sched_barrier(0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      if (kk == 0) DMA_B(0, s_fill, advb, k2rem);
      __builtin_amdgcn_sched_barrier(0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      if (kk == 0) DMA_B(1, s_fill, advb, k2rem);
      __builtin_amdgcn_sched_barrier(0);
      a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
    }
    if (yield == 1) __builtin_amdgcn_s_sleep(1);
    else if (yield == 2) __builtin_amdgcn_s_sleep(2);
    else if (yield == 4) __builtin_amdgcn_s_sleep(4);
    else for (int z = 0; z < (yield >> 3); ++z) __builtin_amdgcn_s_sleep(8);   // yield x 64 clocks
    __syncthreads();   // (with a DMA load in flight the compiler waits vmcnt(0) here) publishes s_fill
    const int t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
    adva += sta; advb += stb;
  }
#undef DMA_A
#undef DMA_B
}

// Epilogue: out[row][col] = acc (+ bias[col]) (+ beta * out[row][col]) for the tile at (m0, n0).
// The MFMA result layout (a lane holds one column of 16 rows) would make every store a 4-byte
// access -- 64 store instructions per thread, 128 bytes per row segment (12 us per tile with
// three workgroups per CU).  When the destination allows 16-byte accesses the tile goes through
// the (now free) LDS ring in two halves of 64 rows and leaves as full 512-byte row segments:
// 16 x 16-byte stores per thread.  Same additions in the same order either way.
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[2][2], float* smem,
                                           float* __restrict__ dst, int ldd, int m0, int n0,
                                           int M, int N, const float* __restrict__ bias, float beta) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // D layout (32x32): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col_l = lane & 31, row_l = 4 * (lane >> 5);
  const bool vec = (ldd % 4 == 0) && (((uintptr_t)dst & 15) == 0) &&
                   ((N - n0) >= BN || (N - n0) % 4 == 0);          // uniform
  if (!vec) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + col_l;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + row_l;
          if (row >= M) continue;
          float* c = dst + (size_t)row * ldd + col;
          float v = acc[mt][nt][r] + bv;
          if (beta != 0.f) v += *c;
          *c = v;
        }
      }
    return;
  }
  const int c4 = tid & 31, rr = tid >> 5;
  const int col = n0 + c4 * 4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && col < N) { bv.x = bias[col]; bv.y = bias[col + 1]; bv.z = bias[col + 2]; bv.w = bias[col + 3]; }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            smem[(mt * 32 + (r & 3) + 8 * (r >> 2) + row_l) * LDT + wn * 64 + nt * 32 + col_l] = acc[mt][nt][r];
    }
    __syncthreads();
    if (col < N) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + half * 64 + rr + 8 * i;
        if (row < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[(rr + 8 * i) * LDT + c4 * 4]);
          float* c = dst + (size_t)row * ldd + col;
          v += bv;
          if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(c);
          *reinterpret_cast<f32x4*>(c) = v;
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The register-staged k-loop on v_mfma_f32_16x16x4_f32 (8 passes = 32 clocks per instruction
// instead of the 64 of 32x32x2; same flops per clock, same LDS reads per flop): for the capped
// weight-gradient groups that share every CU with a BPTT kernel.  That kernel's step is a chain of
// 16 dependent MFMAs per wave, and each of them waits for the matrix pipe of its SIMD to finish
// whatever the co-resident GEMM wave has in flight -- half as long with the short instruction.
// The wave's 64x64 sub-tile is 4x4 accumulator tiles of 16x16 (64 VGPRs as before); fragments:
// lane (fr = lane & 15, fq = lane >> 4) reads element (mn = 16 i + fr, k = 4 kk + fq).
template <bool A_KCONTIG, bool B_KCONTIG>
__device__ __forceinline__ void gemm_kloop16(const float* __restrict__ Ap, int lda,
                                             const float* __restrict__ Bp, int ldb,
                                             int M, int N, int m0, int n0, int kbeg, int kend,
                                             float* smem, f32x4 (&acc)[4][4], int yield) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  float* const sA = smem;
  float* const sB = smem + NSTAGE * STG;
  const float* pa = Ap + (A_KCONTIG ? (size_t)m0 * lda + kbeg : (size_t)kbeg * lda + m0);
  const float* pb = Bp + (B_KCONTIG ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
  const unsigned sta = (A_KCONTIG ? BK : BK * lda) * 4u;
  const unsigned stb = (B_KCONTIG ? BK : BK * ldb) * 4u;
  const int mrem = M - m0, nrem = N - n0;
  const float* enda = Ap + (A_KCONTIG ? (size_t)(M - 1) * lda + kend : (size_t)(kend - 1) * lda + M);
  const float* endb = Bp + (B_KCONTIG ? (size_t)(N - 1) * ldb + kend : (size_t)(kend - 1) * ldb + N);
  const __amdgpu_buffer_rsrc_t rsa = tile_rsrc(pa, enda), rsb = tile_rsrc(pb, endb);
  f32x4 ra[NLD], rb[NLD];
  {
    f32x4 ra1[NLD], rb1[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      ra[i] = load_piece<A_KCONTIG>(rsa, lda, mrem, kend - kbeg, tid, i, 0u);
      rb[i] = load_piece<B_KCONTIG>(rsb, ldb, nrem, kend - kbeg, tid, i, 0u);
    }
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        ra1[i] = load_piece<A_KCONTIG>(rsa, lda, mrem, kend - kbeg - BK, tid, i, sta);
        rb1[i] = load_piece<B_KCONTIG>(rsb, ldb, nrem, kend - kbeg - BK, tid, i, stb);
      }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      store_piece<A_KCONTIG>(sA, tid, ra[i], i, mrem, kend - kbeg);
      store_piece<B_KCONTIG>(sB, tid, rb[i], i, nrem, kend - kbeg);
    }
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        store_piece<A_KCONTIG>(sA + STG, tid, ra1[i], i, mrem, kend - kbeg - BK);
        store_piece<B_KCONTIG>(sB + STG, tid, rb1[i], i, nrem, kend - kbeg - BK);
      }
    }
  }
  unsigned adva = 2 * sta, advb = 2 * stb;     // k-tile kt + 2
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  const int fa = wm * 64 + fr, fb = wn * 64 + fr;
  int s_cur = 0, s_nxt = STG, s_fill = 2 * STG;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = sA[fq * LDT + fa + 16 * i]; b[i] = sB[fq * LDT + fb + 16 * i]; }

  for (int kt = 0; kt < nk; ++kt) {
    float an[4], bn[4];
    int k2rem;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      // fragments of the next k-group (of the next k-tile after the last group)
      const int so = (kk + 1 < BK / 4) ? s_cur + ((kk + 1) * 4 + fq) * LDT : s_nxt + fq * LDT;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
          const int slot = i * 4 + j;       // side work in the MFMAs' shadows, one piece per slot
          if (slot == 0 && kk == 0) {
            k2rem = kend - kbeg - (kt + 2) * BK;
            ra[0] = load_piece<A_KCONTIG>(rsa, lda, mrem, k2rem, tid, 0, adva);
          }
          if (slot == 1) { an[0] = sA[so + fa]; an[1] = sA[so + fa + 16]; }
          if (slot == 2) { an[2] = sA[so + fa + 32]; an[3] = sA[so + fa + 48]; }
          if (slot == 3) { bn[0] = sB[so + fb]; bn[1] = sB[so + fb + 16]; }
          if (slot == 4) { bn[2] = sB[so + fb + 32]; bn[3] = sB[so + fb + 48]; }
          if (slot == 5 && kk == 0) ra[1] = load_piece<A_KCONTIG>(rsa, lda, mrem, k2rem, tid, 1, adva);
          if (slot == 6 && kk == 0) rb[0] = load_piece<B_KCONTIG>(rsb, ldb, nrem, k2rem, tid, 0, advb);
          if (slot == 7 && kk == 0) rb[1] = load_piece<B_KCONTIG>(rsb, ldb, nrem, k2rem, tid, 1, advb);
          if (slot == 8 && kk == BK / 4 - 2) store_piece<A_KCONTIG>(sA + s_fill, tid, ra[0], 0, mrem, k2rem);
          if (slot == 10 && kk == BK / 4 - 2) store_piece<A_KCONTIG>(sA + s_fill, tid, ra[1], 1, mrem, k2rem);
          if (slot == 8 && kk == BK / 4 - 1) store_piece<B_KCONTIG>(sB + s_fill, tid, rb[0], 0, nrem, k2rem);
          if (slot == 10 && kk == BK / 4 - 1) store_piece<B_KCONTIG>(sB + s_fill, tid, rb[1], 1, nrem, k2rem);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
    }
    if (yield == 1) __builtin_amdgcn_s_sleep(1);
    else if (yield == 2) __builtin_amdgcn_s_sleep(2);
    else if (yield == 4) __builtin_amdgcn_s_sleep(4);
    else for (int z = 0; z < (yield >> 3); ++z) __builtin_amdgcn_s_sleep(8);   // yield x 64 clocks
    __syncthreads();
    const int t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
    adva += sta; advb += stb;
  }
}

// store_tile for the 4x4 accumulator tiles of gemm_kloop16.
// D layout (16x16): col = lane & 15, row = 4 * (lane >> 4) + r
__device__ __forceinline__ void store_tile16(const f32x4 (&acc)[4][4], float* smem,
                                             float* __restrict__ dst, int ldd, int m0, int n0,
                                             int M, int N, const float* __restrict__ bias, float beta) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int col_l = lane & 15, row_l = 4 * (lane >> 4);
  const bool vec = (ldd % 4 == 0) && (((uintptr_t)dst & 15) == 0) &&
                   ((N - n0) >= BN || (N - n0) % 4 == 0);          // uniform
  if (!vec) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int col = n0 + wn * 64 + nt * 16 + col_l;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * 64 + mt * 16 + row_l + r;
          if (row >= M) continue;
          float* c = dst + (size_t)row * ldd + col;
          float v = acc[mt][nt][r] + bv;
          if (beta != 0.f) v += *c;
          *c = v;
        }
      }
    return;
  }
  const int c4 = tid & 31, rr = tid >> 5;
  const int col = n0 + c4 * 4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && col < N) { bv.x = bias[col]; bv.y = bias[col + 1]; bv.z = bias[col + 2]; bv.w = bias[col + 3]; }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            smem[(mt * 16 + row_l + r) * LDT + wn * 64 + nt * 16 + col_l] = acc[mt][nt][r];
    }
    __syncthreads();
    if (col < N) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + half * 64 + rr + 8 * i;
        if (row < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[(rr + 8 * i) * LDT + c4 * 4]);
          float* c = dst + (size_t)row * ldd + col;
          v += bv;
          if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(c);
          *reinterpret_cast<f32x4*>(c) = v;
        }
      }
    }
    __syncthreads();
  }
}

template <bool A_KCONTIG, bool B_KCONTIG, bool DMA>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // GEMM_SMEM_BYTES
  // layout: see gemm_kloop

  // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs
  // (private L2 each); give each XCD a contiguous band of N-tiles of one
  // M-row-band so neighbours share the A panel in that XCD's L2.
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  // work item = (K slice z, tile); a capped grid (max_workgroups) walks the items
  // persistently so an overlapped GEMM can be confined to a few CUs
  for (int wi = blockIdx.x; wi < nwg * g.splitk; wi += gridDim.x) {
  int bid = wi % nwg;
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = wi / nwg;
  const float* Ap = g.A;
  const float* Bp = g.B;
  const int lda = g.lda, ldb = g.ldb;
  const int kbeg = z * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (DMA) gemm_kloop_dma<A_KCONTIG, B_KCONTIG>(Ap, lda, Bp, ldb, g.M, g.N, m0, n0, kbeg, kend, smem, acc, 0);
  else gemm_kloop<A_KCONTIG, B_KCONTIG>(Ap, lda, Bp, ldb, g.M, g.N, m0, n0, kbeg, kend, smem, acc, 0);

  if (g.splitk == 1) store_tile(acc, smem, g.C, g.ldc, m0, n0, g.M, g.N, g.bias, g.beta);
  else store_tile(acc, smem, g.slab + (size_t)z * g.M * g.N, g.N, m0, n0, g.M, g.N, nullptr, 0.f);
  }  // work-item loop (the k-loop above ends with a barrier, so LDS reuse is safe)
}

__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ slab,
                                          float* __restrict__ C,
                                          const float* __restrict__ bias, int M,
                                          int N, int ldc, int splitk, float beta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)M * N;
  if (i >= total) return;
  const int row = (int)(i / N), col = (int)(i % N);
  float s = 0.f;
  for (int z = 0; z < splitk; ++z) s += slab[(size_t)z * total + i];
  if (bias) s += bias[col];
  float* c = C + (size_t)row * ldc + col;
  if (beta != 0.f) s += *c;
  *c = s;
}

// dynamic LDS above 64 KB needs the attribute (BK = 32: 67.6 KB); set once per kernel
template <typename KernelT>
static void gemm_allow_lds(KernelT k) {
  if (GEMM_SMEM_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GEMM_SMEM_BYTES);
}
#define GEMM_FOR_ALL_VARIANTS(K_, F_)                                                     \
  F_((K_<true, false, false>)); F_((K_<true, true, false>)); F_((K_<false, false, false>)); \
  F_((K_<false, true, false>)); F_((K_<true, false, true>)); F_((K_<true, true, true>));    \
  F_((K_<false, false, true>)); F_((K_<false, true, true>))
static void gemm_init_once() {
  // function-local static: initialised exactly once, thread-safe (C++11)
  static const bool done = [] { GEMM_FOR_ALL_VARIANTS(gemm_f32_kernel, gemm_allow_lds); return true; }();
  (void)done;
}

// DMA staging (gemm_kloop_dma) needs 16-byte aligned operands with ld % 4 == 0, and K % 4 == 0
// for an operand that is contiguous along k.  DANET_GEMM_DMA is a bit mask: 1 = tile-per-workgroup
// launches, 2 = stream-K / grouped launches that have the GPU to themselves, 4 = grouped launches
// capped to one workgroup per CU (the ones that run beside a BPTT kernel).
// Default 3: beside a BPTT kernel the DMA group is 12 % faster but costs that kernel more than
// it saves (cfg 2: BPTT 450 -> 478 us per launch, 3.37 -> 3.50 ms per step).  Read per launch
// (tests flip it).
static bool dma_env(int bit) { return (danet_opt(OPT_GEMM_DMA) & bit) != 0; }
static bool dma_operand_ok(const float* p, int ld, bool kcontig, int K) {
  return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0) && (!kcontig || K % 4 == 0);
}

static int choose_splitk(int M, int N, int K) {
  const int tiles = cdiv(M, BM) * cdiv(N, BN);
  if (tiles >= 192 || K < 512) return 1;
  // Fill the 256 CUs ~twice over: 160-tile products (dX, dYc: N = 600) gain 1.4-1.9x
  // from 4 slices; the cost is one M*N slab write + read per slice in the reduce
  // kernel, so the slice count is capped (it was 8% of the train step at s = 18).
  const int target = danet_opt(OPT_SPLITK_TARGET) > 0 ? danet_opt(OPT_SPLITK_TARGET) : 512;
  int s = cdiv(target, tiles);
  const int maxs = K / 256;
  if (s > maxs) s = maxs;
  if (s > 8) s = 8;
  return s < 1 ? 1 : s;
}

size_t dn_ws_gemm(int M, int N, int K) {
  const int s = choose_splitk(M, N, K);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// the kernels address an operand through one 32-bit buffer view per tile
static bool operand_fits(int rows, int cols, int ld) {
  return ((size_t)(rows - 1) * ld + cols) * sizeof(float) < ((size_t)1 << 31);
}

static int gemm_launch(hipStream_t stream, int transA, int transB, int M, int N,
                       int K, const float* A, int lda, const float* B, int ldb,
                       float* C, int ldc, const float* bias, float beta, void* ws,
                       size_t ws_bytes, int max_workgroups) {
  gemm_init_once();
  DANET_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: non-positive shape %d %d %d", M, N, K);
  DANET_CHECK_ARG(A && B && C, "gemm: null operand");
  DANET_CHECK_ARG(beta == 0.f || beta == 1.f, "gemm: beta must be 0 or 1");
  DANET_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
                  "gemm: leading dimension too small");
  DANET_CHECK_ARG(operand_fits(transA ? K : M, transA ? M : K, lda) &&
                  operand_fits(transB ? N : K, transB ? K : N, ldb),
                  "gemm: an operand spans 2 GiB or more");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.beta = beta;
  const int stot = choose_splitk(M, N, K);
  const int kchunk = cdiv(cdiv(K, stot), BK) * BK;
  const int splitk = cdiv(K, kchunk);
  g.splitk = splitk; g.kchunk = kchunk; g.slab = nullptr;
  if (splitk > 1) {
    const size_t need = (size_t)splitk * M * N * sizeof(float);
    if (!ws || ws_bytes < need) {
      danet_set_error("gemm: workspace %zu < %zu", ws_bytes, need);
      return DANET_ERR_WORKSPACE;
    }
    g.slab = (float*)ws;
  }
  int nblocks = cdiv(M, BM) * cdiv(N, BN) * splitk;
  if (max_workgroups > 0 && nblocks > max_workgroups) nblocks = max_workgroups;
  dim3 grid(nblocks, 1, 1), block(256);
  const bool ak = !transA, bk = (transB != 0);
  const bool dma = dma_env(1) && dma_operand_ok(A, lda, ak, K) && dma_operand_ok(B, ldb, bk, K);
#define GEMM_LAUNCH(D_)                                                                          \
  do {                                                                                           \
    if (ak && !bk) gemm_f32_kernel<true, false, D_><<<grid, block, GEMM_SMEM_BYTES, stream>>>(g);        \
    else if (ak && bk) gemm_f32_kernel<true, true, D_><<<grid, block, GEMM_SMEM_BYTES, stream>>>(g);     \
    else if (!ak && !bk) gemm_f32_kernel<false, false, D_><<<grid, block, GEMM_SMEM_BYTES, stream>>>(g); \
    else gemm_f32_kernel<false, true, D_><<<grid, block, GEMM_SMEM_BYTES, stream>>>(g);                  \
  } while (0)
  if (dma) GEMM_LAUNCH(true); else GEMM_LAUNCH(false);
#undef GEMM_LAUNCH
  DANET_CHECK_LAUNCH();
  if (splitk > 1) {
    const int64_t total = (int64_t)M * N;
    gemm_splitk_reduce_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, stream>>>(
        g.slab, C, bias, M, N, ldc, splitk, beta);
    DANET_CHECK_LAUNCH();
  }
  return DANET_OK;
}

extern "C" int danet_gemm_f32(danet_stream_t stream_, int transA, int transB,
                              int M, int N, int K, const float* A, int lda,
                              const float* B, int ldb, float* C, int ldc,
                              const float* bias, float beta, void* ws,
                              size_t ws_bytes, int max_workgroups) {
  return gemm_launch((hipStream_t)stream_, transA, transB, M, N, K, A, lda, B, ldb,
                     C, ldc, bias, beta, ws, ws_bytes, max_workgroups);
}

// ---------------------------------------------------------------------------
// stream-K scheduling
// ---------------------------------------------------------------------------
// Alternative schedule for a product that has the GPU to itself (danet_gemm_f32_streamk):
// one tile per workgroup leaves the last round of the 256 CUs mostly empty and
// split-K needs a second kernel.  Here the launch is G persistent workgroups
// (one per CU).  The tiles are dealt to the 8 XCDs in contiguous bands (workgroup b
// runs on XCD b % 8: a band's A/B panels stay in that XCD's L2); within a band the
// (tile, k-iteration) space is cut into G/8 EQUAL contiguous ranges, one per
// workgroup, so every workgroup does the same number of MFMAs whatever the shape.
// A tile whose k-range is cut is finished deterministically inside the kernel:
//   - the workgroup holding a tile's LAST k-segment is its owner; the others
//     publish their partial accumulators (write-through 16-byte stores, drained,
//     then a flag carrying this launch's sequence number) and go on;
//   - every workgroup walks its range from the END backwards, so what it owes a
//     higher-numbered owner is computed first, and the tile it owns itself is
//     computed last -- by then its contributors (all lower-numbered, dispatched
//     earlier, and never waiting before they publish) are long done;
//   - the owner adds the partials to its own segment in ascending workgroup order.
// Same G + same shape => same summation order => bit-reproducible.
// Measured (tools/bench_gemm.py, cfg-2 shapes): dYc 240 -> 137 us, dX 117 -> 78 us,
// gx 106 -> 79 us alone on the GPU; but 2-3 such launches sharing the CUs with each
// other and with a persistent LSTM kernel are SLOWER than the tile-per-workgroup
// launches above (static equal ranges lose to the hardware's dynamic dispatch), so
// the caller opts in per product.
#define SK_MAX_PROBLEMS 6
struct SkProblem {
  const float* A; const float* B; float* C; const float* bias;
  const float* A2; const float* B2;   // optional second operand pair (K-concatenated product)
  int lda2, ldb2;
  int M, N, lda, ldb, ldc;
  int tiles_n;     // N tiles of this problem
  int tile0;       // first global tile index of this problem
  float beta;
};
struct SkArgs {
  SkProblem p[SK_MAX_PROBLEMS];   // a group shares K and the transpose flags
  int nprob, K, nk, tiles;        // nk = k-iterations (of BK) per tile; tiles = total
  int K2, nk1;                    // K-concatenated products: k-iterations [0, nk1) take pair 1 (K),
                                  // [nk1, nk) pair 2 (K2); nk1 == nk without a second pair
  float* slab;          // [G][16][256][4] partial accumulators
  unsigned* flags;      // [G] launch sequence number when slab[w] is valid
  unsigned seq;
  int yield;            // > 0: yield x 64 clocks asleep after every k-tile (32 MFMAs per wave): a group that
                        // runs UNDER a latency-bound recurrent kernel leaves issue slots to it
};

#define SK_SPIN_LIMIT (1u << 22)

// MF16: the 16x16x4 k-loop (register staging only; see gemm_kloop16) -- the accumulators are then
// 4x4 tiles of f32x4; the slab holds a thread's 64 values either way.
template <bool A_KCONTIG, bool B_KCONTIG, bool DMA, bool MF16 = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_sk_kernel(SkArgs sk) {
  static_assert(!(DMA && MF16), "the 16x16x4 k-loop has no DMA staging form");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // GEMM_SMEM_BYTES
  // layout: see gemm_kloop
  const int tid = threadIdx.x;

  const int G8 = gridDim.x >> 3;                 // workgroups per XCD band
  const int band = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tiles = sk.tiles;
  const int tb0 = (int)((int64_t)tiles * band / 8), tb1 = (int)((int64_t)tiles * (band + 1) / 8);
  // HYBRID schedule: every workgroup of the band first owns F = floor(band tiles / G8) WHOLE
  // tiles (data-parallel: nothing is cut, nothing to fix up); only the R = band tiles - F * G8
  // tiles of the ragged last round are cut stream-K style into G8 equal (tile, k-iteration)
  // ranges.  With fewer tiles than workgroups (weight-gradient groups) F = 0 and the launch is
  // pure stream-K as before; with a multiple of G8 tiles nothing is cut at all.
  const int F = (tb1 - tb0) / G8;
  const int rb0 = tb0 + F * G8;                    // first remainder tile of the band
  const int64_t I = (int64_t)(tb1 - rb0) * sk.nk;  // (tile, k-iteration) items of the remainder
  const int64_t lo = I * j / G8, hi = I * (j + 1) / G8;

  const unsigned slab_bytes = gridDim.x * 65536u;
  const __amdgpu_buffer_rsrc_t sres =
      __builtin_amdgcn_make_buffer_rsrc(sk.slab, 0, (int)slab_bytes, 0x00020000);

  // phase 0: the remainder range, walked from its END backwards (what is owed to a higher-
  // numbered owner first, the own cut tile last); phases 1..F: the whole tiles
  int64_t it = hi;
  int fdone = 0;
  for (;;) {
    int tile, kb, ke;
    int64_t tbase = 0;
    const bool rem = it > lo;
    if (rem) {
      const int tl = (int)((it - 1) / sk.nk);              // remainder-local tile
      tbase = (int64_t)tl * sk.nk;
      kb = (int)((lo > tbase ? lo : tbase) - tbase); ke = (int)(it - tbase);
      tile = rb0 + tl;
    } else {
      if (fdone >= F) break;
      tile = tb0 + j * F + fdone;
      kb = 0; ke = sk.nk;
      ++fdone;
    }
    int pi = sk.nprob - 1;
    while (pi > 0 && tile < sk.p[pi].tile0) --pi;
    const SkProblem& pr = sk.p[pi];
    GemmArgs g;
    g.A = pr.A; g.B = pr.B; g.C = pr.C; g.bias = pr.bias;
    g.M = pr.M; g.N = pr.N; g.K = sk.K; g.lda = pr.lda; g.ldb = pr.ldb; g.ldc = pr.ldc;
    g.beta = pr.beta;
    g.slab = nullptr; g.splitk = 1; g.kchunk = sk.K;
    const int ptile = tile - pr.tile0;
    const int tm = ptile / pr.tiles_n, tn = ptile % pr.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x16 acc[2][2];
    f32x4 acc16[4][4];
    if constexpr (MF16) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc16[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    }

    if (kb < sk.nk1) {       // the part of the segment inside the first operand pair
      const int ke1 = ke < sk.nk1 ? ke : sk.nk1;
      if constexpr (MF16) gemm_kloop16<A_KCONTIG, B_KCONTIG>(g.A, g.lda, g.B, g.ldb, g.M, g.N, m0, n0,
                                                             kb * BK, min(g.K, ke1 * BK), smem, acc16, sk.yield);
      else if (DMA) gemm_kloop_dma<A_KCONTIG, B_KCONTIG>(g.A, g.lda, g.B, g.ldb, g.M, g.N, m0, n0,
                                                         kb * BK, min(g.K, ke1 * BK), smem, acc, sk.yield);
      else gemm_kloop<A_KCONTIG, B_KCONTIG>(g.A, g.lda, g.B, g.ldb, g.M, g.N, m0, n0,
                                            kb * BK, min(g.K, ke1 * BK), smem, acc, sk.yield);
    }
    if (ke > sk.nk1) {       // ... and inside the second (same accumulators)
      const int kb2 = (kb > sk.nk1 ? kb : sk.nk1) - sk.nk1, ke2 = ke - sk.nk1;
      if constexpr (MF16) gemm_kloop16<A_KCONTIG, B_KCONTIG>(pr.A2, pr.lda2, pr.B2, pr.ldb2, g.M, g.N, m0, n0,
                                                             kb2 * BK, min(sk.K2, ke2 * BK), smem, acc16, sk.yield);
      else if (DMA) gemm_kloop_dma<A_KCONTIG, B_KCONTIG>(pr.A2, pr.lda2, pr.B2, pr.ldb2, g.M, g.N, m0, n0,
                                                         kb2 * BK, min(sk.K2, ke2 * BK), smem, acc, sk.yield);
      else gemm_kloop<A_KCONTIG, B_KCONTIG>(pr.A2, pr.lda2, pr.B2, pr.ldb2, g.M, g.N, m0, n0,
                                            kb2 * BK, min(sk.K2, ke2 * BK), smem, acc, sk.yield);
    }

    if (ke < sk.nk) {
      // contributor: the tile's later k-segments belong to higher workgroups
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        f32x4 v;
        if constexpr (MF16) {
          v = acc16[q >> 2][q & 3];
        } else {
          v.x = acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 0];
          v.y = acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 1];
          v.z = acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 2];
          v.w = acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 3];
        }
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(v4u, v), sres,
            (unsigned)(((blockIdx.x * 16u + q) * 256u + tid) * 16u), 0, 16 /*sc1*/);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(&sk.flags[blockIdx.x], sk.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (kb > 0) {
        // owner of a cut tile: add to the own segment the partials of the workgroups
        // that hold k-iterations [0, kb) of it, in ascending workgroup order
        const int jf = (int)((tbase * G8) / I);   // first candidate range
        for (int jc = jf; jc < j; ++jc) {
          const int64_t clo = I * jc / G8, chi = I * (jc + 1) / G8;
          if (chi == clo || chi <= tbase || clo >= tbase + kb) continue;   // no share of this tile
          const unsigned wsrc = (unsigned)(band + 8 * jc);
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&sk.flags[wsrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                   sk.seq) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > SK_SPIN_LIMIT) __builtin_trap();   // contributor never ran: fail loudly
            }
          }
          __syncthreads();
#pragma unroll
          for (int q0 = 0; q0 < 16; q0 += 4) {     // 4 loads in flight, 16 temporaries
            v4u raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              raw[u] = __builtin_amdgcn_raw_buffer_load_b128(
                  sres, (unsigned)(((wsrc * 16u + q0 + u) * 256u + tid) * 16u), 0, 16 /*sc1*/);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = q0 + u;
              const f32x4 v = __builtin_bit_cast(f32x4, raw[u]);
              if constexpr (MF16) {
                acc16[q >> 2][q & 3] += v;
              } else {
                acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 0] += v.x;
                acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 1] += v.y;
                acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 2] += v.z;
                acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 3] += v.w;
              }
            }
          }
        }
      }
      if constexpr (MF16) store_tile16(acc16, smem, g.C, g.ldc, m0, n0, g.M, g.N, g.bias, g.beta);
      else store_tile(acc, smem, g.C, g.ldc, m0, n0, g.M, g.N, g.bias, g.beta);
    }
    if (rem) it = tbase + kb;
  }
}

// Workgroups per launch (a multiple of 8, at most DANET_GEMM_WGS = 512 = two per CU):
//   - short-K products with at least a tile per CU get one workgroup per tile (nothing
//     is cut, nothing to fix up);
//   - otherwise at most DANET_GEMM_MAXSPLIT (8) workgroups share a tile, so the owner's
//     serial fix-up stays short; few-tile / long-K products (weight gradients) then run
//     on fewer, longer workgroups -- they are overlapped with other kernels anyway.
static int sk_grid(int tiles, int nk, int max_workgroups) {
  int gmax = danet_opt(OPT_GEMM_WGS) & ~7;
  if (gmax < 8) gmax = 8;
  if (gmax > 1024) gmax = 1024;
  int maxsplit = danet_opt(OPT_GEMM_MAXSPLIT);
  if (maxsplit < 1) maxsplit = 1;
  int lim = gmax;
  if (max_workgroups > 0 && max_workgroups < lim) lim = max_workgroups & ~7;
  if (lim < 8) lim = 8;
  int64_t gsz;
  if (tiles >= 256 && (int64_t)tiles * nk < (int64_t)12 * lim) gsz = cdiv(tiles, 8) * 8;
  else gsz = (int64_t)cdiv(tiles * maxsplit, 8) * 8;
  if (gsz > lim) gsz = lim;
  return (int)gsz;
}
#define SK_MAX_GRID 1024
#define SK_HEADER (SK_MAX_GRID * sizeof(unsigned))

size_t dn_ws_gemm_streamk(int M, int N, int K) {
  (void)M; (void)N; (void)K;
  return SK_HEADER + (size_t)SK_MAX_GRID * 65536;   // flags + one partial tile per workgroup
}

// Event plumbing for hosts that fork side streams behind a stream-K launch: the event given to
// danet_next_launch_events is attached to the NEXT stream-K launch of the calling host thread
// (consumed by it); danet_event_* wrap the runtime's calls so that a ctypes host needs no second
// library handle.
static thread_local hipEvent_t g_start_event = nullptr;
static thread_local hipEvent_t g_stop_event = nullptr;
extern "C" int danet_next_launch_events(void* start, void* stop) {
  g_start_event = (hipEvent_t)start;
  g_stop_event = (hipEvent_t)stop;
  return DANET_OK;
}
void dn_take_launch_events(hipEvent_t* start, hipEvent_t* stop) {   // (lstm.hip)
  *start = g_start_event; *stop = g_stop_event;
  g_start_event = nullptr; g_stop_event = nullptr;
}
hipEvent_t dn_take_stop_event() {            // (gemm_x6.hip; a start event is dropped)
  hipEvent_t a, e;
  dn_take_launch_events(&a, &e);
  return e;
}
extern "C" int danet_event_create(void** event) {
  DANET_CHECK_ARG(event != nullptr, "event_create: null pointer");
  hipEvent_t e = nullptr;
  DANET_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = (void*)e;
  return DANET_OK;
}
extern "C" int danet_event_destroy(void* event) {
  if (event) DANET_CHECK_HIP(hipEventDestroy((hipEvent_t)event));
  return DANET_OK;
}
extern "C" int danet_stream_wait_event(danet_stream_t stream, void* event) {
  DANET_CHECK_ARG(event != nullptr, "stream_wait_event: null event");
  DANET_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return DANET_OK;
}

// second: optional per-problem second operand pairs {A2, lda2, B2, ldb2} contracted over K2
struct SkSecond { const float* A2; int lda2; const float* B2; int ldb2; };
static int sk_launch(danet_stream_t stream_, int transA, int transB, int K, int K2, int nprob,
                     const danet_gemm_problem_t* probs, const SkSecond* second,
                     int max_workgroups, void* ws, size_t ws_bytes) {
  // flag value of the next launch: process-wide, atomically incremented, so launches from any
  // number of host threads get distinct values (a workspace only ever holds earlier ones)
  static std::atomic<unsigned> launch_seq{0x5eed0000u};
  static const bool lds_ok = [] { GEMM_FOR_ALL_VARIANTS(gemm_f32_sk_kernel, gemm_allow_lds); return true; }();
  (void)lds_ok;
  // a caller-supplied event (danet_next_launch_events) rides on this launch's own dispatch
  // packet instead of a separate hipEventRecord behind it (tools/csrc/event_gap.hip: the record costs
  // the stream 4.4 us before its next kernel, the attached event 1.1 us).  Consumed HERE, before
  // any check can return: a rejected call must not leave it armed for an unrelated later launch.
  hipEvent_t stop = dn_take_stop_event();
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(probs && nprob >= 1 && nprob <= SK_MAX_PROBLEMS, "gemm group: 1..%d problems",
                  SK_MAX_PROBLEMS);
  DANET_CHECK_ARG(K > 0 && K2 >= 0, "gemm: non-positive K %d", K);
  DANET_CHECK_ARG(K2 == 0 || (second && K % BK == 0),
                  "gemm: a K-concatenated stream-K product needs K1 %% %d == 0", BK);
  SkArgs sk;
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const danet_gemm_problem_t& q = probs[i];
    DANET_CHECK_ARG(q.M > 0 && q.N > 0, "gemm: non-positive shape %d %d", q.M, q.N);
    DANET_CHECK_ARG(q.A && q.B && q.C, "gemm: null operand");
    DANET_CHECK_ARG(q.beta == 0.f || q.beta == 1.f, "gemm: beta must be 0 or 1");
    DANET_CHECK_ARG(q.lda >= (transA ? q.M : K) && q.ldb >= (transB ? K : q.N) && q.ldc >= q.N,
                    "gemm: leading dimension too small");
    DANET_CHECK_ARG(operand_fits(transA ? K : q.M, transA ? q.M : K, q.lda) &&
                    operand_fits(transB ? q.N : K, transB ? K : q.N, q.ldb),
                    "gemm: an operand spans 2 GiB or more");
    SkProblem& p = sk.p[i];
    p.A = q.A; p.B = q.B; p.C = q.C; p.bias = q.bias;
    p.M = q.M; p.N = q.N; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.beta = q.beta;
    p.A2 = nullptr; p.B2 = nullptr; p.lda2 = p.ldb2 = 0;
    if (K2 > 0) {
      const SkSecond& e = second[i];
      DANET_CHECK_ARG(e.A2 && e.B2 && e.lda2 >= (transA ? q.M : K2) && e.ldb2 >= (transB ? K2 : q.N),
                      "gemm: bad second operand pair");
      DANET_CHECK_ARG(operand_fits(transA ? K2 : q.M, transA ? q.M : K2, e.lda2) &&
                      operand_fits(transB ? q.N : K2, transB ? K2 : q.N, e.ldb2),
                      "gemm: an operand spans 2 GiB or more");
      p.A2 = e.A2; p.B2 = e.B2; p.lda2 = e.lda2; p.ldb2 = e.ldb2;
    }
    p.tiles_n = cdiv(q.N, BN);
    p.tile0 = tiles;
    tiles += cdiv(q.M, BM) * p.tiles_n;
  }
  for (int i = nprob; i < SK_MAX_PROBLEMS; ++i) sk.p[i] = sk.p[0];
  sk.nprob = nprob; sk.K = K; sk.K2 = K2; sk.nk1 = cdiv(K, BK); sk.nk = sk.nk1 + cdiv(K2, BK);
  sk.tiles = tiles;
  const int gsz = sk_grid(tiles, sk.nk, max_workgroups);
  const size_t need = SK_HEADER + (size_t)gsz * 65536;
  if (!ws || ws_bytes < need || ((uintptr_t)ws & 15) != 0) {
    danet_set_error("gemm: workspace %zu < %zu (or not 16-B aligned)", ws_bytes, need);
    return DANET_ERR_WORKSPACE;
  }
  sk.flags = (unsigned*)ws;
  sk.slab = (float*)((char*)ws + SK_HEADER);
  sk.seq = launch_seq.fetch_add(1u, std::memory_order_relaxed) + 1u;
  // DANET_GEMM_YIELD=n: capped (overlapped) group launches sleep n*64 clocks after every k-tile.
  // The recurrent kernel beside such a group slows down with the group's MFMA duty
  // (tools/contention_probe.py: +58 us per BPTT launch beside a 60 % burner, starvation beside a
  // 100 % one) and the group has slack: it only has to finish before that kernel does.  Measured
  // at cfg 2: 0 -> 3.338, 8 -> 3.335, 16 -> 3.316, 24 -> 3.319, 32 -> 3.323, 48 -> 3.335 ms per step.
  sk.yield = (max_workgroups > 0 && max_workgroups <= 256) ? danet_opt(OPT_GEMM_YIELD) : 0;
  dim3 grid(gsz, 1, 1), block(256);
  const bool ak = !transA, bk = (transB != 0);
  bool dma = dma_env((max_workgroups > 0 && max_workgroups <= 256) ? 4 : 2);
  for (int i = 0; i < nprob; ++i) {
    dma = dma && dma_operand_ok(probs[i].A, probs[i].lda, ak, K) && dma_operand_ok(probs[i].B, probs[i].ldb, bk, K);
    if (K2 > 0)
      dma = dma && dma_operand_ok(second[i].A2, second[i].lda2, ak, K2) &&
            dma_operand_ok(second[i].B2, second[i].ldb2, bk, K2);
  }
#define SK_GO(K_)                                                                                   \
  do {                                                                                             \
    if (stop) hipExtLaunchKernelGGL((K_), grid, block, GEMM_SMEM_BYTES, stream, nullptr, stop, 0, sk); \
    else (K_)<<<grid, block, GEMM_SMEM_BYTES, stream>>>(sk);                                        \
  } while (0)
#define SK_LAUNCH(D_)                                                                              \
  do {                                                                                             \
    if (ak && !bk) SK_GO((gemm_f32_sk_kernel<true, false, D_>));                                    \
    else if (ak && bk) SK_GO((gemm_f32_sk_kernel<true, true, D_>));                                 \
    else if (!ak && !bk) SK_GO((gemm_f32_sk_kernel<false, false, D_>));                             \
    else SK_GO((gemm_f32_sk_kernel<false, true, D_>));                                              \
  } while (0)
  // capped groups (they share every CU with a recurrent kernel): the short-instruction k-loop
  const bool mf16 = !dma && sk.yield > 0 && danet_opt(OPT_GEMM_MFMA16) == 1;
  if (mf16) {
    static const bool lds16_ok = [] {
      gemm_allow_lds((const void*)gemm_f32_sk_kernel<true, false, false, true>);
      gemm_allow_lds((const void*)gemm_f32_sk_kernel<true, true, false, true>);
      gemm_allow_lds((const void*)gemm_f32_sk_kernel<false, false, false, true>);
      gemm_allow_lds((const void*)gemm_f32_sk_kernel<false, true, false, true>);
      return true; }();
    (void)lds16_ok;
    if (ak && !bk) SK_GO((gemm_f32_sk_kernel<true, false, false, true>));
    else if (ak && bk) SK_GO((gemm_f32_sk_kernel<true, true, false, true>));
    else if (!ak && !bk) SK_GO((gemm_f32_sk_kernel<false, false, false, true>));
    else SK_GO((gemm_f32_sk_kernel<false, true, false, true>));
  } else if (dma) SK_LAUNCH(true); else SK_LAUNCH(false);
#undef SK_LAUNCH
#undef SK_GO
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_gemm_f32_streamk_grouped(danet_stream_t stream_, int transA, int transB,
                                              int K, int nprob, const danet_gemm_problem_t* probs,
                                              int max_workgroups, void* ws, size_t ws_bytes) {
  return sk_launch(stream_, transA, transB, K, 0, nprob, probs, nullptr, max_workgroups, ws, ws_bytes);
}

// C = op(A1) op(B1) + op(A2) op(B2) (+bias) (+beta C) on the hybrid stream-K schedule: the two
// operand pairs are one K-concatenated contraction (k-iterations of pair 1, then of pair 2,
// continuing the same accumulators), so there are no per-pair slabs and no reduce kernel.
// K1 must be a multiple of 16.  Workspace: dn_ws_gemm_streamk.
extern "C" int danet_gemm_f32_streamk_kcat(danet_stream_t stream_, int transA, int transB, int M, int N,
                                           int K1, const float* A1, int lda1, const float* B1, int ldb1,
                                           int K2, const float* A2, int lda2, const float* B2, int ldb2,
                                           float* C, int ldc, const float* bias, float beta,
                                           void* ws, size_t ws_bytes) {
  if (K2 <= 0) (void)dn_take_stop_event();   // (an armed event dies with the rejected call)
  DANET_CHECK_ARG(K2 > 0, "gemm_streamk_kcat: K2 must be positive");
  danet_gemm_problem_t q;
  q.A = A1; q.lda = lda1; q.B = B1; q.ldb = ldb1; q.C = C; q.ldc = ldc; q.M = M; q.N = N;
  q.bias = bias; q.beta = beta;
  SkSecond e;
  e.A2 = A2; e.lda2 = lda2; e.B2 = B2; e.ldb2 = ldb2;
  return sk_launch(stream_, transA, transB, K1, K2, 1, &q, &e, 0, ws, ws_bytes);
}

extern "C" int danet_gemm_f32_streamk(danet_stream_t stream_, int transA, int transB,
                                      int M, int N, int K, const float* A, int lda,
                                      const float* B, int ldb, float* C, int ldc,
                                      const float* bias, float beta, void* ws,
                                      size_t ws_bytes) {
  danet_gemm_problem_t q;
  q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.C = C; q.ldc = ldc; q.M = M; q.N = N;
  q.bias = bias; q.beta = beta;
  return danet_gemm_f32_streamk_grouped(stream_, transA, transB, K, 1, &q, 0, ws, ws_bytes);
}
