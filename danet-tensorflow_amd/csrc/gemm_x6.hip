// fp32 products on the BF16 matrix cores (gfx950): C = A B^T (+ A2 B2^T) with fp32 operands and an
// fp32-accurate result, for the step's products whose B operand is a WEIGHT (the output projection,
// dYc, dX: DESIGN.md 3.2).
//
// Every fp32 value is EXACTLY hi + mid + lo with three bf16 pieces of 8 significant bits (hi = x
// rounded to 8 bits, mid = the remainder rounded to 8 bits, lo = what is left), and the six largest
// of the nine piece products are accumulated in fp32:
//     A B^T ~ Ahi Bhi + (Ahi Bmid + Amid Bhi) + (Ahi Blo + Alo Bhi + Amid Bmid)
// The dropped products are < 2^-25 |a||b| each: the result is as close to the float64 product as an
// fp32 FMA chain's (tools/bf16x_split_accuracy.py; tests/test_gpu_gemm_x6.py).
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32; six of them per 16 k
// replace eight fp32 instructions -> 2.7x the matrix-core throughput of gemm_f32.hip.
//
// Where the operands travel:
//   A [M][lda] fp32, K-contiguous (an activation): global -> registers -> split -> LDS piece images;
//   B (a weight): packed once per optimizer step by danet_gemm_pack_weights into the matrix
//     instruction's own operand layout -- [piece][32-column tile][k16 step][lane] x 16 bytes -- and
//     loaded by each wave straight from global memory / L2 into its fragment registers, one k-step
//     ahead.  B never touches LDS: the LDS pipe carries A only (0.25 fragment reads per matrix
//     instruction).
// Tile per workgroup 128 x 128 x 16, 4 waves as 2 x 2 with 2 x 2 MFMA tiles of 32 x 32 each, two
// workgroups per CU, three LDS stages; products with too few tiles for the GPU are cut along K into
// `splitk` slices whose partial tiles the workgroup that finishes a tile's LAST slice sums in slice
// order (tickets; deterministic, nobody waits).
// What bounds the k-loop on gfx950 is the vector ALU -- a vector and a matrix instruction of one SIMD
// do not overlap (tools/csrc/mfma_bf16_rate.hip) -- so it carries no address arithmetic on it: B
// fragments and staging loads take their k-tile in the instructions' SCALAR offset, the LDS stages
// are compile-time, and the split is 7 vector instructions per pair of values (csrc/common.h).
#include "common.h"
#include <hip/hip_ext.h>
#include <atomic>
#include "danet_hip.h"
#include "options.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define XBN 128
#define X6_OOB 0xfffffff0u               // voffset beyond num_records: the buffer load returns 0
#define XBK 16

// Geometry of the NT kernel for MI 32-row matrix tiles per wave along M: the workgroup tile is
// (64 MI) x 128 x 16, its 4 waves sit 2 x 2 with (32 MI) x 64 each -- MI x 2 accumulator tiles.
//   MI = 2: 128-row tiles, 256 registers per lane, two workgroups per CU
//   MI = 3 / 4: 192 / 256-row tiles, one workgroup per CU: a B fragment (global / L2 -> registers,
//     never shared between waves) feeds MI matrix instructions instead of two, which takes the
//     kernel off the L1 / L2 bandwidth bound of the 128-row tile (48 KB of B fragments per CU and
//     k-step at two workgroups: > 64 B / clock at full matrix rate)
template <int MI> struct X6Geo {
  static constexpr int BM = 64 * MI;
  static constexpr int IMG = BM * XBK * 2;      // bytes of one piece image: BM rows x 16 k x bf16
  static constexpr int STAGE = 3 * IMG;         // A hi / mid / lo
  static constexpr int LDC_T = XBN + 4;
  static constexpr int EPI = 32 * MI * LDC_T * 4;   // the epilogue's staging of one wave row's rows
  static constexpr int SMEM = 3 * STAGE > EPI ? 3 * STAGE : EPI;   // 3 stages
  static constexpr int NS = 12 * MI;            // matrix instructions per wave and k-step
  static constexpr int WG_PER_CU = MI == 2 ? 2 : 1;
};

struct X6Args {
  const float* A[2];
  const u32x4* Bp[2];                   // [pair] -> packed [3][NT32][KT][64] x 16 bytes
  int lda[2], K[2], KT[2];
  int NT32;
  float* C;                             // the result [M][ldc]
  const float* bias;                    // [N] added to every row, or null
  int M, N, ldc, npair, splitk;
  // Tiles [0, nwhole) are computed whole by the launch's first `nwhole` workgroups (a multiple of 8);
  // every other tile is cut into `splitk` K slices: partial tiles go to slab [splitk][M][N] and the
  // workgroup that finishes a tile's LAST slice sums the slices (in slice order) into C.
  // nwhole = 0: every tile sliced (or, splitk == 1, every tile whole) -- the schedules before round 6.
  int nwhole;
  float* slab;
  unsigned* tickets;                    // [tiles]: (launch sequence number << 8) | slices finished
  unsigned seq;
};

// x = hi + mid + lo exactly; hi and mid are x and the remainder ROUNDED to 8 significant bits (add
// half an ulp, truncate: |mid| <= 2^-9 |x|, |lo| <= 2^-17 |x|); lo has <= 8 significant bits
// left, its truncation is exact
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = (__float_as_uint(x) + 0x8000u) & 0xFFFF0000u;
  const float r = x - __uint_as_float(h);
  m = (__float_as_uint(r) + 0x8000u) & 0xFFFF0000u;
  l = __float_as_uint(r - __uint_as_float(m));
}
// (split_pair: csrc/common.h -- 7 vector instructions per pair of values)
__device__ __forceinline__ void split_pair_pk(f32x2v x, uint32_t& h, uint32_t& m, uint32_t& l) {
  split_pair(x[0], x[1], h, m, l);
}

// the high halves of two words as one word: [hi16(b) | hi16(a)]
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) {
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// byte offset of (row, 4-k group q in 0..3) inside a piece image: 32-byte rows, the two 16-byte
// chunks swapped when (row >> 2) is odd -> 8 consecutive rows of one chunk cover all banks
__device__ __forceinline__ int img_off(int row, int q) {
  const int c = (q >> 1) ^ ((row >> 2) & 1);
  return row * 32 + c * 16 + (q & 1) * 8;
}

__device__ __forceinline__ bf16x8 frag(const char* img, int row, int kb) {
  const int c = kb ^ ((row >> 2) & 1);
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + row * 32 + c * 16));
}

// Workgroups are dealt to the 8 XCDs round-robin; work item = the position this workgroup would have
// if every XCD took a CONTIGUOUS eighth of the item list instead: neighbours in the list (tiles that
// share an operand panel and a K slice) then share one XCD's L2 instead of fetching the panel 8 times.
__device__ __forceinline__ int x6_xcd_item(int b, int W) {
  const int x = b & 7, i = b >> 3, q = W >> 3, r = W & 7;
  return x * q + min(x, r) + i;
}

typedef unsigned v4u __attribute__((__vector_size__(16)));

// One k-step of the pipeline, branch-free (a conditionally executed load would make the compiler
// wait vmcnt(0) where the paths join): matrix instructions on the fragments of tile kt (registers),
// under them the B fragments and the A LDS fragments of tile kt + 1, the split + LDS store of tile
// kt + 2 (registers loaded two steps ago) and the global load of tile kt + 4 into the same registers.
struct X6Ctx {
  __amdgpu_buffer_rsrc_t rsa[2];
  __amdgpu_buffer_rsrc_t rsb[2];      // the packed weights: fragment f of a pair at byte f * 1024 + 16 lane
  const u32x4* bp[2];
  int lda[2], K[2], KT[2];
  int nk0, kt1, NT32, ntb, mrem;
  int srow, sq, lane, fi, kb, wm;
  // the k-loop's staging loads: byte offset of this thread's (row i, k group sq) inside pair `apair`
  // (X6_OOB for rows beyond M); the k-tile's offset travels in the instruction's scalar offset
  unsigned avo[4];
};
template <int MI>
__device__ __forceinline__ void x6_row_offsets(X6Ctx& c, int p) {
  const int lda = p ? c.lda[1] : c.lda[0];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = c.srow + 64 * i;
    c.avo[i] = r < c.mrem ? (unsigned)(r * lda + c.sq * 4) * 4u : X6_OOB;
  }
}

template <int MI>
__device__ __forceinline__ void x6_load_a(const X6Ctx& c, int kt, f32x4 (&R)[MI]) {
  const int p = kt >= c.nk0 ? 1 : 0;                          // uniform
  const int k = (kt - (p ? c.nk0 : 0)) * XBK + c.sq * 4;
  const int K = p ? c.K[1] : c.K[0], lda = p ? c.lda[1] : c.lda[0];
  const __amdgpu_buffer_rsrc_t rs = p ? c.rsa[1] : c.rsa[0];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = c.srow + 64 * i;
    const bool ok = r < c.mrem && k < K && kt < c.kt1;
    const unsigned off = (unsigned)(r * lda + k) * 4u;
    R[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : X6_OOB, 0, 0));
  }
}
template <int MI>
__device__ __forceinline__ void x6_store_a(char* base, const X6Ctx& c, const f32x4 (&R)[MI]) {
  constexpr int IMG = X6Geo<MI>::IMG;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int off = img_off(c.srow + 64 * i, c.sq);
    uint32_t h[2], m[2], l[2];
    split_pair(R[i][0], R[i][1], h[0], m[0], l[0]);
    split_pair(R[i][2], R[i][3], h[1], m[1], l[1]);
    *reinterpret_cast<u32x2*>(base + off) = (u32x2){h[0], h[1]};
    *reinterpret_cast<u32x2*>(base + IMG + off) = (u32x2){m[0], m[1]};
    *reinterpret_cast<u32x2*>(base + 2 * IMG + off) = (u32x2){l[0], l[1]};
  }
}
__device__ __forceinline__ void x6_load_b(const X6Ctx& c, int kt, u32x4 (&fb)[3][2]) {
  kt = min(kt, c.kt1 - 1);
  const int p = kt >= c.nk0 ? 1 : 0;
  const int ktl = kt - (p ? c.nk0 : 0);
  const int KT = p ? c.KT[1] : c.KT[0];
  const u32x4* __restrict__ bp = p ? c.bp[1] : c.bp[0];
#pragma unroll
  for (int pc = 0; pc < 3; ++pc)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fb[pc][j] = bp[((size_t)(pc * c.NT32 + c.ntb + j) * KT + ktl) * 64 + c.lane];
}
template <int MI>
__device__ __forceinline__ void x6_frags_a(const char* sb, const X6Ctx& c, bf16x8 (&fa)[3][MI]) {
#pragma unroll
  for (int pc = 0; pc < 3; ++pc)
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[pc][i] = frag(sb + pc * X6Geo<MI>::IMG, c.wm * 32 * MI + i * 32 + c.fi, c.kb);
}

// the six piece products, small terms first: (A piece, B piece)
__device__ constexpr int X6_PA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int X6_PB[6] = {0, 2, 1, 0, 1, 0};
__host__ __device__ constexpr int x6_pa(int t) { return t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0; }
__host__ __device__ constexpr int x6_pb(int t) { return t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0; }
#define X6_FENCE __builtin_amdgcn_sched_barrier(0);

// One k-step in 12 MI slots of one matrix instruction each (the 2 MI accumulators in turn, piece
// product by piece product); each slot carries a slice of the step's other work behind it and is
// fenced off from its neighbours, so that the vector-ALU, LDS and memory instructions issue in the
// shadow of the 32-cycle matrix instructions instead of in front of them.  MI = 2 (24 slots):
//   slots  0-5   B fragment loads of tile kt + 1 (global / L2 -> registers, 16 bytes per lane)
//   slots  6-11  A fragment reads of tile kt + 1 (LDS -> registers)                     [3 MI slots]
//   slots 12-19  split of the value pairs of tile kt + 2 this thread staged two steps ago (every
//                other slot)                                                            [4 MI slots]
//   slots 20-21  pack + LDS store of the rows' pieces                                   [MI slots]
//   slots 22-23  global loads of tile kt + 4 into the staging registers                 [MI slots]
template <int MI, int SRD, int SWR>
__device__ __forceinline__ void x6_step(X6Ctx& c, char* xsm, int kt, int srd, int swr, f32x16 (&acc)[MI][2],
                                        f32x4 (&R)[MI], const bf16x8 (&fac)[3][MI], bf16x8 (&fan)[3][MI],
                                        const u32x4 (&fbc)[3][2], u32x4 (&fbn)[3][2]) {
  typedef X6Geo<MI> G;
  constexpr int S_A = 6, S_SP = S_A + 3 * MI, S_W = S_SP + 4 * MI, S_L = S_W + MI, S_E = S_L + MI;
  static_assert(S_E <= G::NS, "slot plan");
  // uniform addressing of the next tile's B fragments: fragment (piece, column tile, k-tile) of a
  // pair is 1 KB at ((piece * NT32 + column tile) * KT + k-tile) * 1024 -- all of it in the load's
  // SCALAR offset, the lane's 16 bytes in a loop-invariant register: no vector instruction
  const int ktn = min(kt + 1, c.kt1 - 1);
  const int pn = ktn >= c.nk0 ? 1 : 0;
  const int ktl = ktn - (pn ? c.nk0 : 0);
  const int KTn = pn ? c.KT[1] : c.KT[0];
  const __amdgpu_buffer_rsrc_t rsbn = pn ? c.rsb[1] : c.rsb[0];
  const unsigned bo0 = (unsigned)(c.ntb * KTn + ktl) * 1024u, bop = (unsigned)(c.NT32 * KTn) * 1024u, boj = (unsigned)KTn * 1024u;
  const unsigned lane16 = (unsigned)c.lane * 16u;
  // (the LDS stages are compile-time constants -- the k-loop is unrolled over the 3 stages x 2
  // register sets -- so every LDS address is a loop-invariant register + an immediate)
  // SRD < 0: the stages are run-time values (the loop's tail)
  const char* srdp = xsm + (SRD >= 0 ? SRD : srd) * G::STAGE;
  char* swrp = xsm + (SRD >= 0 ? SWR : swr) * G::STAGE;
  // ... and of the staging loads of tile kt + 4: row offsets in registers (recomputed when the tile
  // belongs to the other pair), the k-tile in the scalar offset, validity of this thread's k group
  // (a ragged last tile, a tile beyond the slice) as one compare
  const int kta = kt + 4;
  const int pa = kta >= c.nk0 ? 1 : 0;
  if (kta == c.nk0) {                    // the first tile of the second pair (the asm keeps it a branch)
    asm volatile("" ::: "memory");
    x6_row_offsets<MI>(c, 1);
  }
  const int ktla = kta - (pa ? c.nk0 : 0);
  const int Ka = pa ? c.K[1] : c.K[0];
  const __amdgpu_buffer_rsrc_t rsaa = pa ? c.rsa[1] : c.rsa[0];
  const int klim = kta < c.kt1 ? Ka - ktla * XBK : 0;            // k values of the tile that exist
  const bool kok = c.sq * 4 < klim;
  const unsigned aso = kta < c.kt1 ? (unsigned)ktla * (XBK * 4u) : 0u;
  uint32_t h[MI][2], m[MI][2], l[MI][2];
#pragma unroll
  for (int S = 0; S < G::NS; ++S) {
    const int t = S / (2 * MI), i = (S >> 1) % MI, j = S & 1;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fac[x6_pa(t)][i], __builtin_bit_cast(bf16x8, fbc[x6_pb(t)][j]),
                                                        acc[i][j], 0, 0, 0);
    if (S < S_A) {
      fbn[S >> 1][S & 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rsbn, lane16, bo0 + (S >> 1) * bop + (S & 1) * boj, 0));
    } else if (S < S_SP) {
      const int a = S - S_A, pc = a / MI, ii = a % MI;
      fan[pc][ii] = frag(srdp + pc * G::IMG, c.wm * 32 * MI + ii * 32 + c.fi, c.kb);
    } else if (S < S_W) {
      const int a = S - S_SP;
      if ((a & 1) == 0) {
        const int I = a >> 2, HF = (a >> 1) & 1;
        split_pair_pk(HF ? __builtin_shufflevector(R[I], R[I], 2, 3) : __builtin_shufflevector(R[I], R[I], 0, 1), h[I][HF], m[I][HF], l[I][HF]);
      }
    } else if (S < S_L) {
      const int I = S - S_W;
      const int off = img_off(c.srow + 64 * I, c.sq);
      *reinterpret_cast<u32x2*>(swrp + off) = (u32x2){h[I][0], h[I][1]};
      *reinterpret_cast<u32x2*>(swrp + G::IMG + off) = (u32x2){m[I][0], m[I][1]};
      *reinterpret_cast<u32x2*>(swrp + 2 * G::IMG + off) = (u32x2){l[I][0], l[I][1]};
    } else if (S < S_E) {
      const int I = S - S_L;
      R[I] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsaa, kok ? c.avo[I] : X6_OOB, aso, 0));
    }
    X6_FENCE
  }
}
#define X6_STEP(SRD, SWR, R, FAC, FAN, FBC, FBN)                                                     \
  { x6_step<MI, SRD, SWR>(c, xsm, kt, srd, swr, acc, R, FAC, FAN, FBC, FBN);                         \
    __syncthreads();                                                                                 \
    ++kt; }
#define X6_STEP_RT(R, FAC, FAN, FBC, FBN)                                                            \
  { x6_step<MI, -1, -1>(c, xsm, kt, srd, swr, acc, R, FAC, FAN, FBC, FBN);                           \
    __syncthreads();                                                                                 \
    ++kt; srd = swr; swr = swr == 2 ? 0 : swr + 1; }

template <int MI>
__global__ __launch_bounds__(256, X6Geo<MI>::WG_PER_CU) void gemm_x6_nt_kernel(X6Args g) {
  typedef X6Geo<MI> G;
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + XBN - 1) / XBN;
  const int nt = ((g.M + G::BM - 1) / G::BM) * tiles_n;
  // HYBRID schedule (round 6): a product with a ragged last round of tiles (the projection: 672 tiles on
  // 512 slots) runs its first `nwhole` tiles whole and cuts only the remainder along K, so that the last
  // round is full and short instead of 31 % full and as long as the first
  int z, bid;
  bool sliced;                                          // (uniform)
  if ((int)blockIdx.x < g.nwhole) {
    bid = x6_xcd_item(blockIdx.x, g.nwhole); z = 0; sliced = false;
  } else {
    const int j = x6_xcd_item((int)blockIdx.x - g.nwhole, (int)gridDim.x - g.nwhole);
    const int nrem = nt - g.nwhole;
    z = j / nrem;                                       // K slice (slice-major: an XCD's items share the K range of
    bid = g.nwhole + j % nrem;                          // both operands; a tile's slices on ONE XCD measured 0.5 % slower)
    sliced = g.splitk > 1;
  }
  const int m0 = (bid / tiles_n) * G::BM, n0 = (bid % tiles_n) * XBN;

  const int nk0 = (g.K[0] + XBK - 1) / XBK;
  const int nkt = nk0 + (g.npair > 1 ? (g.K[1] + XBK - 1) / XBK : 0);
  const int per = sliced ? (nkt + g.splitk - 1) / g.splitk : nkt;
  const int kt0 = z * per, kt1 = min(nkt, kt0 + per);

  X6Ctx c;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int pp = p < g.npair ? p : 0;
    const float* base = g.A[pp] + (size_t)m0 * g.lda[pp];
    const float* end = g.A[pp] + (size_t)(g.M - 1) * g.lda[pp] + g.K[pp];
    c.rsa[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)((end - base) * 4), 0x00020000);
    c.bp[p] = g.Bp[pp]; c.lda[p] = g.lda[pp]; c.K[p] = g.K[pp]; c.KT[p] = g.KT[pp];
    c.rsb[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(g.Bp[pp]), 0, 3 * g.NT32 * g.KT[pp] * 1024, 0x00020000);
  }
  c.nk0 = nk0; c.kt1 = kt1; c.NT32 = g.NT32; c.ntb = n0 / 32 + wn * 2; c.mrem = g.M - m0;
  c.srow = tid >> 2; c.sq = tid & 3; c.lane = lane; c.fi = lane & 31; c.kb = lane >> 5; c.wm = wm;
  const int fi = c.fi, kb = c.kb;
  x6_row_offsets<MI>(c, kt0 + 4 >= nk0 ? 1 : 0);

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt0 < kt1) {
    // tile t lives in LDS stage (t - kt0) % 3.  Prologue: tiles kt0 and kt0 + 1 in LDS, kt0 + 2 and
    // kt0 + 3 in registers, the fragments of tile kt0 loaded.
    f32x4 R0[MI], R1[MI];
    u32x4 fbA[3][2], fbB[3][2];
    bf16x8 faA[3][MI], faB[3][MI];
    x6_load_a<MI>(c, kt0, R0); x6_load_a<MI>(c, kt0 + 1, R1);
    x6_load_b(c, kt0, fbA);
    x6_store_a<MI>(xsm, c, R0); x6_store_a<MI>(xsm + G::STAGE, c, R1);
    x6_load_a<MI>(c, kt0 + 2, R0); x6_load_a<MI>(c, kt0 + 3, R1);
    __syncthreads();
    x6_frags_a<MI>(xsm, c, faA);
    int kt = kt0, srd = 1, swr = 2;
    while (kt + 5 < kt1) {               // 3 stages x 2 register sets: back where it started
      X6_STEP(1, 2, R0, faA, faB, fbA, fbB)
      X6_STEP(2, 0, R1, faB, faA, fbB, fbA)
      X6_STEP(0, 1, R0, faA, faB, fbA, fbB)
      X6_STEP(1, 2, R1, faB, faA, fbB, fbA)
      X6_STEP(2, 0, R0, faA, faB, fbA, fbB)
      X6_STEP(0, 1, R1, faB, faA, fbB, fbA)
    }
    while (kt + 1 < kt1) {
      X6_STEP_RT(R0, faA, faB, fbA, fbB)
      X6_STEP_RT(R1, faB, faA, fbB, fbA)
    }
    if (kt < kt1) X6_STEP_RT(R0, faA, faB, fbA, fbB)
  }

  if (sliced) {
    // K slices (N % 4 == 0, 16-byte aligned C: the host's condition for slicing).  Every slice's
    // partial tile goes to its slab with write-through (sc1) stores; a ticket per tile counts the
    // slices that have landed, and the workgroup that draws the last ticket sums the slabs IN SLICE
    // ORDER (its own included: whoever is last, the same additions in the same order) + bias into
    // C.  Nobody ever waits: no assumption about dispatch order or co-residency.  The ticket
    // cells live in a fixed region at the start of a workspace that belongs to this entry point
    // (zero once, cleared by the last arriver); they carry the launch's sequence number while it
    // runs, so what an aborted launch left behind reads as "no slice yet".
    __shared__ unsigned x6_last;
    float* ct = reinterpret_cast<float*>(xsm);          // [32 MI][XBN + 4]
    constexpr int LDC_T = G::LDC_T;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        g.slab, 0, (int)((size_t)g.splitk * g.M * g.N * sizeof(float)), 0x00020000);
    const unsigned plane = (unsigned)g.M * (unsigned)g.N * 4u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb) * LDC_T + wn * 64 + j * 32 + fi] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4 * MI; ++it) {
        const int rr = (tid >> 5) + 8 * it, c4 = tid & 31;
        const int row = m0 + half * 32 * MI + rr, col = n0 + c4 * 4;
        const unsigned off = (row < g.M && col < g.N) ? (unsigned)z * plane + ((unsigned)row * g.N + col) * 4u : X6_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u*>(&ct[rr * LDC_T + c4 * 4]), srs, off, 0, 16 /*sc1*/);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned* cell = g.tickets + bid;
      const unsigned tag = g.seq << 8;
      unsigned old = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nw;
      do {
        nw = (old & 0xFFFFFF00u) == tag ? old + 1u : (tag | 1u);
      } while (!__hip_atomic_compare_exchange_strong(cell, &old, nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT));
      x6_last = (nw & 0xFFu) == (unsigned)g.splitk ? 1u : 0u;
    }
    __syncthreads();
    if (x6_last == 0u) return;
    const float* __restrict__ bias = g.bias;
    // (the loads of four row groups in flight together; a slice beyond splitk is an out-of-range offset that
    // returns zeros and is not added.  The modelled plans use <= 4 slices; a pinned plan with more takes the
    // dependent loop behind them)
    const int ns = g.splitk;
    constexpr int XS = 4;
#pragma unroll 1
    for (int it0 = 0; it0 < 8 * MI; it0 += 4) {
      f32x4 w[4][XS];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = (tid >> 5) + 8 * (it0 + u), c4 = tid & 31;
        const int row = m0 + rr, col = n0 + c4 * 4;
        ok[u] = row < g.M && col < g.N;
        const unsigned off = ((unsigned)row * g.N + col) * 4u;
#pragma unroll
        for (int zz = 0; zz < XS; ++zz)
          w[u][zz] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              srs, (ok[u] && zz < ns) ? (unsigned)zz * plane + off : X6_OOB, 0, 16 /*sc1*/));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = (tid >> 5) + 8 * (it0 + u), c4 = tid & 31;
        const int row = m0 + rr, col = n0 + c4 * 4;
        if (ok[u]) {
          f32x4 v = w[u][0];
#pragma unroll
          for (int zz = 1; zz < XS; ++zz)
            if (zz < ns) v += w[u][zz];               // (uniform)
          const unsigned off = ((unsigned)row * g.N + col) * 4u;
          for (int zz = XS; zz < ns; ++zz)
            v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)zz * plane + off, 0, 16 /*sc1*/));
          if (bias) v += (f32x4){bias[col], bias[col + 1], bias[col + 2], bias[col + 3]};
          *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = v;
        }
      }
    }
    // (every slice has arrived: nobody else touches this ticket in this launch)
    if (tid == 0) __hip_atomic_store(g.tickets + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // C/D layout 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The tile leaves
  // through the (now free) LDS, one wave row's 32 MI rows at a time, as 16-byte stores of full
  // 512-byte row segments when the destination allows it; 4-byte stores otherwise.
  float* __restrict__ dst = g.C;
  const float* __restrict__ bias = g.bias;
  const bool vec = (g.ldc % 4 == 0) && (((uintptr_t)g.C & 15) == 0) && (n0 + XBN <= g.N) &&
                   (!bias || ((uintptr_t)bias & 15) == 0);                                  // uniform
  if (vec) {
    float* ct = reinterpret_cast<float*>(xsm);          // [32 MI][XBN + 4]
    constexpr int LDC_T = G::LDC_T;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb) * LDC_T + wn * 64 + j * 32 + fi] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4 * MI; ++it) {
        const int rr = (tid >> 5) + 8 * it, c4 = tid & 31;
        const int row = m0 + half * 32 * MI + rr;
        if (row < g.M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&ct[rr * LDC_T + c4 * 4]);
          if (bias) v += *reinterpret_cast<const f32x4*>(bias + n0 + c4 * 4);
          *reinterpret_cast<f32x4*>(dst + (size_t)row * g.ldc + n0 + c4 * 4) = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + fi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        if (row < g.M && col < g.N) dst[(size_t)row * g.ldc + col] = acc[i][j][r] + (bias ? bias[col] : 0.f);
      }
    }
}

// ------------------------------------------------------------------ weights in operand layout
// out [3][NT32][KT][64] x 16 bytes: lane l of block (piece, nt, kt) holds the piece's bf16 values of
// B(n = 32 nt + l % 32, k = 16 kt + 8 (l / 32) + 0..7); B(n, k) = src[n * sn + k * sk]; zero padded.
// One launch packs every weight of the table: blockIdx.y = weight.
#define X6_MAX_PACK 16
struct X6PackJob { const float* src; u32x4* out; int64_t sn, sk; int N, K, NT32, KT; };
struct X6PackArgs { X6PackJob job[X6_MAX_PACK]; };

__global__ __launch_bounds__(256) void gemm_x6_pack_kernel(X6PackArgs a) {
  const X6PackJob& q = a.job[blockIdx.y];
  const int N = q.N, K = q.K, NT32 = q.NT32, KT = q.KT;
  const float* __restrict__ src = q.src;
  u32x4* __restrict__ out = q.out;
  const int64_t sn = q.sn, sk = q.sk;
  const bool kvec = sk == 1 && (sn & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  // (32-bit index arithmetic: the host checks NT32 * KT * 64 < 2^31 -- a 64-bit divide and modulo per
  // 8 values were a third of this kernel's instructions)
  const unsigned total = (unsigned)NT32 * (unsigned)KT * 64u;
  const size_t plane = (size_t)total;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const int l = (int)(idx & 63u);
    const unsigned t = idx >> 6;
    const int nt = (int)(t / (unsigned)KT), kt = (int)(t - (unsigned)nt * (unsigned)KT);
    const int n = nt * 32 + (l & 31), k0 = kt * 16 + 8 * (l >> 5);
    uint32_t h[8], m[8], lo[8];
    float x[8];
    if (kvec && n < N && k0 + 8 <= K) {
      // k-contiguous weight (dYc, dX: sk == 1, rows and base 16-byte aligned): the lane's 8 values are
      // two 16-byte loads instead of eight 4-byte loads a row stride apart
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + n * sn + k0);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + n * sn + k0 + 4);
      x[0] = a0[0]; x[1] = a0[1]; x[2] = a0[2]; x[3] = a0[3];
      x[4] = a1[0]; x[5] = a1[1]; x[6] = a1[2]; x[7] = a1[3];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (n < N && k0 + j < K) ? src[n * sn + (k0 + j) * sk] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(x[j], h[j], m[j], lo[j]);
    out[idx] = (u32x4){pack_hi(h[0], h[1]), pack_hi(h[2], h[3]), pack_hi(h[4], h[5]), pack_hi(h[6], h[7])};
    out[plane + idx] = (u32x4){pack_hi(m[0], m[1]), pack_hi(m[2], m[3]), pack_hi(m[4], m[5]), pack_hi(m[6], m[7])};
    out[2 * plane + idx] = (u32x4){pack_hi(lo[0], lo[1]), pack_hi(lo[2], lo[3]), pack_hi(lo[4], lo[5]), pack_hi(lo[6], lo[7])};
  }
}

static int x6_nt32(int N) { return cdiv(N, XBN) * (XBN / 32); }
size_t dn_ws_gemm_pack(int N, int K) { return (size_t)3 * x6_nt32(N) * cdiv(K, XBK) * 1024; }

extern "C" int danet_gemm_pack_weights(danet_stream_t stream, int n, const danet_gemm_pack_t* jobs) {
  DANET_CHECK_ARG(jobs && n >= 1, "gemm_pack_weights: empty table");
  for (int base = 0; base < n; base += X6_MAX_PACK) {
    const int cnt = min(X6_MAX_PACK, n - base);
    X6PackArgs a;
    int64_t most = 0;
    for (int i = 0; i < cnt; ++i) {
      const danet_gemm_pack_t& j = jobs[base + i];
      DANET_CHECK_ARG(j.N > 0 && j.K > 0 && j.src && j.out && (((uintptr_t)j.out) & 15) == 0 &&
                      j.out_bytes >= dn_ws_gemm_pack(j.N, j.K), "gemm_pack_weights: bad job %d", base + i);
      X6PackJob& q = a.job[i];
      q.src = j.src; q.out = (u32x4*)j.out; q.sn = j.stride_n; q.sk = j.stride_k;
      q.N = j.N; q.K = j.K; q.NT32 = x6_nt32(j.N); q.KT = cdiv(j.K, XBK);
      most = max(most, (int64_t)q.NT32 * q.KT * 64);
    }
    DANET_CHECK_ARG(most < (1ll << 31), "gemm_pack_weights: a weight of 2^31 / 8 values or more");
    for (int i = cnt; i < X6_MAX_PACK; ++i) a.job[i] = a.job[0];
    dim3 grid((unsigned)min((int64_t)1024, cdiv64(most, 256)), (unsigned)cnt);
    gemm_x6_pack_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    DANET_CHECK_LAUNCH();
  }
  return DANET_OK;
}

// How many K slices?  `tiles` workgroups of `nkt` k-steps each on 512 slots (2 per CU).  Model of the
// launch, in units of one whole tile's time: the workgroups run in rounds of 512; a last round that
// fills a fraction f of the slots still takes 0.55 + 0.45 f of a round (its workgroups have the CUs
// to themselves but hide less latency); s slices divide a workgroup's time by s and cost the slab
// traffic of s partial tiles per tile plus one more launch (calibrated on the TN groups of cfg 2 and
// cfg 4: 0.032 tile-times per slice for tiles / nkt = 1, i.e. a short-K product with a large output
// is never sliced).  Measured against it: 160 tiles x 150-256 steps -> 3 (2 / 4: +15 %), 100 -> 5,
// 105 -> 4, 570 x 256 -> 2-3 (1: +2 % of the cfg-4h600 step), 266 x 256 -> 3.
static int x6_slices(int tiles, int nkt, int min_steps, int max_slices, double slab_cost = 0.032) {
  double best = 1e30;
  int bs = 1;
  for (int s = 1; s <= max_slices; ++s) {
    if (s > 1 && nkt / s < min_steps) break;
    const double r = (double)tiles * s / 512.0;
    const double whole = (double)(long long)r, frac = r - whole;
    const double rounds = whole + (frac > 0.0 ? 0.55 + 0.45 * frac : 0.0);
    const double cost = rounds / s + (s > 1 ? 0.02 + slab_cost * tiles / max(nkt, 1) * (s - 1) : 0.0);
    if (cost < best - 1e-9) { best = cost; bs = s; }
  }
  return bs;
}
// The NT kernel's plan: rows per workgroup tile (64 MI) and K slices.  MI = 2 unless pinned: the
// taller tiles were built to take the kernel off what looked like an L1 / L2 bandwidth bound (half
// the B-fragment traffic per matrix instruction, one workgroup per CU) and measured NO faster
// anywhere (profiles/r05_i_gemm_x6_nt_plans.txt: 4096^3 678.6 us at MI = 4 vs 680.8 at MI = 2; the
// step's shapes 0-10 % slower, they have too few 256-row tiles).  What bounds the kernel is the
// vector ALU: on gfx950 a vector instruction and a matrix instruction of one SIMD do not overlap
// (tools/csrc/mfma_bf16_rate.hip, profiles/r05_i_mfma_bf16_rate.txt: 48 matrix instructions + 102
// vector instructions per wave run at exactly 1536 / (1536 + 4 * 102) of the matrix-only rate, one
// or two waves per SIMD alike), and the in-kernel split of A is 1.8 vector instructions per matrix
// instruction whatever the tile shape.  option gemm_x6_plan = MI + 16 * slices pins either.
// Hybrid (round 6): with more than one round of tiles and a ragged last one, the whole rounds run whole
// (nwhole = a multiple of 512 tiles) and only the remainder is sliced -- the same model decides: cost of
// the remainder as a product of its own + the whole rounds, against the unsliced launch.  It pays where K
// is long enough for the remainder's slab traffic to be small against its k-loop.  Pinned:
// option gemm_x6_plan += 65536 (hybrid with the given slice count) / 131072 (never hybrid).
struct X6Plan { int mi, s, nwhole; };
static double x6_rounds(double r) {
  const double whole = (double)(long long)r, frac = r - whole;
  return whole + (frac > 0.0 ? 0.55 + 0.45 * frac : 0.0);
}
static X6Plan x6_plan(int M, int N, int nkt) {
  const int pin = danet_opt(OPT_GEMM_X6_PLAN);
  const int pin_mi = pin & 15, pin_s = (pin >> 4) & 0xFFF, pin_h = pin >> 16;
  X6Plan p;
  p.mi = (pin_mi >= 2 && pin_mi <= 4) ? pin_mi : 2;
  p.nwhole = 0;
  const int tiles = cdiv(M, 64 * p.mi) * cdiv(N, XBN);
  // (one workgroup per CU at MI > 2: the model's 512 slots hold twice the tiles)
  const int slots = p.mi == 2 ? 512 : 256;
  p.s = pin_s >= 1 ? min(min(pin_s, 255), max(nkt, 1)) :   // (8-bit slice counter in a ticket)
         x6_slices(p.mi == 2 ? tiles : 2 * tiles, nkt, 12, 4);
  const int whole = tiles / slots * slots, rem = tiles - whole;
  if (pin_h != 2 && whole > 0 && rem > 0 && (pin_h == 1 || pin_s == 0)) {
    const double plain = x6_rounds((double)tiles * p.s / slots) / p.s +
                         (p.s > 1 ? 0.02 + 0.032 * tiles / max(nkt, 1) * (p.s - 1) : 0.0);
    // (a margin of 8 %: the model is good to about that, and a short-K product gains nothing measurable --
    // the projection, 672 tiles x 38 k-steps: 89.1 us plain, 93.0 / 88.2 / 105.0 with 2 / 3 / 4 slices on the
    // remainder -- while the hoisted input half at H = 600, 608 tiles x 75 k-steps, goes from 146.5 to 128 us:
    // profiles/r06_b_gemm_x6_hybrid.txt)
    double best = pin_h == 1 ? 1e30 : 0.92 * plain;
    int bs = 0;
    for (int s = 2; s <= 4; ++s) {
      if (pin_h == 1 && pin_s >= 2 && s != min(pin_s, 4)) continue;
      if (nkt / s < 8) break;
      const double cost = (double)whole / slots + x6_rounds((double)rem * s / slots) / s +
                          0.02 + 0.032 * rem / max(nkt, 1) * (s - 1);
      if (cost < best - 1e-9) { best = cost; bs = s; }
    }
    if (bs) { p.s = bs; p.nwhole = whole; }
  }
  return p;
}

// Workspace of a sliced product: a FIXED ticket region at its start (one word per tile, so every call on
// this workspace agrees where the tickets are and slab data never lands on them), then the slabs
// [s][M][N].  The caller zero-initialises the workspace once and gives it to nothing else; a ticket is
// back at zero when its launch has finished (the last arriver clears it), and carries the launch's
// tag while the launch runs, so even an aborted launch's leftovers read as "no slice yet".
#define X6_MAX_TICKETS 4096
#define X6_TICKET_BYTES (X6_MAX_TICKETS * sizeof(unsigned))
static size_t x6_slab_bytes(int s, int M, int N) { return align_up((size_t)s * M * N * sizeof(float), 256); }
size_t dn_ws_gemm_x6(int M, int N, int K1, int K2) {
  const X6Plan p = x6_plan(M, N, cdiv(K1, XBK) + cdiv(K2, XBK));
  return p.s > 1 ? X6_TICKET_BYTES + x6_slab_bytes(p.s, M, N) : 0;
}

hipEvent_t dn_take_stop_event();   // gemm_f32.hip: the event armed by danet_next_launch_events

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once per (kernel,
// device) -- `done` = the kernel's bit mask of devices that have it; a failed set is an error
static hipError_t x6_set_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
  return e;
}

extern "C" int danet_gemm_x6(danet_stream_t stream_, int M, int N,
                             int K1, const float* A1, int lda1, const void* B1pk,
                             int K2, const float* A2, int lda2, const void* B2pk,
                             float* C, int ldc, const float* bias, void* ws, size_t ws_bytes) {
  // (consumed before any check can return: see sk_launch)
  hipEvent_t stop = dn_take_stop_event();
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(M > 0 && N > 0 && K1 > 0 && K2 >= 0 && ldc >= N, "gemm_x6: bad shape");
  DANET_CHECK_ARG(A1 && B1pk && C && (K2 == 0 || (A2 && B2pk)), "gemm_x6: null operand");
  if (K1 % 4 || lda1 % 4 || lda1 < K1 || (K2 > 0 && (K2 % 4 || lda2 % 4 || lda2 < K2)) ||
      ((((uintptr_t)A1 | (uintptr_t)A2) & 15) != 0)) {
    danet_set_error("gemm_x6: K and lda must be multiples of 4 and A 16-byte aligned (K1=%d K2=%d)", K1, K2);
    return DANET_ERR_UNSUPPORTED;
  }
  DANET_CHECK_ARG(((((uintptr_t)B1pk | (uintptr_t)B2pk) & 15) == 0), "gemm_x6: packed weight alignment");
  DANET_CHECK_ARG((int64_t)M * lda1 < (1ll << 29) && (K2 == 0 || (int64_t)M * lda2 < (1ll << 29)),
                  "gemm_x6: an operand spans 2 GiB or more");
  // (the packed weights are read through buffer resources with 32-bit sizes)
  DANET_CHECK_ARG((int64_t)3 * x6_nt32(N) * cdiv(K1, XBK) * 1024 < (1ll << 31) &&
                  (int64_t)3 * x6_nt32(N) * cdiv(K2, XBK) * 1024 < (1ll << 31),
                  "gemm_x6: a packed weight spans 2 GiB or more");
  X6Args g;
  g.A[0] = A1; g.Bp[0] = (const u32x4*)B1pk; g.lda[0] = lda1; g.K[0] = K1; g.KT[0] = cdiv(K1, XBK);
  g.A[1] = A2; g.Bp[1] = (const u32x4*)B2pk; g.lda[1] = lda2; g.K[1] = K2; g.KT[1] = cdiv(K2, XBK);
  g.npair = K2 > 0 ? 2 : 1;
  g.M = M; g.N = N; g.NT32 = x6_nt32(N); g.bias = bias;
  const int nkt = cdiv(K1, XBK) + cdiv(K2, XBK);
  const X6Plan plan = x6_plan(M, N, nkt);
  int s = plan.s;
  const int nt = cdiv(M, 64 * plan.mi) * cdiv(N, XBN);
  if (s > 1 && (N % 4 != 0 || ldc % 4 != 0 || ((uintptr_t)C & 15) != 0 || nt > X6_MAX_TICKETS ||
                (size_t)s * M * N * sizeof(float) >= 0xFFFFFFF0ull)) s = 1;   // (the slice sum is vectorised; 32-bit slab offsets)
  int nwhole = s > 1 ? plan.nwhole : 0;
  g.C = C; g.ldc = ldc; g.slab = nullptr; g.tickets = nullptr; g.seq = 0;
  if (s > 1) {
    const size_t need = X6_TICKET_BYTES + x6_slab_bytes(s, M, N);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 15) != 0) {
      danet_set_error("gemm_x6: workspace %zu < %zu (or not 16-B aligned)", ws_bytes, need);
      return DANET_ERR_WORKSPACE;
    }
    // ticket tag of this launch: process-wide, so launches from any thread / stream differ
    static std::atomic<unsigned> launch_seq{0x00a5c3u};
    g.slab = (float*)((char*)ws + X6_TICKET_BYTES);
    g.tickets = (unsigned*)ws;
    g.seq = launch_seq.fetch_add(1, std::memory_order_relaxed) & 0xFFFFFFu;
  }
  g.splitk = s; g.nwhole = nwhole;
  dim3 grid((unsigned)(nwhole + (nt - nwhole) * s)), block(256);
  hipEvent_t kstop = stop;
#define X6_LAUNCH(MI)                                                                                          \
  { static std::atomic<unsigned long long> done{0};                                                            \
    DANET_CHECK_HIP(x6_set_lds((const void*)gemm_x6_nt_kernel<MI>, X6Geo<MI>::SMEM, done));                    \
    if (kstop) hipExtLaunchKernelGGL(gemm_x6_nt_kernel<MI>, grid, block, X6Geo<MI>::SMEM, stream, nullptr, kstop, 0, g); \
    else gemm_x6_nt_kernel<MI><<<grid, block, X6Geo<MI>::SMEM, stream>>>(g); }
  if (plan.mi == 4) X6_LAUNCH(4) else if (plan.mi == 3) X6_LAUNCH(3) else X6_LAUNCH(2)
#undef X6_LAUNCH
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// ================================================================== TN: both operands are activations
// C[M][N] (+)= A^T B with A [K][lda] (M contiguous) and B [K][ldb] (N contiguous): the weight
// gradients dW = x^T da (K = T*B).  Neither operand can be packed ahead of time and both are K-MAJOR,
// while the matrix instruction wants 8 consecutive k per lane.  The split pieces are therefore stored
// in LDS the way they arrive -- 4 consecutive m of one k per thread, as [k/4][m/16][4][16] blocks of
// bf16 -- and ds_read_b64_tr_b16 (each group of 16 lanes transposes a 4 x 16 block) delivers 4
// consecutive k of one m per lane; two such reads are one operand (tools/csrc/tr_read_layout.hip
// checks the recipe through the matrix instruction).  Both LDS directions are conflict-free: the
// k-rows of a block are rotated by the block's m index, so the 16 lanes of a write that share a k-row
// hit every bank once; a read instruction takes two adjacent 128-byte blocks per 32 lanes.
// Up to 6 products sharing K in one launch (a layer's dWx / dWh of both directions), tile per
// workgroup 128 x 128 x 16, K cut into `splitk` slices summed in slice order by a second kernel.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

#define TIMG 4096                        // one piece image: [16 k][128 m] bf16
#define TSTAGE (6 * TIMG)                // A hi/mid/lo, B hi/mid/lo: 24 KB
#define X6T_SMEM_BYTES (3 * TSTAGE)      // 72 KB: two workgroups per CU
#define X6T_MAX_PROBLEMS 6

struct X6TProblem {
  const float* A; const float* B; float* C;
  int lda, ldb, ldc, M, N;
  int Mmain;                             // rows covered by 128-row matrix-core tiles; the rest are TAIL rows
  float beta;
  int tile0, tiles_n;
  long long slab_off;                    // this problem's [M][N] inside a K slice's slab (floats)
};
struct X6TArgs {
  X6TProblem p[X6T_MAX_PROBLEMS];
  int nprob, K, splitk, tiles;
  int ntail;                             // leading workgroups of the launch that compute tail rows
  long long slab_stride;                 // floats per K slice
  float* slab;
};

struct X6TCtx {
  __amdgpu_buffer_rsrc_t rsa, rsb;
  int lda, ldb, K, kt1, mrem, nrem;
  int krow;            // this thread's k row inside a k-tile: 4 wave + lane / 16
  int mq4;             // first of its 4 columns in load 0: 4 (lane % 16); load 1: + 64
  int woff;            // byte offset of its 8-byte write inside a piece image (load 0; load 1: + 512)
  int roffa[2], roffb[2];   // byte offsets of its first transpose read of A tile 0 / 1, B tile 0 / 1
  // the k-loop's staging loads (RAGGED = false): byte offset of (k row krow, first column) of load li
  // inside its operand, X6_OOB for columns beyond the operand; the k-tile travels in the scalar offset
  unsigned lvo[4];
};

template <bool RAGGED>
__device__ __forceinline__ f32x4 x6t_load(__amdgpu_buffer_rsrc_t rs, int ld, int rem, int K, int kt, int kt1,
                                          int krow, int c) {
  const int k = kt * 16 + krow;
  const bool ok = kt < kt1 && k < K && c < rem;
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(k * ld + c) * 4u : X6_OOB, 0, 0));
  if (RAGGED) {
#pragma unroll
    for (int e = 1; e < 4; ++e) v[e] = (c + e < rem) ? v[e] : 0.f;
  }
  return v;
}
// the four loads of one k-tile: A columns [mq4, +4) and [mq4 + 64, +4), B likewise
template <bool RAGGED>
__device__ __forceinline__ void x6t_load_tile(const X6TCtx& c, int kt, f32x4 (&R)[4]) {
  R[0] = x6t_load<RAGGED>(c.rsa, c.lda, c.mrem, c.K, kt, c.kt1, c.krow, c.mq4);
  R[1] = x6t_load<RAGGED>(c.rsa, c.lda, c.mrem, c.K, kt, c.kt1, c.krow, c.mq4 + 64);
  R[2] = x6t_load<RAGGED>(c.rsb, c.ldb, c.nrem, c.K, kt, c.kt1, c.krow, c.mq4);
  R[3] = x6t_load<RAGGED>(c.rsb, c.ldb, c.nrem, c.K, kt, c.kt1, c.krow, c.mq4 + 64);
}
__device__ __forceinline__ void x6t_write(char* stage, const X6TCtx& c, int li, const uint32_t (&h)[2],
                                          const uint32_t (&m)[2], const uint32_t (&l)[2]) {
  char* p = stage + (li >> 1) * 3 * TIMG + c.woff + (li & 1) * 512;
  *reinterpret_cast<u32x2*>(p) = (u32x2){h[0], h[1]};
  *reinterpret_cast<u32x2*>(p + TIMG) = (u32x2){m[0], m[1]};
  *reinterpret_cast<u32x2*>(p + 2 * TIMG) = (u32x2){l[0], l[1]};
}
__device__ __forceinline__ void x6t_store_tile(char* stage, const X6TCtx& c, const f32x4 (&R)[4]) {
#pragma unroll
  for (int li = 0; li < 4; ++li) {
    uint32_t h[2], m[2], l[2];
    split_pair(R[li][0], R[li][1], h[0], m[0], l[0]);
    split_pair(R[li][2], R[li][3], h[1], m[1], l[1]);
    x6t_write(stage, c, li, h, m, l);
  }
}
// operand fragment: 8 consecutive k of this lane's row through two transpose reads
__device__ __forceinline__ bf16x8 x6t_frag(const char* img, int roff) {
  const u16x4 a = __builtin_bit_cast(u16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + roff)));
  const u16x4 b = __builtin_bit_cast(u16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + roff + 1024)));
  return __builtin_bit_cast(bf16x8, (u16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}
__device__ __forceinline__ void x6t_frags(const char* stage, const X6TCtx& c, bf16x8 (&fa)[3][2], bf16x8 (&fb)[3][2]) {
#pragma unroll
  for (int pc = 0; pc < 3; ++pc)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[pc][i] = x6t_frag(stage + pc * TIMG, c.roffa[i]);
      fb[pc][i] = x6t_frag(stage + (3 + pc) * TIMG, c.roffb[i]);
    }
}

template <int S>
__device__ __forceinline__ void x6t_mf(f32x16 (&acc)[2][2], const bf16x8 (&fa)[3][2], const bf16x8 (&fb)[3][2]) {
  constexpr int t = S >> 2, i = (S >> 1) & 1, j = S & 1;
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X6_PA[t]][i], fb[X6_PB[t]][j], acc[i][j], 0, 0, 0);
}

// One k-step in 24 fenced slots of one matrix instruction each (cf. x6_step):
//   slots  0-11  the 12 operand fragments of tile kt + 1 (two transpose reads each)
//   slots 12-19  split of the 8 value pairs of tile kt + 2 staged two steps ago
//   slots 20-23  LDS store of one load's pieces + the global load of tile kt + 4 into its registers
template <bool RAGGED, int SRD, int SWR>
__device__ __forceinline__ void x6t_step(const X6TCtx& c, char* xsm, int kt, int srd, int swr, f32x16 (&acc)[2][2],
                                         f32x4 (&R)[4], const bf16x8 (&fac)[3][2], const bf16x8 (&fbc)[3][2],
                                         bf16x8 (&fan)[3][2], bf16x8 (&fbn)[3][2]) {
  const char* srdp = xsm + (SRD >= 0 ? SRD : srd) * TSTAGE;      // SRD < 0: run-time stages (the loop's tail)
  char* swrp = xsm + (SRD >= 0 ? SWR : swr) * TSTAGE;
  uint32_t h[4][2], m[4][2], l[4][2];
#define X6T_FA(S, PC, I) x6t_mf<S>(acc, fac, fbc); fan[PC][I] = x6t_frag(srdp + (PC) * TIMG, c.roffa[I]); X6_FENCE
#define X6T_FB(S, PC, I) x6t_mf<S>(acc, fac, fbc); fbn[PC][I] = x6t_frag(srdp + (3 + (PC)) * TIMG, c.roffb[I]); X6_FENCE
  X6T_FA(0, 0, 0) X6T_FA(1, 0, 1) X6T_FA(2, 1, 0) X6T_FA(3, 1, 1) X6T_FA(4, 2, 0) X6T_FA(5, 2, 1)
  X6T_FB(6, 0, 0) X6T_FB(7, 0, 1) X6T_FB(8, 1, 0) X6T_FB(9, 1, 1) X6T_FB(10, 2, 0) X6T_FB(11, 2, 1)
#undef X6T_FA
#undef X6T_FB
#define X6T_S(S, LI, HF) x6t_mf<S>(acc, fac, fbc); split_pair_pk(__builtin_shufflevector(R[LI], R[LI], 2 * (HF), 2 * (HF) + 1), h[LI][HF], m[LI][HF], l[LI][HF]); X6_FENCE
  X6T_S(12, 0, 0) X6T_S(13, 0, 1) X6T_S(14, 1, 0) X6T_S(15, 1, 1) X6T_S(16, 2, 0) X6T_S(17, 2, 1) X6T_S(18, 3, 0) X6T_S(19, 3, 1)
#undef X6T_S
  // (RAGGED = false: no vector instruction per load but the select of an out-of-range offset for a
  // k row beyond K / beyond the slice -- one compare per step)
  const int kta = kt + 4;
  const bool kok = c.krow < (kta < c.kt1 ? c.K - kta * 16 : 0);
  const unsigned soa = kta < c.kt1 ? (unsigned)(kta * 16 * c.lda) * 4u : 0u, sob = kta < c.kt1 ? (unsigned)(kta * 16 * c.ldb) * 4u : 0u;
#define X6T_W(S, LI) x6t_mf<S>(acc, fac, fbc); x6t_write(swrp, c, LI, h[LI], m[LI], l[LI]);          \
  if (RAGGED) R[LI] = x6t_load<true>((LI) < 2 ? c.rsa : c.rsb, (LI) < 2 ? c.lda : c.ldb, (LI) < 2 ? c.mrem : c.nrem, c.K, kt + 4, \
                                     c.kt1, c.krow, c.mq4 + ((LI) & 1) * 64);                              \
  else R[LI] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((LI) < 2 ? c.rsa : c.rsb,   \
                                     kok ? c.lvo[LI] : X6_OOB, (LI) < 2 ? soa : sob, 0)); X6_FENCE
  X6T_W(20, 0) X6T_W(21, 1) X6T_W(22, 2) X6T_W(23, 3)
#undef X6T_W
}

// TAIL rows.  M = 129 (the bottom layer's dWx: 129 spectrogram bins) would cost a second 128-row tile
// row with ONE valid row -- 20 of the group's 100 tiles at cfg 2.  A problem with 1..X6T_TAIL_MAX rows
// beyond a multiple of 128 gets those rows from plain fp32 FMA chains instead: the first `ntail`
// workgroups of the launch (dispatched first, they run in slots the tiles leave idle)
// take one K slice x 64 columns each -- 16 k-lanes x 16 column groups of 4, the lanes summed in lane
// order -- and write to the rows of the slice's slab (or of C) that the tiles of the problem do not
// cover, so the slice sum treats them like any other row.  Exact fp32 products, deterministic.
// Measured (bottom layer's group, cfg 2, alone): 89.5 -> 85.7 us; workgroups of 256 columns x 4
// k-lanes (a wave per lane, 1 KB row segments) take longer than the tiles: 114 us.
#define X6T_TAIL_MAX 4
__device__ __forceinline__ void x6t_tail(const X6TArgs& g, int t, char* xsm) {
  int pi = -1, per_p = 0;
#pragma unroll
  for (int i = 0; i < X6T_MAX_PROBLEMS; ++i) {
    if (i < g.nprob && pi < 0 && g.p[i].Mmain < g.p[i].M) {
      per_p = ((g.p[i].N + 63) / 64) * g.splitk;
      if (t < per_p) pi = i; else t -= per_p;
    }
  }
  if (pi < 0) return;
  const X6TProblem& q = g.p[pi];
  const int chunks = (q.N + 63) / 64;
  const int z = t / chunks, n0 = (t % chunks) * 64;
  const int nkt = (g.K + 15) / 16;
  const int per = (nkt + g.splitk - 1) / g.splitk;
  const int k0 = z * per * 16, k1 = min(g.K, (z + 1) * per * 16);
  const int tid = threadIdx.x, kl = tid >> 4, cg = tid & 15;
  const int n = n0 + cg * 4;
  const int R = q.M - q.Mmain;                         // 1..X6T_TAIL_MAX
  f32x4 acc[X6T_TAIL_MAX];
#pragma unroll
  for (int r = 0; r < X6T_TAIL_MAX; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (n < q.N) {                                       // (N % 4 == 0: a column group is whole or absent)
    const float* __restrict__ bp = q.B + n;
    const float* __restrict__ ap = q.A + q.Mmain;
#pragma unroll 4
    for (int k = k0 + kl; k < k1; k += 16) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + (size_t)k * q.ldb);
#pragma unroll
      for (int r = 0; r < X6T_TAIL_MAX; ++r)
        if (r < R) {
          const float a = ap[(size_t)k * q.lda + r];
          acc[r][0] = fmaf(a, bv[0], acc[r][0]); acc[r][1] = fmaf(a, bv[1], acc[r][1]);
          acc[r][2] = fmaf(a, bv[2], acc[r][2]); acc[r][3] = fmaf(a, bv[3], acc[r][3]);
        }
    }
  }
  f32x4* red = reinterpret_cast<f32x4*>(xsm);          // [16 k-lanes][16 column groups]
  const bool sliced = g.splitk > 1;
#pragma unroll
  for (int r = 0; r < X6T_TAIL_MAX; ++r) {
    if (r >= R) break;                                 // (uniform)
    __syncthreads();
    red[kl * 16 + cg] = acc[r];
    __syncthreads();
    if (kl == 0 && n < q.N) {
      f32x4 v = red[cg];
      for (int j = 1; j < 16; ++j) v += red[j * 16 + cg];
      const int row = q.Mmain + r;
      float* o = sliced ? g.slab + (size_t)z * g.slab_stride + q.slab_off + (size_t)row * q.N + n
                        : q.C + (size_t)row * q.ldc + n;
      if (!sliced && q.beta != 0.f) v += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = v;
    }
  }
}

template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void gemm_x6_tn_kernel(X6TArgs g) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  if ((int)blockIdx.x < g.ntail) {                     // (uniform)
    x6t_tail(g, (int)blockIdx.x, xsm);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int item = x6_xcd_item((int)blockIdx.x - g.ntail, (int)gridDim.x - g.ntail);
  const int z = item / g.tiles;                        // K slice
  const int t = item % g.tiles;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < X6T_MAX_PROBLEMS; ++i)
    if (i < g.nprob && t >= g.p[i].tile0) pi = i;      // uniform
  const X6TProblem& q = g.p[pi];
  const int tl = t - q.tile0;
  const int m0 = (tl / q.tiles_n) * 128, n0 = (tl % q.tiles_n) * 128;
  const int M = q.Mmain, N = q.N, K = g.K;             // (tail rows: x6t_tail)

  const int nkt = (K + 15) / 16;
  const int per = (nkt + g.splitk - 1) / g.splitk;
  const int kt0 = z * per, kt1 = min(nkt, kt0 + per);

  X6TCtx c;
  {
    const float* ab = q.A + m0;
    const float* bb = q.B + n0;
    // (whole 16-byte groups of the last k row are in range)
    c.rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ab), 0, (int)(((size_t)(K - 1) * q.lda + ((M - m0 + 3) & ~3)) * 4), 0x00020000);
    c.rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bb), 0, (int)(((size_t)(K - 1) * q.ldb + ((N - n0 + 3) & ~3)) * 4), 0x00020000);
  }
  c.lda = q.lda; c.ldb = q.ldb; c.K = K; c.kt1 = kt1; c.mrem = M - m0; c.nrem = N - n0;
  const int kr = lane >> 4, mq = lane & 15;
  c.krow = 4 * wave + kr;
  c.mq4 = 4 * mq;
#pragma unroll
  for (int li = 0; li < 4; ++li) {
    const int col = c.mq4 + (li & 1) * 64;
    c.lvo[li] = col < (li < 2 ? c.mrem : c.nrem) ? (unsigned)(c.krow * (li < 2 ? c.lda : c.ldb) + col) * 4u : X6_OOB;
  }
  // a block's four k-rows sit in slot (k % 4 + m-block) % 4 of its 128 bytes (the transpose read takes
  // one address per lane, so the rows of a block may be permuted): the 16 lanes of a write that share
  // a k-row then cover four m-blocks in four different 32-byte slots -- every bank once
  c.woff = (wave * 8 + (mq >> 2)) * 128 + ((kr + (mq >> 2)) & 3) * 32 + (mq & 3) * 8;
  {
    const int i16 = lane & 15, mbit = (lane >> 4) & 1, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ba = (wm * 64 + i * 32 + 16 * mbit) >> 4, bb = (wn * 64 + i * 32 + 16 * mbit) >> 4;
      c.roffa[i] = (2 * kh * 8 + ba) * 128 + (((i16 >> 2) + ba) & 3) * 32 + (i16 & 3) * 8;
      c.roffb[i] = (2 * kh * 8 + bb) * 128 + (((i16 >> 2) + bb) & 3) * 32 + (i16 & 3) * 8;
    }
  }
  const int fi = lane & 31, kb = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt0 < kt1) {
    f32x4 R0[4], R1[4];
    bf16x8 faA[3][2], fbA[3][2], faB[3][2], fbB[3][2];
    x6t_load_tile<RAGGED>(c, kt0, R0); x6t_load_tile<RAGGED>(c, kt0 + 1, R1);
    x6t_store_tile(xsm, c, R0); x6t_store_tile(xsm + TSTAGE, c, R1);
    x6t_load_tile<RAGGED>(c, kt0 + 2, R0); x6t_load_tile<RAGGED>(c, kt0 + 3, R1);
    __syncthreads();
    x6t_frags(xsm, c, faA, fbA);
    int kt = kt0, srd = 1, swr = 2;
    // (unrolled over the 3 LDS stages x 2 register sets: every LDS address is a loop-invariant
    // register + an immediate; the tail takes the stages at run time)
#define X6T_STEP(SRD, SWR, R, FAC, FBC, FAN, FBN)                                                    \
    { x6t_step<RAGGED, SRD, SWR>(c, xsm, kt, srd, swr, acc, R, FAC, FBC, FAN, FBN);                  \
      __syncthreads();                                                                               \
      ++kt; }
#define X6T_STEP_RT(R, FAC, FBC, FAN, FBN)                                                           \
    { x6t_step<RAGGED, -1, -1>(c, xsm, kt, srd, swr, acc, R, FAC, FBC, FAN, FBN);                    \
      __syncthreads();                                                                               \
      ++kt; srd = swr; swr = swr == 2 ? 0 : swr + 1; }
    while (kt + 5 < kt1) {
      X6T_STEP(1, 2, R0, faA, fbA, faB, fbB)
      X6T_STEP(2, 0, R1, faB, fbB, faA, fbA)
      X6T_STEP(0, 1, R0, faA, fbA, faB, fbB)
      X6T_STEP(1, 2, R1, faB, fbB, faA, fbA)
      X6T_STEP(2, 0, R0, faA, fbA, faB, fbB)
      X6T_STEP(0, 1, R1, faB, fbB, faA, fbA)
    }
    while (kt + 1 < kt1) {
      X6T_STEP_RT(R0, faA, fbA, faB, fbB)
      X6T_STEP_RT(R1, faB, fbB, faA, fbA)
    }
    if (kt < kt1) X6T_STEP_RT(R0, faA, fbA, faB, fbB)
#undef X6T_STEP
#undef X6T_STEP_RT
  }

  // epilogue: the tile leaves through LDS in two halves of 64 rows (16-byte stores of 512-byte row
  // segments) when the destination allows it; beta = 1 adds what is there (one K slice only:
  // with slices the reduce kernel applies beta)
  const bool sliced = g.splitk > 1;
  float* __restrict__ dst = sliced ? g.slab + (size_t)z * g.slab_stride + q.slab_off : q.C;
  const int ldc = sliced ? N : q.ldc;
  const float beta = sliced ? 0.f : q.beta;
  const bool vec = (ldc % 4 == 0) && (((uintptr_t)dst & 15) == 0) && (n0 + 128 <= N);   // uniform
  if (vec) {
    float* ct = reinterpret_cast<float*>(xsm);          // [64][132]
    constexpr int LDC_T = 132;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb) * LDC_T + wn * 64 + j * 32 + fi] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rr = (tid >> 5) + 8 * it, c4 = tid & 31;
        const int row = m0 + half * 64 + rr;
        if (row < M) {
          f32x4* o = reinterpret_cast<f32x4*>(dst + (size_t)row * ldc + n0 + c4 * 4);
          f32x4 v = *reinterpret_cast<const f32x4*>(&ct[rr * LDC_T + c4 * 4]);
          if (beta != 0.f) v += *o;
          *o = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + fi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        if (row < M && col < N) {
          float* o = dst + (size_t)row * ldc + col;
          *o = acc[i][j][r] + (beta != 0.f ? *o : 0.f);
        }
      }
    }
}

// C = beta C + the K slices in slice order; blockIdx.y = problem (N % 4 == 0, ldc % 4 == 0).
// (Round 6 built the NT kernel's in-launch sum for this kernel as well -- a ticket per tile and per tail
// chunk, the last arriver adds the slabs -- and measured it SLOWER than this second launch: layer group
// 136.2 vs 130.6 us, bottom layer's group (6 slices) 93.0 vs 79.3, dWout 95.0 vs 91.0, the cfg-2 step
// 2.462-2.471 vs 2.434-2.441 ms on one box: 80-160 last arrivers each pull 0.4-0.5 MB through
// latency-bound write-through loads where 1024 workgroups of this launch stream it.
// profiles/r06_a_tn_last_arriver.txt, profiles/r06_a_tn_last_arriver_experiment.patch.)
__global__ __launch_bounds__(256) void gemm_x6_tn_reduce_kernel(X6TArgs g) {
  const X6TProblem& q = g.p[blockIdx.y];
  const int M = q.M, N = q.N, ldc = q.ldc;
  const float beta = q.beta;
  const float* __restrict__ slab = g.slab + q.slab_off;
  const int64_t n4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(slab)[i];
    for (int s = 1; s < g.splitk; ++s) v += reinterpret_cast<const f32x4*>(slab + (size_t)s * g.slab_stride)[i];
    const int64_t e = i * 4;
    f32x4* o = reinterpret_cast<f32x4*>(q.C + (size_t)(e / N) * ldc + (e % N));
    if (beta != 0.f) v += *o;
    *o = v;
  }
}

#ifdef X6T_FORCE_S   // experiment builds: a pinned slice count for the layer groups (tools, EXPERIMENTS)
static int x6t_splitk(int tiles, int K) { return tiles >= 100 ? X6T_FORCE_S : x6_slices(tiles, cdiv(K, 16), 16, 8); }
#else
// (slab cost 0.045 per slice instead of the NT products' 0.032: the groups run BESIDE a BPTT kernel, where the slab
// round trip shares the memory system with the kernel's exchange.  It changes one decision among the step's shapes
// -- the 570-tile layer groups of H = 600: 3 -> 2 slices, cfg 4 as written 7.85-7.87 -> 7.75-7.76 ms, two alternating
// pairs of pinned builds; cfg 2's 160-tile groups stay at 3 (2: the same within noise; 4 / 5: +5-10 us))
static int x6t_splitk(int tiles, int K) { return x6_slices(tiles, cdiv(K, 16), 16, 8, 0.045); }
#endif
// rows of a problem covered by 128-row tiles: all of them, unless 1..X6T_TAIL_MAX rows hang over a
// multiple of 128 (then those are tail rows: x6t_tail).  include/danet_hip.h states the same rule
// for the tile count a caller passes to danet_workspace_bytes(DANET_WS_GEMM_X6_TN, ...).
static int x6t_main_rows(int M, int N) {
  const int tail = M % 128;
  return (M > 128 && tail >= 1 && tail <= X6T_TAIL_MAX && N % 4 == 0) ? M - tail : M;
}
size_t dn_ws_gemm_x6_tn(long long sum_mn, int tiles, int K) {
  const int s = x6t_splitk(tiles, K);
  return s > 1 ? (size_t)s * (size_t)sum_mn * sizeof(float) : 0;
}

extern "C" int danet_gemm_x6_tn_grouped(danet_stream_t stream_, int K, int nprob, const danet_gemm_problem_t* probs,
                                        void* ws, size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(probs && nprob >= 1 && nprob <= X6T_MAX_PROBLEMS, "gemm_x6_tn: 1..%d problems", X6T_MAX_PROBLEMS);
  DANET_CHECK_ARG(K > 0, "gemm_x6_tn: non-positive K %d", K);
  X6TArgs g;
  int tiles = 0;
  long long sum_mn = 0;
  bool slicable = true, ragged = false;
  for (int i = 0; i < nprob; ++i) {
    const danet_gemm_problem_t& q = probs[i];
    DANET_CHECK_ARG(q.M > 0 && q.N > 0 && q.A && q.B && q.C, "gemm_x6_tn: bad problem %d", i);
    DANET_CHECK_ARG(q.bias == nullptr && (q.beta == 0.f || q.beta == 1.f), "gemm_x6_tn: no bias; beta 0 or 1");
    DANET_CHECK_ARG(q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N, "gemm_x6_tn: leading dimension too small");
    if (q.lda % 4 || q.ldb % 4 || ((((uintptr_t)q.A | (uintptr_t)q.B)) & 15) != 0) {
      danet_set_error("gemm_x6_tn: lda / ldb must be multiples of 4 and A / B 16-byte aligned (problem %d)", i);
      return DANET_ERR_UNSUPPORTED;
    }
    DANET_CHECK_ARG((long long)K * q.lda < (1ll << 29) && (long long)K * q.ldb < (1ll << 29),
                    "gemm_x6_tn: an operand spans 2 GiB or more");
    X6TProblem& p = g.p[i];
    p.A = q.A; p.B = q.B; p.C = q.C; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.M = q.M; p.N = q.N;
    p.beta = q.beta;
    p.Mmain = ((((uintptr_t)q.C) & 15) == 0 && q.ldc % 4 == 0) ? x6t_main_rows(q.M, q.N) : q.M;
    p.tiles_n = cdiv(q.N, 128);
    p.tile0 = tiles;
    p.slab_off = sum_mn;
    tiles += cdiv(p.Mmain, 128) * p.tiles_n;
    sum_mn += (long long)q.M * q.N;
    // Rows are read in whole 16-byte groups inside the row pitch (lda / ldb are multiples of 4).  What sits
    // in the pad of a row whose valid length is a multiple of 2 only reaches accumulator rows / columns
    // beyond M / N, which are never stored.  With an ODD valid length the last valid value shares its
    // split pair (m, m ^ 1) with the first pad value, and a non-finite pad would turn the valid one's
    // remainder into NaN (Inf * 0 in split_pair's v_dot2c): such groups take the kernel variant that
    // masks the pad per element (round 6, advisor finding; the step's own shapes are all even).
    ragged = ragged || (p.Mmain % 2 != 0) || (q.N % 2 != 0);
    slicable = slicable && q.N % 4 == 0 && q.ldc % 4 == 0 && (((uintptr_t)q.C) & 15) == 0 && ((long long)q.M * q.N) % 4 == 0;
  }
  for (int i = nprob; i < X6T_MAX_PROBLEMS; ++i) g.p[i] = g.p[0];
  g.nprob = nprob; g.K = K; g.tiles = tiles;
  int s = slicable ? x6t_splitk(tiles, K) : 1;
  g.splitk = s;
  g.slab_stride = sum_mn;
  g.slab = (float*)ws;
  g.ntail = 0;
  for (int i = 0; i < nprob; ++i)
    if (g.p[i].Mmain < g.p[i].M) g.ntail += cdiv(g.p[i].N, 64) * s;
  if (s > 1) {
    const size_t need = (size_t)s * (size_t)sum_mn * sizeof(float);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 15) != 0) {
      danet_set_error("gemm_x6_tn: workspace %zu < %zu (or not 16-B aligned)", ws_bytes, need);
      return DANET_ERR_WORKSPACE;
    }
  }
  static std::atomic<unsigned long long> done[2];
  dim3 grid((unsigned)(tiles * s + g.ntail)), block(256);
  if (ragged) {
    DANET_CHECK_HIP(x6_set_lds((const void*)gemm_x6_tn_kernel<true>, X6T_SMEM_BYTES, done[1]));
    gemm_x6_tn_kernel<true><<<grid, block, X6T_SMEM_BYTES, stream>>>(g);
  } else {
    DANET_CHECK_HIP(x6_set_lds((const void*)gemm_x6_tn_kernel<false>, X6T_SMEM_BYTES, done[0]));
    gemm_x6_tn_kernel<false><<<grid, block, X6T_SMEM_BYTES, stream>>>(g);
  }
  DANET_CHECK_LAUNCH();
  if (s > 1) {
    long long most = 0;
    for (int i = 0; i < nprob; ++i) most = max(most, (long long)probs[i].M * probs[i].N / 4);
    dim3 rgrid((unsigned)min((long long)1024, (most + 255) / 256), (unsigned)nprob);
    gemm_x6_tn_reduce_kernel<<<rgrid, block, 0, stream>>>(g);
    DANET_CHECK_LAUNCH();
  }
  return DANET_OK;
}
