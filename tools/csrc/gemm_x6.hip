// STATUS: PROTOTYPE, not part of libdanet_hip.so (round 4).  tools/bench_gemm_x6.py builds it on
// demand and prints error and time next to the product's exact-fp32 kernels; results and what a
// product version needs: DESIGN.md 8.0, profiles/r04_gemm_x6_prototype*.txt.
//
// fp32 products on the BF16 matrix cores (gfx950).  Every fp32 value is EXACTLY hi + mid + lo with
// three bf16 pieces of 8 significant bits (hi = x rounded to 8 bits, mid = the remainder rounded
// to 8 bits, lo = what is left), and the six largest of the nine piece products are accumulated in
// fp32:
//     A B^T ~ Ahi Bhi + (Ahi Bmid + Amid Bhi) + (Ahi Blo + Alo Bhi + Amid Bmid)
// The dropped products are < 2^-25 |a||b| each; the result is as close to the float64 product as
// an fp32 FMA chain's (tools/bf16x_split_accuracy.py, tests).  v_mfma_f32_32x32x16_bf16 runs at 16x
// the rate of v_mfma_f32_32x32x2_f32; six of them per 16 k replace eight fp32 instructions -> 2.7x
// the matrix-core throughput of the exact-fp32 kernels in gemm_f32.hip.
//
// Operands:  A [M][lda] fp32, K-contiguous: an activation, split on its way from global memory to
//            LDS (vector-ALU work in the matrix instructions' shadow);
//            B [N][ldb] as three bf16 piece arrays, K-contiguous: a WEIGHT, split once per step by
//            danet_split3_bf16 (every row panel of A re-uses it);
// optional second operand pair (K-concatenation: dX = da_f Wx_f^T + da_b Wx_b^T), beta = 0, no bias.
// Schedule: tile per workgroup, 128 x 128 x 16, 4 waves as 2 x 2 with 2 x 2 MFMA tiles of 32 x 32
// each; products with too few tiles for the GPU are cut along K into `splitk` slices whose partial
// tiles a second kernel sums in slice order (deterministic).  Register staging two k-tiles ahead,
// two LDS stages of six piece images [128 rows][16 k] bf16, the two 16-byte chunks of a row
// swapped on odd groups of four rows (conflict-free 16-byte fragment reads).
#include "common.h"
extern "C" void danet_set_error(const char* fmt, ...) { (void)fmt; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define XBM 128
#define XBN 128
#define XBK 16
#define XIMG (XBM * XBK * 2)            // bytes of one piece image: 128 rows x 16 k x bf16 = 4 KB
#define XSTAGE (6 * XIMG)               // A hi/mid/lo, B hi/mid/lo
#define X6_SMEM_BYTES (2 * XSTAGE)      // 48 KB

struct X6Args {
  const float* A[2];
  const uint16_t* Bp[2][3];             // [pair][piece]
  int lda[2], ldb[2], K[2];
  float* C;                             // splitk == 1: the result; else slab [splitk][M][N]
  int M, N, ldc, npair, splitk;
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

// registers of one k-tile in flight: A as fp32 (two rows x 4 k per thread), B as pieces
struct X6Regs { f32x4 a[2]; u32x2 b[3][2]; };

__global__ __launch_bounds__(256, 2) void gemm_x6_nt_kernel(X6Args g) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + XBN - 1) / XBN;
  const int nt = ((g.M + XBM - 1) / XBM) * tiles_n;
  const int z = blockIdx.x / nt;                       // K slice
  int bid = blockIdx.x % nt;
  if ((nt & 7) == 0) bid = (bid & 7) * (nt >> 3) + (bid >> 3);     // an XCD walks a band of tiles
  const int m0 = (bid / tiles_n) * XBM, n0 = (bid % tiles_n) * XBN;

  // k-tiles of the concatenated contraction: [0, nk0) pair 0, [nk0, nkt) pair 1; this slice's range
  const int nk0 = (g.K[0] + XBK - 1) / XBK;
  const int nkt = nk0 + (g.npair > 1 ? (g.K[1] + XBK - 1) / XBK : 0);
  const int per = (nkt + g.splitk - 1) / g.splitk;
  const int kt0 = z * per, kt1 = min(nkt, kt0 + per);

  // staging map: thread -> (row = tid / 4 + 64 i, q = tid % 4): 4 threads cover a row's 16 k
  const int srow = tid >> 2, sq = tid & 3;
  auto load = [&](int kt, X6Regs& R) {
    const int p = kt >= nk0 ? 1 : 0;
    const int k = (kt - (p ? nk0 : 0)) * XBK + sq * 4;
    const int K = g.K[p];
    const float* __restrict__ Ap = g.A[p];
    const int lda = g.lda[p], ldb = g.ldb[p];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ar = m0 + srow + 64 * i, br = n0 + srow + 64 * i;
      R.a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ar < g.M && k < K) R.a[i] = *reinterpret_cast<const f32x4*>(Ap + (size_t)ar * lda + k);
      const bool ok = br < g.N && k < K;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        R.b[pc][i] = (u32x2){0u, 0u};
        if (ok) R.b[pc][i] = *reinterpret_cast<const u32x2*>(g.Bp[p][pc] + (size_t)br * ldb + k);
      }
    }
  };
  auto store = [&](int st, const X6Regs& R) {
    char* base = xsm + st * XSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = img_off(srow + 64 * i, sq);
      uint32_t h[4], m[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(R.a[i][j], h[j], m[j], l[j]);
      *reinterpret_cast<u32x2*>(base + off) = (u32x2){pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
      *reinterpret_cast<u32x2*>(base + XIMG + off) = (u32x2){pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
      *reinterpret_cast<u32x2*>(base + 2 * XIMG + off) = (u32x2){pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        *reinterpret_cast<u32x2*>(base + (3 + pc) * XIMG + off) = R.b[pc][i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fi = lane & 31, kb = lane >> 5;

  if (kt0 < kt1) {
    // prologue: tile kt0 -> stage 0; tiles kt0 + 1 and kt0 + 2 in registers.  R0 / R1 alternate:
    // at iteration kt the set holding tile kt + 1 is stored and refilled with tile kt + 3.
    X6Regs R0, R1;
    load(kt0, R0);
    store(0, R0);
    if (kt0 + 1 < kt1) load(kt0 + 1, R1);
    if (kt0 + 2 < kt1) load(kt0 + 2, R0);
    __syncthreads();
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const char* sb = xsm + stage * XSTAGE;
      bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[pc][i] = frag(sb + pc * XIMG, wm * 64 + i * 32 + fi, kb);
          fb[pc][i] = frag(sb + (3 + pc) * XIMG, wn * 64 + i * 32 + fi, kb);
        }
      // the next tile goes to the other stage while this one is multiplied (that stage was last
      // read before the barrier that ended the previous iteration); its register set is refilled
      // with the tile three ahead
      const bool odd = ((kt - kt0) & 1) != 0;          // uniform: which set holds tile kt + 1
      if (kt + 1 < kt1) { if (odd) store(stage ^ 1, R0); else store(stage ^ 1, R1); }
      if (kt + 3 < kt1) { if (odd) load(kt + 3, R0); else load(kt + 3, R1); }
      // small terms first; the four tiles interleaved: an accumulator is reused every 4th MFMA
#define X6_TERM(PA, PB)                                                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i], fb[PB][j], acc[i][j], 0, 0, 0);
      X6_TERM(2, 0) X6_TERM(0, 2) X6_TERM(1, 1) X6_TERM(1, 0) X6_TERM(0, 1) X6_TERM(0, 0)
#undef X6_TERM
      __syncthreads();
      stage ^= 1;
    }
  }

  // C/D layout 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The tile leaves
  // through the (now free) LDS in two halves of 64 rows as 16-byte stores of full 512-byte row
  // segments when the destination allows it; 4-byte stores otherwise.
  float* __restrict__ dst = g.C + (g.splitk > 1 ? (size_t)z * g.M * g.ldc : 0);
  const bool vec = (g.ldc % 4 == 0) && (((uintptr_t)g.C & 15) == 0) && (n0 + XBN <= g.N);   // uniform
  if (vec) {
    float* ct = reinterpret_cast<float*>(xsm);          // [64][XBN + 4]
    constexpr int LDC_T = XBN + 4;
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
        if (row < g.M)
          *reinterpret_cast<f32x4*>(dst + (size_t)row * g.ldc + n0 + c4 * 4) =
              *reinterpret_cast<const f32x4*>(&ct[rr * LDC_T + c4 * 4]);
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
        if (row < g.M && col < g.N) dst[(size_t)row * g.ldc + col] = acc[i][j][r];
      }
    }
}

// C[m][n] = sum over the K slices, in slice order (slab rows are dense: ld = N)
__global__ __launch_bounds__(256) void gemm_x6_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C,
                                                             int M, int N, int ldc, int splitk) {
  const int64_t n4 = (int64_t)M * N / 4;
  const int64_t plane = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(slab)[i];
    for (int s = 1; s < splitk; ++s) v += reinterpret_cast<const f32x4*>(slab + s * plane)[i];
    const int64_t e = i * 4;
    const int row = (int)(e / N), col = (int)(e % N);
    *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = v;
  }
}

// ------------------------------------------------------------------ operand pieces of a weight
// hi / mid / lo [n] bf16 (as uint16) of x [n]; n % 4 == 0, 16-byte aligned x, 8-byte aligned pieces
__global__ __launch_bounds__(256) void split3_kernel(int64_t n4, const f32x4* __restrict__ x,
                                                     u32x2* __restrict__ hi, u32x2* __restrict__ mid,
                                                     u32x2* __restrict__ lo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = x[i];
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[j], h[j], m[j], l[j]);
    hi[i] = (u32x2){pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
    mid[i] = (u32x2){pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
    lo[i] = (u32x2){pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
  }
}

// the pieces of x [M][N]^T: out [N][ldo] (ldo >= M), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void split3_transpose_kernel(int M, int N, const float* __restrict__ x, int ldx,
                                                               uint16_t* __restrict__ hi, uint16_t* __restrict__ mid,
                                                               uint16_t* __restrict__ lo, int ldo) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    t[r][tx] = (m0 + r < M && n0 + tx < N) ? x[(size_t)(m0 + r) * ldx + n0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (n0 + r < N && m0 + tx < ldo) {
      uint32_t h, m, l;
      split3(m0 + tx < M ? t[tx][r] : 0.f, h, m, l);
      const size_t o = (size_t)(n0 + r) * ldo + m0 + tx;
      hi[o] = (uint16_t)(h >> 16); mid[o] = (uint16_t)(m >> 16); lo[o] = (uint16_t)(l >> 16);
    }
}

extern "C" int danet_split3_bf16(danet_stream_t stream, int64_t n, const float* x, uint16_t* hi,
                                 uint16_t* mid, uint16_t* lo) {
  DANET_CHECK_ARG(n > 0 && n % 4 == 0 && x && hi && mid && lo, "split3: n must be a positive multiple of 4");
  DANET_CHECK_ARG((((uintptr_t)x) & 15) == 0 && ((((uintptr_t)hi | (uintptr_t)mid | (uintptr_t)lo)) & 7) == 0,
                  "split3: alignment");
  const int grid = (int)min((int64_t)2048, cdiv64(n / 4, 256));
  split3_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n / 4, (const f32x4*)x, (u32x2*)hi, (u32x2*)mid, (u32x2*)lo);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_split3_bf16_transpose(danet_stream_t stream, int M, int N, const float* x, int ldx,
                                           uint16_t* hi, uint16_t* mid, uint16_t* lo, int ldo) {
  DANET_CHECK_ARG(M > 0 && N > 0 && x && hi && mid && lo && ldx >= N && ldo >= M, "split3_transpose: bad args");
  dim3 grid((unsigned)cdiv(N, 32), (unsigned)cdiv(ldo, 32));
  split3_transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(M, N, x, ldx, hi, mid, lo, ldo);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

static int x6_splitk(int M, int N, int nkt) {
  // cut along K while the tiles alone leave more than a third of the workgroup slots (2 per CU)
  // empty and every slice keeps >= 24 k-tiles; <= 4 slices
  const int nt = cdiv(M, XBM) * cdiv(N, XBN);
  int s = 1;
  while (s < 4 && nt * s * 3 < 512 * 2 && nkt / (s + 1) >= 24) ++s;
  return s;
}

size_t dn_ws_gemm_x6(int M, int N, int K1, int K2) {
  const int s = x6_splitk(M, N, cdiv(K1, XBK) + cdiv(K2, XBK));
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int danet_gemm_x6_nt(danet_stream_t stream, int M, int N,
                                int K1, const float* A1, int lda1, const uint16_t* B1h, const uint16_t* B1m,
                                const uint16_t* B1l, int ldb1,
                                int K2, const float* A2, int lda2, const uint16_t* B2h, const uint16_t* B2m,
                                const uint16_t* B2l, int ldb2,
                                float* C, int ldc, void* ws, size_t ws_bytes) {
  DANET_CHECK_ARG(M > 0 && N > 0 && K1 > 0 && K2 >= 0 && ldc >= N, "gemm_x6: bad shape");
  DANET_CHECK_ARG(A1 && B1h && B1m && B1l && C && (K2 == 0 || (A2 && B2h && B2m && B2l)), "gemm_x6: null operand");
  if (K1 % 4 || lda1 % 4 || ldb1 % 4 || lda1 < K1 || ldb1 < K1 ||
      (K2 > 0 && (K2 % 4 || lda2 % 4 || ldb2 % 4 || lda2 < K2 || ldb2 < K2))) {
    danet_set_error("gemm_x6: K and leading dimensions must be multiples of 4 (K1=%d K2=%d)", K1, K2);
    return DANET_ERR_UNSUPPORTED;
  }
  DANET_CHECK_ARG(((((uintptr_t)A1 | (uintptr_t)A2) & 15) == 0) &&
                  ((((uintptr_t)B1h | (uintptr_t)B1m | (uintptr_t)B1l | (uintptr_t)B2h | (uintptr_t)B2m |
                     (uintptr_t)B2l) & 7) == 0), "gemm_x6: operand alignment (A 16 bytes, pieces 8 bytes)");
  static const bool once = [] {
    return hipFuncSetAttribute((const void*)gemm_x6_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               X6_SMEM_BYTES) == hipSuccess; }();
  (void)once;
  X6Args g;
  g.A[0] = A1; g.Bp[0][0] = B1h; g.Bp[0][1] = B1m; g.Bp[0][2] = B1l; g.lda[0] = lda1; g.ldb[0] = ldb1; g.K[0] = K1;
  g.A[1] = A2; g.Bp[1][0] = B2h; g.Bp[1][1] = B2m; g.Bp[1][2] = B2l; g.lda[1] = lda2; g.ldb[1] = ldb2; g.K[1] = K2;
  g.npair = K2 > 0 ? 2 : 1;
  g.M = M; g.N = N;
  const int nkt = cdiv(K1, XBK) + cdiv(K2, XBK);
  int s = x6_splitk(M, N, nkt);
  if (s > 1 && (N % 4 != 0 || ldc % 4 != 0 || ((uintptr_t)C & 15) != 0)) s = 1;   // (the reduce kernel is vectorised)
  if (s > 1) {
    const size_t need = (size_t)s * M * N * sizeof(float);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 15) != 0) {
      danet_set_error("gemm_x6: workspace %zu < %zu (or not 16-B aligned)", ws_bytes, need);
      return DANET_ERR_WORKSPACE;
    }
    g.C = (float*)ws; g.ldc = N;
  } else {
    g.C = C; g.ldc = ldc;
  }
  g.splitk = s;
  const int nt = cdiv(M, XBM) * cdiv(N, XBN);
  gemm_x6_nt_kernel<<<nt * s, 256, X6_SMEM_BYTES, (hipStream_t)stream>>>(g);
  DANET_CHECK_LAUNCH();
  if (s > 1) {
    const int grid = (int)min((int64_t)2048, cdiv64((int64_t)M * N / 4, 256));
    gemm_x6_reduce_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float*)ws, C, M, N, ldc, s);
    DANET_CHECK_LAUNCH();
  }
  return DANET_OK;
}
