// fp32 products on the BF16 matrix cores (gfx950): C = A B^T with every fp32 operand written as
// hi + mid + lo, three bf16 pieces of 8 significant bits each (an EXACT decomposition: the pieces
// are successive truncations), and the six largest of the nine piece products summed in fp32
// accumulators:
//     A B^T ~ Ahi Bhi + (Ahi Bmid + Amid Bhi) + (Ahi Blo + Alo Bhi + Amid Bmid)
// The dropped terms are <= 3 * 2^-24 of |a||b| per product -- below the rounding an fp32 FMA chain
// commits (measured on the step's shapes, tools/bf16x_split_accuracy.py: 2.4e-7 max / 1.0e-7 rms
// relative to the float64 product against 6.6e-7 / 3.1e-7 for a float32 product).  What it buys:
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, six of them replace
// eight fp32 instructions' worth of k -> 2.7x the matrix-core throughput of the exact-fp32 kernels
// in gemm_f32.hip (2.5 PFLOP/s / 6 = 417 TFLOP/s nominal against 157).
//
// STATUS: PROTOTYPE, not part of libdanet_hip.so (round 4; tools/bench_gemm_x6.py builds it on
// demand and prints accuracy and time next to the product's exact-fp32 kernels).  Measured: correct;
// 165 TFLOP/s at 4096^3 (1.21x the exact-fp32 kernel) and 1.18x on cfg 4's projection, but no gain
// at cfg 2's shapes (projection 144 vs 138 us; the N = 600 products 0.52x without a stream-K
// schedule), and with ONE accumulator for all six terms the error at K = 2580 is 1.8e-6 against
// 6e-7 for the fp32 kernel (967 roundings of small terms into a large accumulator).  What a product
// version needs is listed in DESIGN.md 8.
//
// Scope: the NT form -- both operands K-contiguous, A [M][lda], B [N][ldb] -- with an
// optional second operand pair (K-concatenation), beta = 0, no bias: the products on the critical
// path of a train step (output projection with the transposed weight, dYc, dX).  Tile per
// workgroup, 128 x 128 x 16, 4 waves as 2 x 2, each 2 x 2 MFMA tiles of 32 x 32; register staging
// (the split is vector-ALU work on the way from global memory to LDS), two LDS stages of six
// piece images [128 rows][16 k] bf16 with the two 16-byte chunks of a row swapped on odd row
// pairs (conflict-free 16-byte fragment reads).
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
  const float* A[2]; const float* B[2];
  int lda[2], ldb[2], K[2];
  float* C;
  int M, N, ldc, npair;
};

// x = hi + mid + lo exactly; hi and mid are x and the remainder ROUNDED to 8 significant bits
// (add half an ulp, truncate), so |mid| <= 2^-9 |x|, |lo| <= 2^-17 |x| and the dropped products
// (mid lo, lo mid, lo lo) stay below 2^-25 |a||b|; with plain truncation they are 16x larger and
// all of one sign (measured 1.9e-6 against 6e-7 for the fp32 kernel).  lo has <= 8 significant
// bits left: its truncation is exact.
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

// byte offset of (row, 4-k group q in 0..3) inside a piece image: row-major 32 B rows, the two
// 16-byte chunks swapped when (row >> 2) is odd -> 8 consecutive rows of one chunk cover all banks
__device__ __forceinline__ int img_off(int row, int q) {
  const int c = (q >> 1) ^ ((row >> 2) & 1);
  return row * 32 + c * 16 + (q & 1) * 8;
}

// one float4 (row, k = 4 q .. 4 q + 3) -> three 8-byte LDS writes
__device__ __forceinline__ void stage_vec(char* img3 /* hi image; mid at +XIMG, lo at +2 XIMG */, int row, int q,
                                          f32x4 v) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split3(v[j], h[j], m[j], l[j]);
  const int off = img_off(row, q);
  *reinterpret_cast<u32x2*>(img3 + off) = (u32x2){pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
  *reinterpret_cast<u32x2*>(img3 + XIMG + off) = (u32x2){pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
  *reinterpret_cast<u32x2*>(img3 + 2 * XIMG + off) = (u32x2){pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
}

__device__ __forceinline__ bf16x8 frag(const char* img, int row, int kb) {
  const int c = kb ^ ((row >> 2) & 1);
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + row * 32 + c * 16));
}

__global__ __launch_bounds__(256, 2) void gemm_x6_nt_kernel(X6Args g) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + XBN - 1) / XBN;
  // consecutive workgroups share an A row panel (one XCD band each: b % 8 walks the bands)
  const int nt = ((g.M + XBM - 1) / XBM) * tiles_n;
  int bid = blockIdx.x;
  if ((nt & 7) == 0) bid = (bid & 7) * (nt >> 3) + (bid >> 3);
  const int m0 = (bid / tiles_n) * XBM, n0 = (bid % tiles_n) * XBN;

  // staging map: thread -> (row = tid / 4 + 64 i, q = tid % 4): 4 threads cover a row's 16 floats
  const int srow = tid >> 2, sq = tid & 3;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fi = lane & 31, kb = lane >> 5;
  int stage = 0;
  bool first = true;
  for (int p = 0; p < g.npair; ++p) {
    const float* __restrict__ Ap = g.A[p];
    const float* __restrict__ Bp = g.B[p];
    const int lda = g.lda[p], ldb = g.ldb[p], K = g.K[p];
    const int nk = (K + XBK - 1) / XBK;
    auto load = [&](int kt, f32x4 (&ra)[2], f32x4 (&rb)[2]) {
      const int k = kt * XBK + sq * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ar = m0 + srow + 64 * i, br = n0 + srow + 64 * i;
        ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ar < g.M && k < K) ra[i] = *reinterpret_cast<const f32x4*>(Ap + (size_t)ar * lda + k);
        if (br < g.N && k < K) rb[i] = *reinterpret_cast<const f32x4*>(Bp + (size_t)br * ldb + k);
      }
    };
    auto store = [&](int st, const f32x4 (&ra)[2], const f32x4 (&rb)[2]) {
      char* base = xsm + st * XSTAGE;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        stage_vec(base, srow + 64 * i, sq, ra[i]);
        stage_vec(base + 3 * XIMG, srow + 64 * i, sq, rb[i]);
      }
    };
    f32x4 ra[2], rb[2];
    load(0, ra, rb);
    if (!first) __syncthreads();          // the previous pair's last stage has been read by everyone
    store(stage, ra, rb);
    if (nk > 1) load(1, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const char* sb = xsm + stage * XSTAGE;
      // fragments of this k-tile: [piece][row tile]
      bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[pc][i] = frag(sb + pc * XIMG, wm * 64 + i * 32 + fi, kb);
          fb[pc][i] = frag(sb + (3 + pc) * XIMG, wn * 64 + i * 32 + fi, kb);
        }
      // the next tile goes to the other stage while this one is multiplied (it was last read
      // before the barrier that ended the previous iteration)
      if (kt + 1 < nk) store(stage ^ 1, ra, rb);
      if (kt + 2 < nk) load(kt + 2, ra, rb);
      // small terms first; the four tiles interleaved so that an accumulator is reused every 4th MFMA
#define X6_TERM(PA, PB)                                                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i], fb[PB][j], acc[i][j], 0, 0, 0);
      X6_TERM(2, 0) X6_TERM(0, 2) X6_TERM(1, 1) X6_TERM(1, 0) X6_TERM(0, 1) X6_TERM(0, 0)
#undef X6_TERM
      __syncthreads();
      stage ^= 1;
    }
    first = false;
  }

  // C/D layout 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The tile leaves
  // through the (now free) LDS in two halves of 64 rows as 16-byte stores of full 512-byte row
  // segments when the destination allows it; 4-byte stores otherwise.
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
      // 64 rows x 32 float4: thread -> (row = tid / 32 + 8 it, c4 = tid % 32)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rr = (tid >> 5) + 8 * it, c4 = tid & 31;
        const int row = m0 + half * 64 + rr;
        if (row < g.M)
          *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + n0 + c4 * 4) =
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
        if (row < g.M && col < g.N) g.C[(size_t)row * g.ldc + col] = acc[i][j][r];
      }
    }
}

// out [N][M] = in [M][N]^T (the output projection's weight, once per step): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_f32_kernel(int M, int N, const float* __restrict__ in, int ldi,
                                                            float* __restrict__ out, int ldo) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (m0 + r < M && n0 + tx < N) t[r][tx] = in[(size_t)(m0 + r) * ldi + n0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (n0 + r < N && m0 + tx < M) out[(size_t)(n0 + r) * ldo + m0 + tx] = t[tx][r];
}

extern "C" int danet_transpose_f32(danet_stream_t stream, int M, int N, const float* in, int ldi,
                                   float* out, int ldo) {
  DANET_CHECK_ARG(M > 0 && N > 0 && in && out && ldi >= N && ldo >= M, "transpose: bad args");
  dim3 grid((unsigned)cdiv(N, 32), (unsigned)cdiv(M, 32));
  transpose_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(M, N, in, ldi, out, ldo);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_gemm_x6_nt_supported(int M, int N, int K1, int lda1, int ldb1, int K2, int lda2,
                                          int ldb2, int ldc) {
  if (M <= 0 || N <= 0 || K1 <= 0 || K2 < 0 || ldc < N) return 0;
  if (K1 % 4 || lda1 % 4 || ldb1 % 4 || lda1 < K1 || ldb1 < K1) return 0;
  if (K2 > 0 && (K2 % 4 || lda2 % 4 || ldb2 % 4 || lda2 < K2 || ldb2 < K2)) return 0;
  return 1;
}

extern "C" int danet_gemm_x6_nt(danet_stream_t stream, int M, int N,
                                int K1, const float* A1, int lda1, const float* B1, int ldb1,
                                int K2, const float* A2, int lda2, const float* B2, int ldb2,
                                float* C, int ldc) {
  DANET_CHECK_ARG(A1 && B1 && C && (K2 == 0 || (A2 && B2)), "gemm_x6: null operand");
  if (!danet_gemm_x6_nt_supported(M, N, K1, lda1, ldb1, K2, lda2, ldb2, ldc)) {
    danet_set_error("gemm_x6: K and leading dimensions must be multiples of 4 (K1=%d K2=%d)", K1, K2);
    return DANET_ERR_UNSUPPORTED;
  }
  DANET_CHECK_ARG(((((uintptr_t)A1 | (uintptr_t)B1 | (uintptr_t)A2 | (uintptr_t)B2) & 15) == 0),
                  "gemm_x6: operands must be 16-byte aligned");
  DANET_CHECK_ARG((size_t)M * lda1 < ((size_t)1 << 40), "gemm_x6: shape");
  static const bool once = [] {
    return hipFuncSetAttribute((const void*)gemm_x6_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               X6_SMEM_BYTES) == hipSuccess; }();
  (void)once;
  X6Args g;
  g.A[0] = A1; g.B[0] = B1; g.lda[0] = lda1; g.ldb[0] = ldb1; g.K[0] = K1;
  g.A[1] = A2; g.B[1] = B2; g.lda[1] = lda2; g.ldb[1] = ldb2; g.K[1] = K2;
  g.npair = K2 > 0 ? 2 : 1;
  g.C = C; g.M = M; g.N = N; g.ldc = ldc;
  const int nt = cdiv(M, XBM) * cdiv(N, XBN);
  gemm_x6_nt_kernel<<<nt, 256, X6_SMEM_BYTES, (hipStream_t)stream>>>(g);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}
