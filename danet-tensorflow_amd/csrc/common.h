// Shared helpers for libdanet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "danet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

#define DANET_WAVE 64

// compute units of the current device (csrc/lstm.hip; 256 when no device answers)
int dn_num_cus();

// ---- fill lists: every "not yet published" pattern / zero block of one call in ONE launch (csrc/lstm.hip)
#define FILL_MAXSEG 32
struct FillArgs {
  void* ptr[FILL_MAXSEG];
  unsigned long long n16[FILL_MAXSEG];   // 16-byte words per segment
  unsigned value[FILL_MAXSEG];
  int nseg;
};
// thread `tid` of `stride` fills its share of every segment
__device__ __forceinline__ void fill_share(const FillArgs& a, unsigned long long tid, unsigned long long stride) {
  typedef unsigned fv4u __attribute__((__vector_size__(16)));
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const unsigned v = a.value[sgi];
    const fv4u w = {v, v, v, v};
    fv4u* p = reinterpret_cast<fv4u*>(a.ptr[sgi]);
    for (unsigned long long i = tid; i < a.n16[sgi]; i += stride) p[i] = w;
  }
}
// csrc/pointwise.hip: danet_center with an optional rider (taken only by the one-launch form:
// *rider_taken says whether the launch carried it)
int dn_center(hipStream_t stream, int B, int T, int D, const float* in, int in_layout, int ld_in, float* out,
              int out_layout, int ld_out, float* mean, const FillArgs* rider, bool* rider_taken);

// fp32 products on the bf16 matrix cores (csrc/gemm_x6.hip; the recurrent half of
// lstm_fwd_fx_kernel): every fp32 value is EXACTLY hi + mid + lo with three bf16 pieces of 8
// significant bits.  (x0, x1) -> the packed bf16 pairs of their pieces (x0 in the low half):
// v_cvt_pk_bf16_f32 rounds to nearest even, the remainders are exact in fp32 and the third piece
// has <= 8 significant bits left.
//
// The remainder x - bf16(x) is formed by ONE instruction per value: v_dot2c_f32_bf16 computes
// d += a.lo * b.lo + a.hi * b.hi in fp32 with a single rounding, so with the piece pair as `a` and
// (-1, 0) / (0, -1) as `b` it returns x0 - hi0 / x1 - hi1 -- exact, like the subtraction it replaces
// (tools/csrc/dot2_split_exact.hip: bit-identical pieces for 4 x 8.4 M pairs over every binade,
// denormals included) -- without the two instructions that expanded the pair to fp32 first: 7 vector
// instructions per pair of values instead of 9, and on gfx950 every vector instruction of a matrix
// loop is paid in matrix-core time.  One difference, for non-finite data only: a
// value whose bf16 rounding is +-Inf (|x| >= 3.39e38, Inf) makes its PARTNER's remainder NaN
// (Inf * 0); in the subtracting form only the value itself produced Inf - Inf.
// The multipliers live in scalar registers behind an asm barrier: written as constants, the compiler
// encodes (-1, 0) as the inline constant -1.0, which the instruction reads as 0xbf800000 = (0, -1).
#ifdef DANET_SPLIT_SUBTRACT
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){x0, x1}, bf16x2v));
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){r0, r1}, bf16x2v));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){s0, s1}, bf16x2v));
}
#else
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  uint32_t c0 = 0x0000bf80u, c1 = 0xbf800000u;          // bf16 pairs (-1, 0) and (0, -1)
  asm("" : "+s"(c0));
  asm("" : "+s"(c1));
  const bf16x2v n0 = __builtin_bit_cast(bf16x2v, c0), n1 = __builtin_bit_cast(bf16x2v, c1);
  const bf16x2v hv = __builtin_convertvector((f32x2v){x0, x1}, bf16x2v);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hv, n0, x0, false);
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(hv, n1, x1, false);
  const bf16x2v mv = __builtin_convertvector((f32x2v){r0, r1}, bf16x2v);
  const float s0 = __builtin_amdgcn_fdot2_f32_bf16(mv, n0, r0, false);
  const float s1 = __builtin_amdgcn_fdot2_f32_bf16(mv, n1, r1, false);
  h = __builtin_bit_cast(uint32_t, hv);
  m = __builtin_bit_cast(uint32_t, mv);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){s0, s1}, bf16x2v));
}
#endif

extern "C" void danet_set_error(const char* fmt, ...);

#define DANET_CHECK_ARG(cond, ...)                         \
  do {                                                     \
    if (!(cond)) {                                         \
      danet_set_error(__VA_ARGS__);                        \
      return DANET_ERR_ARG;                                \
    }                                                      \
  } while (0)

#define DANET_CHECK_HIP(expr)                                               \
  do {                                                                      \
    hipError_t e__ = (expr);                                                \
    if (e__ != hipSuccess) {                                                \
      danet_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                  \
      return DANET_ERR_LAUNCH;                                              \
    }                                                                       \
  } while (0)

#define DANET_CHECK_LAUNCH() DANET_CHECK_HIP(hipGetLastError())

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- wave64 reductions -----------------------------------------------------
// Sum over the 64 lanes, the same value in every lane.  Round 6: data-parallel-primitive adds inside a row
// of 16 lanes (v_add_f32_dpp: quad swaps, half-row and row mirrors), two row broadcasts and ONE v_readlane --
// seven vector instructions, no LDS -- instead of six ds_bpermute round trips through the LDS crossbar per value
// (__shfl_xor): the head kernels reduce 33-113 accumulators per wave in their epilogues, 656 of the fused
// separator + loss forward's 3 017 instructions were ds_bpermute.  Fixed order: deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shuffled(float v) {
  // (rows outside ROW_MASK, and lanes whose source lane does not exist, read 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_shuffled<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_shuffled<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]: lane ^ 2      -> every quad uniform
  v += dpp_shuffled<0x141, 0xF>(v);    // row_half_mirror: the other quad of the half-row
  v += dpp_shuffled<0x140, 0xF>(v);    // row_mirror: the other half of the row -> every row of 16 uniform
  v += dpp_shuffled<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3: row 0 + row 1, row 2 + row 3
  v += dpp_shuffled<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// the same for N values at once, step by step ACROSS the values: consecutive instructions are independent,
// so the two wait states a DPP read needs behind the vector write it depends on cost nothing (one value at a
// time the compiler fills them with s_nop: 814 of them in the fused separator + loss forward's epilogue)
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_shuffled<0xB1, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_shuffled<0x4E, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_shuffled<0x141, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_shuffled<0x140, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_shuffled<0x142, 0xA>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_shuffled<0x143, 0xC>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x multiple of 64 (<= 1024); result valid in all
// threads.  `red` = LDS scratch of >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// accurate variants used on the parity path (1e-4 relative through T steps)
__device__ __forceinline__ float sigmoid_acc(float x) {
  return 1.0f / (1.0f + expf(-x));
}

// Hardware-exponential variants for the latency-critical LSTM gate math
// (v_exp_f32 / v_rcp_f32: ~1 ulp each; |relative error| of the results ~1e-6,
// two orders below the 1e-4 parity bar -- checked by the GPU parity tests).
// -DDANET_LSTM_ACCURATE_MATH restores libm expf/tanhf.
__device__ __forceinline__ float sigmoid_hw(float x) {
#ifdef DANET_LSTM_ACCURATE_MATH
  return sigmoid_acc(x);
#else
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));      // v_rcp_f32 (HIP's __frcp_rn is the 11-instruction IEEE division)
#endif
}
__device__ __forceinline__ float tanh_hw(float x) {
#ifdef DANET_LSTM_ACCURATE_MATH
  return tanhf(x);
#else
  // tanh(x) = sign(x) * (1 - e) / (1 + e), e = exp(-2|x|) in (0, 1]: no overflow,
  // no cancellation for large |x|; for tiny |x| the (1 - e) cancellation is
  // absolute-error ~1e-8, relative to the O(1) tensor scale used by the bar
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
#endif
}
