// Contention probes for tools/contention_probe.py (diagnostic, not part of the library):
// a pure-MFMA burner and a pure memory streamer, each as 256-thread persistent workgroups.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern "C" __global__ __launch_bounds__(256) void mfma_burn_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a = threadIdx.x * 1e-6f, b = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}

extern "C" __global__ __launch_bounds__(256) void mem_stream_kernel(const float4* in, float* out,
                                                                   long n4, int iters) {
  float s = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (int it = 0; it < iters; ++it)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 v = in[i];
      s += v.x + v.y + v.z + v.w;
    }
  if (s == 123.456f) out[0] = s;
}

extern "C" int probe_mfma(void* stream, int wgs, float* out, int iters) {
  mfma_burn_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(out, iters);
  return (int)hipGetLastError();
}
extern "C" int probe_mem(void* stream, int wgs, const void* in, float* out, long n4, int iters) {
  mem_stream_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>((const float4*)in, out, n4, iters);
  return (int)hipGetLastError();
}

// MFMA burner with a duty cycle: `mf` back-to-back MFMAs (SHAPE 0: 32x32x2 = 64 clocks each,
// SHAPE 1: 16x16x4 = 32 clocks each) followed by `idle` x 64 clocks of s_sleep, repeated.  Does the
// GRANULARITY of a neighbour's MFMAs matter to a latency-bound kernel that shares the SIMD?
typedef float f32x4p __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_duty_kernel(float* out, int iters, int mf, int idle) {
  f32x16 a32[2]; f32x4p a16[4];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) a32[i][r] = 0.f;
  for (int i = 0; i < 4; ++i) a16[i] = (f32x4p){0.f, 0.f, 0.f, 0.f};
  const float a = threadIdx.x * 1e-3f + 0.5f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < mf; ++m) {
      if (SHAPE == 0) a32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a32[m & 1], 0, 0, 0);
      else a16[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[m & 3], 0, 0, 0);
    }
    for (int z = 0; z < idle; ++z) __builtin_amdgcn_s_sleep(1);
  }
  float s = a32[0][0] + a32[1][0] + a16[0][0] + a16[1][0] + a16[2][0] + a16[3][0];
  if (s == 123.456f) out[0] = s;
}
extern "C" int probe_mfma_duty(void* stream, int wgs, float* out, int shape, int iters, int mf, int idle) {
  if (shape == 0) mfma_duty_kernel<0><<<wgs, 256, 0, (hipStream_t)stream>>>(out, iters, mf, idle);
  else mfma_duty_kernel<1><<<wgs, 256, 0, (hipStream_t)stream>>>(out, iters, mf, idle);
  return (int)hipGetLastError();
}

// ---- CU-masked streams (tools/cumask_probe.py) ---------------------------------------------
extern "C" int probe_stream_create_cumask(const unsigned* mask, int nwords, void** stream) {
  hipStream_t s;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
  *stream = (void*)s;
  return (int)e;
}
extern "C" int probe_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }

// where does each workgroup run: out[b] = XCC_ID << 16 | HW_ID[15:0] (after `spin` clocks, so that
// a grid of <= #CUs workgroups is resident all at once)
extern "C" __global__ __launch_bounds__(256) void whereami_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
}
extern "C" int probe_whereami(void* stream, int wgs, unsigned* out, int spin) {
  whereami_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(out, spin);
  return (int)hipGetLastError();
}
