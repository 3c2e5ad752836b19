// STFT / iSTFT front-end and back-end (gfx950).
//
// danet_stft replaces scipy.signal.stft(x, window=FFT_WND, nperseg=N,
// noverlap=N-S)[2].astype(complex64).T (reference app/utils.py:117-122,
// app/datasets/TIMIT/process.py:93-97, app/datasets/WSJ0/process.py:175-179):
// boundary='zeros' (N/2 zeros each side), padded=True (zero tail so the last
// frame is full), frame t = extended samples [t*S, t*S+N) (integer framing is
// exact), times the window, N-point real FFT, times 1/sum(window).
// One workgroup per frame: the frame is staged in LDS (bit-reversed), the
// log2(N) radix-2 stages run in LDS with an LDS twiddle table (sincospi, so
// twiddles are correctly rounded), and the N/2+1 bins stream out as float2.
//
// danet_istft replaces utils.istft (app/utils.py:53-75): per-frame inverse real
// FFT (float64 like numpy's accumulators) times the window into `ws`, then a
// gather pass overlap-adds the <= N/S frames covering each output sample and
// divides by the overlap-added window^2 -- deterministic, no atomics.
#include "common.h"

template <typename T>
__device__ __forceinline__ void fft_stages(T* re, T* im, const T* twr, const T* twi, int N,
                                           int logN) {
  for (int s = 1; s <= logN; ++s) {
    const int half = 1 << (s - 1);
    for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {
      const int grp = i >> (s - 1), pos = i & (half - 1);
      const int i0 = (grp << s) + pos, i1 = i0 + half;
      const int tw = pos << (logN - s);
      const T wr = twr[tw], wi = twi[tw];
      const T xr = re[i1] * wr - im[i1] * wi;
      const T xi = re[i1] * wi + im[i1] * wr;
      const T ar = re[i0], ai = im[i0];
      re[i1] = ar - xr; im[i1] = ai - xi;
      re[i0] = ar + xr; im[i0] = ai + xi;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int x, int logN) { return (int)(__brev((unsigned)x) >> (32 - logN)); }

static int ilog2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return ((1 << l) == n) ? l : -1;
}

extern "C" int danet_stft_num_frames(int64_t Ls, int N, int S) {
  if (N <= 0 || S <= 0 || S > N || Ls < N) return DANET_ERR_ARG;   // scipy raises for Ls < N
  const int64_t ext = Ls + 2 * (int64_t)(N / 2);
  const int64_t nadd = ((-(ext - N)) % S + S) % S % N;
  return (int)((ext + nadd - N) / S + 1);
}

// 1/sum(window) is recomputed by every workgroup (N <= 4096 adds, float64 like
// scipy's `1.0 / win.sum()**2` then sqrt) so the call stays allocation-free and
// needs no host round trip.
__global__ void stft_kernel(int64_t Ls, int N, int logN, int S, int T,
                                      const float* __restrict__ x,
                                      const float* __restrict__ window,
                                      float2* __restrict__ out) {
  extern __shared__ float sm[];
  float* re = sm;
  float* im = sm + N;
  float* twr = sm + 2 * N;
  float* twi = twr + N / 2;
  __shared__ double wsum_s;
  const int t = blockIdx.x, sig = blockIdx.y;
  const float* xs = x + (int64_t)sig * Ls;
  const int64_t start = (int64_t)t * S - N / 2;
  // sum(window) in float64 like scipy's win.sum() on the float32 window upcast
  // by `1.0 / win.sum()**2` (result is rounded to float32 when applied)
  double ws = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) ws += (double)window[i];
  ws = wave_sum_d(ws);
  if ((threadIdx.x & 63) == 0) ((double*)re)[threadIdx.x >> 6] = ws;   // N >= 64 -> room
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += ((double*)re)[w];
    wsum_s = tot;
  }
  __syncthreads();
  const float scale = (float)(1.0 / wsum_s);
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const int64_t pos = start + i;
    float v = 0.f;
    if (pos >= 0 && pos < Ls) v = xs[pos] * window[i];
    const int r = bitrev(i, logN);
    re[r] = v;
    im[r] = 0.f;
  }
  for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
    float sn, cs;
    sincospif(-2.0f * (float)k / (float)N, &sn, &cs);
    twr[k] = cs; twi[k] = sn;
  }
  __syncthreads();
  fft_stages<float>(re, im, twr, twi, N, logN);
  const int F = N / 2 + 1;
  float2* o = out + ((int64_t)sig * T + t) * F;
  for (int k = threadIdx.x; k < F; k += blockDim.x) o[k] = make_float2(re[k] * scale, im[k] * scale);
}

extern "C" int danet_stft(danet_stream_t stream_, int n_sig, int64_t Ls, int N, int S,
                          const float* x, const float* window, float* out_c64) {
  hipStream_t stream = (hipStream_t)stream_;
  const int logN = ilog2_exact(N);
  DANET_CHECK_ARG(n_sig > 0 && n_sig <= 65535, "stft: n_sig");
  DANET_CHECK_ARG(logN >= 6 && logN <= 12, "stft: N must be a power of two in [64, 4096]");
  DANET_CHECK_ARG(S > 0 && S <= N, "stft: stride");
  DANET_CHECK_ARG(x && window && out_c64, "stft: null pointer");
  const int T = danet_stft_num_frames(Ls, N, S);
  if (T < 0) {
    danet_set_error("stft: window is longer than input signal (Ls=%lld < N=%d)", (long long)Ls, N);
    return DANET_ERR_ARG;
  }
  const int threads = N / 2 < 64 ? 64 : (N / 2 > 256 ? 256 : N / 2);
  const size_t lds = (size_t)3 * N * sizeof(float);
  dim3 grid(T, n_sig);
  stft_kernel<<<grid, threads, lds, stream>>>(Ls, N, logN, S, T, x, window,
                                                        (float2*)out_c64);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// ------------------------------------------------------------------- iSTFT
__global__ void istft_frames_kernel(int T, int N, int logN, int F,
                                    const float2* __restrict__ X,
                                    const float* __restrict__ window,
                                    double* __restrict__ frames /* [n_sig][used][N] */,
                                    int used) {
  extern __shared__ double smd[];
  double* re = smd;
  double* im = smd + N;
  double* twr = smd + 2 * N;
  double* twi = twr + N / 2;
  const int n = blockIdx.x, sig = blockIdx.y;
  const float2* Xf = X + ((int64_t)sig * T + n) * F;
  // Hermitian extension; imaginary parts of DC and Nyquist are ignored (irfft)
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    double r, i;
    if (k <= N / 2) {
      const float2 v = Xf[k];
      r = v.x; i = (k == 0 || k == N / 2) ? 0.0 : (double)v.y;
    } else {
      const float2 v = Xf[N - k];
      r = v.x; i = -(double)v.y;
    }
    const int rv = bitrev(k, logN);
    re[rv] = r; im[rv] = i;
  }
  for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
    double sn, cs;
    sincospi(2.0 * (double)k / (double)N, &sn, &cs);   // exp(+2 pi i k / N)
    twr[k] = cs; twi[k] = sn;
  }
  __syncthreads();
  fft_stages<double>(re, im, twr, twi, N, logN);
  double* o = frames + ((int64_t)sig * used + n) * N;
  const double inv = 1.0 / (double)N;
  for (int m = threadIdx.x; m < N; m += blockDim.x)
    o[m] = re[m] * inv * (double)window[m];                 // utils.py:71
}

__global__ void istft_ola_kernel(int T, int N, int S, int used,
                                 const double* __restrict__ frames,
                                 const float* __restrict__ window, double* __restrict__ out) {
  const int64_t len = (int64_t)T * S;
  const int sig = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    // frames n with n*S <= i < n*S + N, n < used
    int64_t n_hi = i / S;
    if (n_hi > used - 1) n_hi = used - 1;
    int64_t n_lo = (i - N + S) / S;   // ceil((i-N+1)/S) for i-N+1 > 0
    if (i - N + 1 <= 0) n_lo = 0;
    double acc = 0.0, wsum = 0.0;
    for (int64_t n = n_lo; n <= n_hi; ++n) {       // ascending n = reference add order
      const int m = (int)(i - n * S);
      if (m < 0 || m >= N) continue;
      acc += frames[((int64_t)sig * used + n) * N + m];
      const double w = (double)window[m];
      wsum += w * w;                                // utils.py:72
    }
    out[(int64_t)sig * len + i] = (wsum != 0.0) ? acc / wsum : acc;   // utils.py:73-74
  }
}

static int istft_used_frames(int T, int N, int S) {
  // len(range(0, T*S - N, S))
  const int64_t stop = (int64_t)T * S - N;
  if (stop <= 0) return 0;
  return (int)((stop + S - 1) / S);
}

size_t dn_ws_istft(int n_sig, int T, int N, int S) {
  const int used = istft_used_frames(T, N, S);
  return (size_t)n_sig * (used > 0 ? used : 1) * N * sizeof(double);
}

extern "C" int danet_istft(danet_stream_t stream_, int n_sig, int T, int N, int S,
                           const float* X_c64, const float* window, double* out, void* ws,
                           size_t ws_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  const int logN = ilog2_exact(N);
  DANET_CHECK_ARG(n_sig > 0 && n_sig <= 65535 && T > 0, "istft: shape");
  DANET_CHECK_ARG(logN >= 6 && logN <= 11, "istft: N must be a power of two in [64, 2048]");
  DANET_CHECK_ARG(S > 0 && S <= N, "istft: stride");
  DANET_CHECK_ARG(X_c64 && window && out, "istft: null pointer");
  if (!ws || ws_bytes < dn_ws_istft(n_sig, T, N, S)) {
    danet_set_error("istft: workspace too small");
    return DANET_ERR_WORKSPACE;
  }
  const int used = istft_used_frames(T, N, S);
  const int F = N / 2 + 1;
  if (used > 0) {
    const int threads = N / 2 < 64 ? 64 : (N / 2 > 256 ? 256 : N / 2);
    const size_t lds = (size_t)3 * N * sizeof(double);
    dim3 grid(used, n_sig);
    istft_frames_kernel<<<grid, threads, lds, stream>>>(T, N, logN, F, (const float2*)X_c64,
                                                        window, (double*)ws, used);
    DANET_CHECK_LAUNCH();
  }
  const int64_t len = (int64_t)T * S;
  dim3 grid2((unsigned)min((int64_t)1024, cdiv64(len, 256)), n_sig);
  istft_ola_kernel<<<grid2, 256, 0, stream>>>(T, N, S, used, (const double*)ws, window, out);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}
