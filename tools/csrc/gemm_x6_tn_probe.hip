// Probe (GPU box): time of the TN group kernel of csrc/gemm_x6.hip with parts of its k-step removed
// (-DX6T_NO_SPLIT / NO_LDSREAD / NO_WRITE / NO_GLOBAL): which pipe bounds it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idanet-tensorflow_amd/csrc [-DX6T_...] -o tn_probe tools/csrc/gemm_x6_tn_probe.hip
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
extern "C" void danet_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
hipEvent_t dn_take_stop_event() { return nullptr; }
#include "../../danet-tensorflow_amd/csrc/gemm_x6.hip"

int main() {
  const int K = 4096;
  const int Ms[4] = {600, 600, 300, 300}, N = 1200;
  danet_gemm_problem_t pr[4];
  double fl = 0;
  for (int i = 0; i < 4; ++i) {
    float *A, *B, *C;
    hipMalloc(&A, (size_t)K * 600 * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)Ms[i] * N * 4);
    hipMemset(A, 0, (size_t)K * 600 * 4); hipMemset(B, 0, (size_t)K * N * 4);
    pr[i] = {A, 600, B, N, C, N, Ms[i], N, nullptr, 0.f};
    fl += 2.0 * Ms[i] * N * K;
  }
  void* ws; const size_t wsb = (size_t)64 << 20; hipMalloc(&ws, wsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (danet_gemm_x6_tn_grouped(nullptr, K, 4, pr, ws, wsb)) return 1;
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < 20; ++i) danet_gemm_x6_tn_grouped(nullptr, K, 4, pr, ws, wsb);
  hipEventRecord(e1, nullptr);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%8.1f us  %6.1f TFLOP/s\n", ms / 20 * 1e3, fl / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
