// danet_workspace_bytes: the ONE scratch-size query of the ABI (include/danet_hip.h).  Every entry
// point that takes `ws` names its DANET_WS_* op and dims there; the per-kernel size functions live
// next to the kernels that use the scratch (dn_ws_*, C++ linkage, not part of the ABI).
#include <stddef.h>
#include <stdint.h>
#include "danet_hip.h"

extern "C" void danet_set_error(const char* fmt, ...);

size_t dn_ws_istft(int n_sig, int T, int N, int S);
size_t dn_ws_gemm(int M, int N, int K);
size_t dn_ws_gemm_streamk(int M, int N, int K);
size_t dn_ws_colsum(int M, int N);
size_t dn_ws_lstm(int T, int B, int H, int ndir);
size_t dn_ws_attractor_truth(int B, int C, int64_t N, int E);
size_t dn_ws_attractor_anchor(int B, int C, int64_t N, int E, int A);
size_t dn_ws_separate_bwd(int B, int C, int64_t N, int E);
size_t dn_ws_separate_pit(int B, int C, int64_t N, int E);
size_t dn_ws_separate_pit_records(int B, int64_t N);
size_t dn_ws_pit_mse(int B, int C, int64_t N);
int dn_center_mean_elems(int B);
size_t dn_ws_gemm_x6(int M, int N, int K1, int K2);
size_t dn_ws_gemm_pack(int N, int K);
size_t dn_ws_gemm_x6_tn(long long sum_mn, int tiles, int K);
size_t dn_ws_separate_pit_grad(int B, int C, int64_t N, int E);

extern "C" size_t danet_workspace_bytes(int op, const int64_t* d, int n) {
  static const int kDims[DANET_WS_COUNT] = {4, 3, 3, 2, 4, 4, 5, 4, 4, 2, 3, 1, 4, 2, 3, 4};
  if (op < 0 || op >= DANET_WS_COUNT || !d || n != kDims[op]) {
    danet_set_error("workspace_bytes: op %d takes %d dims, got %d", op,
                    (op >= 0 && op < DANET_WS_COUNT) ? kDims[op] : -1, n);
    return (size_t)-1;
  }
  for (int i = 0; i < n; ++i)
    if (d[i] < 0 || (d[i] > 0x7fffffff && !(i == 2 && op >= DANET_WS_ATTRACTOR_TRUTH) &&
                     !(i == 1 && op == DANET_WS_SEPARATE_PIT_RECORDS) &&
                     !(i == 0 && op == DANET_WS_GEMM_X6_TN))) {
      danet_set_error("workspace_bytes: dim %d of op %d out of range", i, op);
      return (size_t)-1;
    }
  switch (op) {
    case DANET_WS_ISTFT: return dn_ws_istft((int)d[0], (int)d[1], (int)d[2], (int)d[3]);
    case DANET_WS_GEMM: return dn_ws_gemm((int)d[0], (int)d[1], (int)d[2]);
    case DANET_WS_GEMM_STREAMK: return dn_ws_gemm_streamk((int)d[0], (int)d[1], (int)d[2]);
    case DANET_WS_COLSUM: return dn_ws_colsum((int)d[0], (int)d[1]);
    case DANET_WS_LSTM: return dn_ws_lstm((int)d[0], (int)d[1], (int)d[2], (int)d[3]);
    case DANET_WS_ATTRACTOR_TRUTH: return dn_ws_attractor_truth((int)d[0], (int)d[1], d[2], (int)d[3]);
    case DANET_WS_ATTRACTOR_ANCHOR:
      return dn_ws_attractor_anchor((int)d[0], (int)d[1], d[2], (int)d[3], (int)d[4]);
    case DANET_WS_SEPARATE_BWD: return dn_ws_separate_bwd((int)d[0], (int)d[1], d[2], (int)d[3]);
    case DANET_WS_SEPARATE_PIT: return dn_ws_separate_pit((int)d[0], (int)d[1], d[2], (int)d[3]);
    case DANET_WS_SEPARATE_PIT_RECORDS: return dn_ws_separate_pit_records((int)d[0], d[1]);
    case DANET_WS_PIT_MSE: return dn_ws_pit_mse((int)d[0], (int)d[1], d[2]);
    case DANET_WS_CENTER_MEAN: return (size_t)dn_center_mean_elems((int)d[0]) * sizeof(float);
    case DANET_WS_GEMM_X6: return dn_ws_gemm_x6((int)d[0], (int)d[1], (int)d[2], (int)d[3]);
    case DANET_WS_GEMM_PACK: return dn_ws_gemm_pack((int)d[0], (int)d[1]);
    case DANET_WS_GEMM_X6_TN: return dn_ws_gemm_x6_tn((long long)d[0], (int)d[1], (int)d[2]);
    case DANET_WS_SEPARATE_PIT_GRAD: return dn_ws_separate_pit_grad((int)d[0], (int)d[1], d[2], (int)d[3]);
  }
  return (size_t)-1;
}
