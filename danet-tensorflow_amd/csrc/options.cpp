// Option table of libdanet_hip.so (see options.h); no environment access anywhere in the library.
#include <atomic>
#include <string.h>
#include "danet_hip.h"
#include "options.h"

extern "C" void danet_set_error(const char* fmt, ...);

namespace {
struct OptDef { const char* name; int dflt; };
const OptDef kDefs[OPT_COUNT] = {
  {"gemm_dma", 3}, {"splitk_target", 512}, {"gemm_wgs", 512}, {"gemm_maxsplit", 8},
  {"gemm_yield", 16}, {"lstm_fwd_un", 0}, {"lstm_bwd_s", 0},
  {"lstm_bwd_u", 0}, {"lstm_spin_limit", 0}, {"lstm_fault_inject", 0}, {"lstm_xmap", -1},
  {"lstm_fwd_small", 1}, {"lstm_fwd_fused", -1}, {"lstm_bwd_twin_xcd", 1}, {"gemm_mfma16", 1},
  {"gemm_x6_plan", 0},
};
std::atomic<int> g_val[OPT_COUNT];
std::atomic<bool> g_init{false};
void ensure_init() {
  if (g_init.load(std::memory_order_acquire)) return;
  static std::atomic_flag once = ATOMIC_FLAG_INIT;
  if (!once.test_and_set()) {
    for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kDefs[i].dflt, std::memory_order_relaxed);
    g_init.store(true, std::memory_order_release);
  } else {
    while (!g_init.load(std::memory_order_acquire)) {}
  }
}
int find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kDefs[i].name) == 0) return i;
  return -1;
}
}  // namespace

int danet_opt(int id) {
  ensure_init();
  return g_val[id].load(std::memory_order_relaxed);
}

extern "C" int danet_set_option(const char* name, int value) {
  ensure_init();
  const int i = find(name);
  if (i < 0) { danet_set_error("unknown option '%s'", name ? name : "(null)"); return DANET_ERR_ARG; }
  g_val[i].store(value, std::memory_order_relaxed);
  return DANET_OK;
}
extern "C" int danet_get_option(const char* name, int* value) {
  ensure_init();
  const int i = find(name);
  if (i < 0 || !value) { danet_set_error("unknown option '%s'", name ? name : "(null)"); return DANET_ERR_ARG; }
  *value = g_val[i].load(std::memory_order_relaxed);
  return DANET_OK;
}
extern "C" void danet_reset_options(void) {
  ensure_init();
  for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kDefs[i].dflt, std::memory_order_relaxed);
}
extern "C" int danet_option_count(void) { return OPT_COUNT; }
extern "C" const char* danet_option_name(int index) {
  return (index >= 0 && index < OPT_COUNT) ? kDefs[index].name : nullptr;
}
