// Optional per-kernel timing: when enabled, every kernel launch of the library is bracketed by a
// pair of hipEvents recorded on the caller's stream (no synchronisation at record time).
// eks_profile_drain() synchronises on the recorded events and returns (name, milliseconds) pairs.
// Used by bench.py for the `roofline` object; off by default and then costs one branch per launch.
// Level 2 brackets only the step's two roofline kernels - the smoother's replay kernels (HBM-bound) and the NLL
// grid kernel (VALU-bound, the longest) - four event records per step instead of twelve inside bench.py's
// timed region.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "eks_internal.hpp"

namespace eks {

namespace {
struct KnobTable {
  bool set[KNOB_COUNT];
  int val[KNOB_COUNT];
  KnobTable() { load(); }
  int load() {
    static const char* const names[KNOB_COUNT] = {
        "EKS_SMOOTH_UNFUSED", "EKS_SUMMARIZE_REVERSE", "EKS_REPLAY_FORWARD", "EKS_REPLAY_RECOMPUTE", "EKS_SCAN_CH",
        "EKS_DENSE_CHUNK", "EKS_NLL_NCL", "EKS_NLL_CHUNK", "EKS_NLL_CHUNK0", "EKS_NLL_WPB",
        "EKS_DENSE_LEGACY", "EKS_NLL_GRAD_UNFUSED",
        "EKS_NLL_GRAD_CHUNK", "EKS_DENSE_TREE_SCAN", "EKS_DENSE_DUAL_GRAD", "EKS_NLL_LEGACY", "EKS_MED_ROWS", "EKS_DW_CHUNK", "EKS_ADAM_PER_ITERATION", "EKS_MED_FINISH_THREADS", "EKS_MED_BRACKET_THREADS", "EKS_NLL_NOLAG", "EKS_NLL_GRAD_TREE", "EKS_ADAM_STREAM", "EKS_ADAM_LAG_RHO_PPM", "EKS_ADAM_LAG_HEAD"};
    int n = 0;
    for (int i = 0; i < KNOB_COUNT; ++i) {
      const char* v = getenv(names[i]);
      set[i] = v != nullptr;
      val[i] = v ? atoi(v) : 0;
      n += set[i];
    }
    return n;
  }
};
KnobTable& knob_table() {
  static KnobTable t;                // read once, on first use (eks_knobs_reload: the tests' hook)
  return t;
}
}  // namespace

bool knob_set(Knob k) { return knob_table().set[k]; }
int knob_int(Knob k, int dflt) { return knob_table().set[k] ? knob_table().val[k] : dflt; }

struct ProfEntry {
  const char* name;
  hipEvent_t a, b;
};
static int g_prof_level = 0;          // 0 off, 1 every scope, 2 only the *_replay scopes
static std::vector<ProfEntry> g_prof;
static std::vector<hipEvent_t> g_pool;  // events are reused: creating one costs microseconds
static std::mutex g_prof_mu;

static hipEvent_t take_event() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

static bool scope_live(const char* name) {
  if (g_prof_level == 1) return true;
  if (g_prof_level != 2) return false;
  // the roofline kernels of a step: the smoother's replay (HBM-bound) and the NLL grid (VALU-bound, the longest)
  const size_t n = strlen(name);
  return (n >= 7 && strcmp(name + n - 7, "_replay") == 0) || strcmp(name, "diag_nll_summarize") == 0;
}

ProfScope::ProfScope(const char* name, hipStream_t st) : name_(name), st_(st), live_(scope_live(name)) {
  if (!live_) return;
  a_ = take_event();
  b_ = take_event();
  hipEventRecord(a_, st_);
}
ProfScope::~ProfScope() {
  if (!live_) return;
  hipEventRecord(b_, st_);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back({name_, a_, b_});
}

}  // namespace eks

extern "C" {

int eks_knobs_reload(void) { return eks::knob_table().load(); }

int eks_profile_enable(int on) {
  eks::g_prof_level = on < 0 ? 0 : on > 2 ? 1 : on;
  return EKS_OK;
}

int eks_profile_drain(char* names, size_t names_bytes, float* ms, int32_t max_n) {
  std::lock_guard<std::mutex> lk(eks::g_prof_mu);
  int n = 0;
  size_t off = 0;
  for (auto& e : eks::g_prof) {
    hipEventSynchronize(e.b);
    float t = 0.f;
    hipEventElapsedTime(&t, e.a, e.b);
    eks::g_pool.push_back(e.a);
    eks::g_pool.push_back(e.b);
    const size_t len = strlen(e.name) + 1;
    if (n < max_n && names && ms && off + len <= names_bytes) {
      memcpy(names + off, e.name, len);
      off += len;
      ms[n++] = t;
    }
  }
  eks::g_prof.clear();
  return n;
}

}  // extern "C"
