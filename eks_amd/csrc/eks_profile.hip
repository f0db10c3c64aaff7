// Optional per-kernel timing: when enabled, every kernel launch of the library is bracketed by a
// pair of hipEvents recorded on the caller's stream (no synchronisation at record time).
// eks_profile_drain() synchronises on the recorded events and returns (name, milliseconds) pairs.
// Used by bench.py for the `roofline` object; off by default and then costs one branch per launch.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "eks_internal.hpp"

namespace eks {

struct ProfEntry {
  const char* name;
  hipEvent_t a, b;
};
static bool g_prof_on = false;
static std::vector<ProfEntry> g_prof;
static std::mutex g_prof_mu;

ProfScope::ProfScope(const char* name, hipStream_t st) : name_(name), st_(st), live_(g_prof_on) {
  if (!live_) return;
  hipEventCreate(&a_);
  hipEventCreate(&b_);
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

int eks_profile_enable(int on) {
  eks::g_prof_on = on != 0;
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
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
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
