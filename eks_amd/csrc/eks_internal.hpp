// Internal declarations shared by the translation units of libeks_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/eks_hip.h"

namespace eks {

struct DiagModel;

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int hip_status(hipError_t e) { return e == hipSuccess ? EKS_OK : EKS_ERR_HIP_BASE - (int)e; }

// MI355X deals the workgroups of a launch to its 8 XCDs round-robin (block b runs on XCD b % 8) and every XCD
// has its own L2.  Launches whose NEIGHBOURING blocks touch the same cache lines (pieces of one line, shared rows)
// want those blocks on one XCD: this maps the hardware block index to a logical one such that XCD x works on the
// x-th contiguous eighth of the logical range (a bijection for any grid size).
__device__ __forceinline__ int xcd_contiguous_block(int b, int nb) {
  const int x = b & 7, i = b >> 3, q = nb >> 3, r = nb & 7;
  return x * q + (x < r ? x : r) + i;
}

// A/B tuning knobs: environment variables read ONCE, the first time any entry
// point asks (thread-safe function-local static), never in the per-call host path - the library stays
// re-entrant and a hot loop of eks_smooth calls does not walk the environment block.
enum Knob {
  KNOB_SMOOTH_UNFUSED, KNOB_SUMMARIZE_REVERSE, KNOB_REPLAY_FORWARD, KNOB_REPLAY_RECOMPUTE, KNOB_SCAN_CH,
  KNOB_DENSE_CHUNK, KNOB_NLL_NCL, KNOB_NLL_CHUNK, KNOB_NLL_CHUNK0, KNOB_NLL_WPB,
  KNOB_DENSE_LEGACY, KNOB_NLL_GRAD_UNFUSED, KNOB_NLL_GRAD_CHUNK,
  KNOB_DENSE_TREE_SCAN, KNOB_DENSE_DUAL_GRAD, KNOB_NLL_LEGACY, KNOB_MED_ROWS, KNOB_DW_CHUNK, KNOB_ADAM_PER_ITERATION, KNOB_MED_FINISH_THREADS, KNOB_MED_BRACKET_THREADS, KNOB_NLL_NOLAG, KNOB_NLL_GRAD_TREE, KNOB_ADAM_STREAM, KNOB_ADAM_LAG_RHO_PPM, KNOB_ADAM_LAG_HEAD, KNOB_COUNT
};
bool knob_set(Knob k);               // the variable exists
int knob_int(Knob k, int dflt);      // its integer value, or dflt when unset

// First-call latency: HIP loads a translation unit's code object the first time one of its kernels is used (1-3 MB
// each here).  Every unit defines an empty kernel and a `touch_<unit>()` that asks for its attributes - which loads
// the unit's code object and nothing else - so that eks_warmup can load what a caller is going to need ahead of the
// first real call (eks_api.hip).
#define EKS_DEFINE_TOUCH(unit)                                                                              \
  namespace eks {                                                                                           \
  __global__ void touch_##unit##_kernel() {}                                                                \
  void touch_##unit() {                                                                                     \
    hipFuncAttributes a;                                                                                    \
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(touch_##unit##_kernel));                   \
  }                                                                                                         \
  }
void touch_misc();
void touch_diag();
void touch_diag_nll();
void touch_dense();
void touch_dense_wave();
void touch_dense_wide();
void touch_loss();
void touch_loss_ar1();
void touch_multicam();
void touch_lag_adam();

// per-kernel timing scope (eks_profile.hip); a no-op unless eks_profile_enable(1) was called
class ProfScope {
 public:
  ProfScope(const char* name, hipStream_t st);
  ~ProfScope();

 private:
  const char* name_;
  hipStream_t st_;
  bool live_;
  hipEvent_t a_, b_;
};

// scalar-chain path (eks_diag.hip)
size_t diag_smooth_workspace_bytes(int T, int N);
int diag_smooth(const eks_dims_t& d, const float* y, const float* var, const DiagModel& M,
                float* ms, float* Vs, void* ws, size_t ws_bytes, hipStream_t st);
size_t diag_nll_workspace_bytes(int T, int N, int n_cand);
struct AdamFuse;   // eks_adam.hpp
int diag_nll(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M,
             const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
             void* ws, size_t ws_bytes, hipStream_t st, const AdamFuse* fuse = nullptr, double* s_out = nullptr,
             int32_t* idx_out = nullptr);   // s_out: also the argmin over the candidates (eks_nll_argmin)
bool diag_nll_grad_tree(int T, int K, int D);
bool diag_nll_adam_persist_ok(int T, int K, int D, int n_blocks);
int diag_nll_adam_persist(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M, int n_iters,
                          double* nll, double* dnll, const AdamFuse& F, int32_t* n_active, hipStream_t st);
// the search from cached lag sums (eks_lag_adam.hip): one pre-pass + one launch per eks_adam_run call
bool diag_lag_adam_ok(int T, int K, int D, int n_blocks);
size_t diag_lag_adam_workspace_bytes(int T, int N);
int diag_lag_sums(const eks_dims_t& d, const float* y, const double* A, const AdamFuse* F, void* ws, size_t ws_bytes,
                  hipStream_t st);
int diag_lag_adam(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M, int n_iters, double* nll,
                  double* dnll, const AdamFuse& F, void* ws, size_t ws_bytes, hipStream_t st);
size_t adam_extra_bytes(int N);     // tail of the NLL workspace: keypoint -> block map, tile tickets, counter
int32_t* nll_ws_tickets(void* ws, int T, int N, int n_cand);   // the tile tickets inside that tail
int adam_prepare(int n_blocks, int K, const int32_t* offs, const int32_t* members, int32_t* kp_block,
                 int32_t* counter_a, int32_t* counter_b, hipStream_t st);

// general small-matrix path (eks_dense.hip)
struct DenseModel {
  const double *m0, *S0, *A, *C, *Q, *s;
};
size_t dense_smooth_workspace_bytes(int T, int K, int D, int O);
int dense_smooth(const eks_dims_t& d, const float* y, const float* var, const DenseModel& M,
                 float* ms, float* Vs, void* ws, size_t ws_bytes, hipStream_t st);
// narrow sessions: wave-per-64-chunks form (eks_dense_wave.hip)
bool dense_wave_covers(int T, int K, int D, int O);
size_t dense_wave_workspace_bytes(int T, int K, int D);
bool dense_wave_ar1_covers(int T, int K, int D, int O);
int dense_wave_ar1_score_step(const eks_dims_t& d, const float* y, const float* var, const double* m0, const double* S0,
                              const double* C, const double* latent_var, double lr, double tol, int cap, double* state,
                              double* a, double* q, double* da, double* dq, double* nll, double* dnll,
                              int32_t* n_active, void* ws, size_t ws_bytes, hipStream_t st);
int dense_wave_ar1_score(const eks_dims_t& d, const float* y, const float* var, const double* m0, const double* S0,
                         const double* C, const double* a, const double* q, const double* da, const double* dq,
                         int n_tan, double* nll, double* dnll, void* ws, size_t ws_bytes, hipStream_t st);
int dense_wave_score(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& M, double* nll,
                     double* dnll, void* ws, size_t ws_bytes, hipStream_t st);
int dense_wave_smooth(const eks_dims_t& d, const float* y, const float* var, const DenseModel& M, float* ms,
                      float* Vs, void* ws, size_t ws_bytes, hipStream_t st);
// wide sessions: prefetching summarize / per-lane sequential scan / checkpointed replay (eks_dense_wide.hip)
struct DenseModelPtrs;
bool dense_wide_covers(int D, int O, int B);
int dense_wide_summarize(int T, int K, int D, int O, int B, int nc, const DenseModelPtrs& M, const double* s,
                         const float* y, const float* var, const double* rconst, double* elems, int soa,
                         double* first, hipStream_t st);   // var == nullptr: SCORE form (constant R)
int dense_score_finish(int K, int rows, const double* part_ll, const double* part_score, double* nll, double* dnll,
                       hipStream_t st);
bool dense_score_covers(int T, int K, int D, int O);
size_t dense_score_workspace_bytes(int T, int K, int D, int O);
int dense_score(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& M, double* nll,
                double* dnll, void* ws, size_t ws_bytes, hipStream_t st);
size_t dense_wide_scan_scratch_doubles(int K, int D, int nc);
int dense_wide_scan(int K, int D, int nc, const double* elems, const double* first, double* scratch,
                    double* chunk_in, double* chunk_out, hipStream_t st);
int dense_wide_replay(int T, int K, int D, int O, int B, int nc, const DenseModelPtrs& M, const double* s,
                      const float* y, const float* var, const double* rconst, double* part_ll,
                      double* part_score, const double* pre, const double* suf, const double* bprior,
                      const double* bsuffix, const double* chunk_in, const double* chunk_out, float* ms,
                      float* Vs, int vs_diag, hipStream_t st);
size_t dense_nll_workspace_bytes(int T, int K, int D, int O, int n_cand);
int dense_nll(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& M,
              const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
              void* ws, size_t ws_bytes, hipStream_t st);

size_t ekf_smooth_workspace_bytes(int T, int K, int smooth);
int ekf_smooth(const eks_dims_t& d, int n_data_keypoints, const float* y, const float* var,
               const double* rconst, const DenseModel& M, const double* cams, int n_cams,
               double* xlin, int max_sweeps, double tol, float* ms, float* Vs, double* nll,
               double* info, void* ws, size_t ws_bytes, hipStream_t st);

size_t ar1_nll_workspace_bytes(int T, int K, int D, int n_tan);
int ar1_nll(const eks_dims_t& d, const float* y, const float* var, const double* m0,
            const double* S0, const double* C, const double* a, const double* q, const double* da,
            const double* dq, int n_tan, double* nll, double* dnll, void* ws, size_t ws_bytes,
            hipStream_t st);

// misc (eks_misc.hip)
size_t const_r_workspace_bytes(int T, int N);
int const_r(int T, int N, const float* var, double min_var, double* rconst, void* ws,
            size_t ws_bytes, hipStream_t st);
int np_nanstd_rows(int K, int n, const float* d, const int32_t* leaves, int n_leaves, const int32_t* ops, int n_ops,
                   float* out, hipStream_t st);
int np_nanstd_diff_rows(int n_frames, int K, int O, const float* x, const int32_t* leaves, int n_leaves, const int32_t* ops,
                        int n_ops, float* out, hipStream_t st);
int order_stats(int T, int N, const float* x, int r_lo, int r_hi, float* out, int32_t* nan_count,
                hipStream_t st);
int argmin_s(int K, int n_cand, const double* nll, const double* s_cand, double* s_out,
             int32_t* idx_out, hipStream_t st);
int adam_step(int n_blocks, const int32_t* offs, const int32_t* members, const double* nll,
              const double* dnll, double lr, double lo, double hi, double tol, int cap,
              double* state, double* s_keypoint, int32_t* n_active, hipStream_t st);
int pupil_adam_step(int n, const double* latent_var, const double* nll, const double* dnll, double lr,
                    double tol, int cap, double* state, double* a, double* q, double* da, double* dq,
                    int32_t* n_active, hipStream_t st);
int ensemble_stats(int M, int V, int T, int K, const float* markers, int avg_mode, int var_mode,
                   float nan_replacement, float* stats, hipStream_t st);

// multicam helpers (eks_multicam.hip)
int maha_inflate(int K, int N, int C, int L, const double* x, float* v, const double* W, const double* mu,
                 const int32_t* active, double epsilon, double threshold, double scalar, double* maha,
                 int32_t* n_inflated, hipStream_t st);
int multicam_tables(int V, int T, int K, int D, const float* stats, const float* ev, const float* ms,
                    const float* Vs, const double* Cm, const double* mean, double* tables, double* latent,
                    hipStream_t st);

// dispatch on the state dimension of the general path (D = 1 .. 6 are instantiated)
#define EKS_DISPATCH_D(D_, BODY) \
  switch (D_) {                  \
    case 1: { constexpr int DD = 1; BODY; } break; \
    case 2: { constexpr int DD = 2; BODY; } break; \
    case 3: { constexpr int DD = 3; BODY; } break; \
    case 4: { constexpr int DD = 4; BODY; } break; \
    case 5: { constexpr int DD = 5; BODY; } break; \
    case 6: { constexpr int DD = 6; BODY; } break; \
    default: return EKS_ERR_UNSUPPORTED;           \
  }

}  // namespace eks
