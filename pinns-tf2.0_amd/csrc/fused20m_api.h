// fused20m_api.h -- host-side interface of the float32 register-stash kernel (kernels_fused20m.h) at the depths other
// than the benchmark's 8.  hp["layers"] is free-form in the reference (1d-burgers/inf_cont_burgers.py:23-43): a 4x20,
// 6x20 or 10x20 net used to drop to the HBM-stash kernel (11.5 / 15.2 TFLOP/s against 19.8 for 8x20,
// profiles/r02_time_shapes.txt).  The instantiations live in their own translation unit (fused20m_unit.hip) so that
// the three units of the library compile concurrently; H = 8 stays in engine.hip (the ablation builds relink it).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_generic.h"

namespace pinn {

inline bool fused20m_depth_ok(int n_hidden) { return n_hidden == 4 || n_hidden == 6 || n_hidden == 8 || n_hidden == 10; }

// one loss+gradient evaluation, n_hidden in {4, 6, 10} (pde 0: Burgers inference, 1: identification); returns a hipError_t
int fused20m_launch_depth(int pde, const NetDesc& nd, const SetDesc& sd, const float* th, const float* img,
                          const float* xs, const float* ts, const float* tgt, float lbx, float lbt, float sx, float st,
                          float nu, float* part, int R, int n_wg, hipStream_t stream, long long* stamps,
                          hipEvent_t ev_start, hipEvent_t ev_stop);

// Launch plan of path 2 for n_pad points on n_cu compute units: k_fused20m (one workgroup per CU, full register stash);
// with PINN_F32_RECOMPUTE=1 in the environment, 8 hidden layers and more tiles than CUs: the experimental k_fused20r
// (kernels_fused20r.h: even layers stashed, odd layers recomputed, two workgroups per CU -- built, parity-green,
// measured slower because of register spills; kept for the record and the test that pins it).
// n_wg = workgroups = partial gradient rows.
struct Fused20mPlan { int n_wg; int recompute; };
Fused20mPlan fused20m_plan(int n_hidden, int n_pad, int n_cu);
int fused20r_launch_any(int pde, const NetDesc& nd, const SetDesc& sd, const float* th, const float* img,
                        const float* xs, const float* ts, const float* tgt, float lbx, float lbt, float sx, float st,
                        float nu, float* part, int R, int n_wg, hipStream_t stream, long long* stamps,
                        hipEvent_t ev_start, hipEvent_t ev_stop);

}  // namespace pinn
