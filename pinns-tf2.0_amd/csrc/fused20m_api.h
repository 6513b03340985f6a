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

}  // namespace pinn
