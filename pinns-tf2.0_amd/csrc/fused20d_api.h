// fused20d_api.h -- host-side interface of the float64 register-stash kernel (kernels_fused20d.h).  The kernel lives in
// its own translation unit (fused20d_unit.hip) because it is compiled with -mllvm -amdgpu-mfma-vgpr-form=1: with 240
// AGPRs taken by the stash, hipcc's default (matrix results in AGPRs) costs two v_accvgpr_read per result, 1 900 of
// them per tile; the flag keeps the results in VGPRs.  The other kernels keep the default allocation they were tuned
// with.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_generic.h"

namespace pinn {

// 16-value gradient blocks of one wave: dense 0 (5), hidden layers 1..H-1 (30 each), dense H (6)
constexpr int fused20d_blocks(int H) { return 5 + (H - 1) * 30 + 6; }
// the flat weight vector is brought into LDS by LDS-DMA in whole 1-KiB pieces (128 doubles)
inline size_t fused20d_weight_doubles(int n_theta) { return ((size_t)n_theta + 127) / 128 * 128; }
// one-tile launches (workgroups >= tiles) keep no accumulators: every wave parks the UNFOLDED partial blocks of one phase of
// the reverse sweep (<= 30 blocks x 64 lanes) in a staging area, double buffered, and the workgroup sums them phase by phase
constexpr int FUSED20D_STAGE_WAVE = 30 * 64;                    // doubles per wave and buffer
constexpr int FUSED20D_STAGE_BUF = 4 * FUSED20D_STAGE_WAVE;     // doubles per buffer (four waves)
inline size_t fused20d_lds_bytes(int n_hidden, int n_theta) {
  const size_t acc = (size_t)4 * fused20d_blocks(n_hidden) * 16, stage = (size_t)2 * FUSED20D_STAGE_BUF;
  return (fused20d_weight_doubles(n_theta) + (acc > stage ? acc : stage) + 4 * 256) * sizeof(double);   // + loss-part slots
}

// entry e = 16 * block + 4 * i + j of a wave's block list -> flat parameter index (reference layout), -1 = padding;
// out holds fused20d_blocks(H) x 16 ints
void fused20d_row_index(const NetDesc& nd, int H, int* out);

inline bool fused20d_depth_ok(int n_hidden) { return n_hidden == 4 || n_hidden == 6 || n_hidden == 8; }
// Launch plan of path 7: 64-point tiles, persistent over n_wg = min(tiles, CUs) workgroups = partial gradient rows.
// one loss+gradient evaluation (pde 0: Burgers inference, 1: identification; 4, 6 or 8 hidden layers); returns a hipError_t
int fused20d_launch_any(int pde, const NetDesc& nd, const SetDesc& sd, const double* th, const double* xs,
                        const double* ts, const double* tgt, double lbx, double lbt, double sx, double st, double nu,
                        double* part, int R, int n_wg, const int* row_index, hipStream_t stream,
                        long long* stamps, hipEvent_t ev_start, hipEvent_t ev_stop);

}  // namespace pinn
