// fused20d_unit.hip -- translation unit of k_fused20d (see fused20d_api.h for why it is separate).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -c fused20d_unit.hip
#include "kernels_fused20d.h"
#include "kernels_fused20dh.h"

#include <cstdlib>

namespace pinn {

// entry e = 16 * block + 4 * i + j of a wave's block list -> flat parameter index (reference layout), -1 = padding
void fused20d_row_index(const NetDesc& nd, int H, int* out) {
  const int NBLK = fused20d_blocks(H), BLK_H = 5 + (H - 1) * 30;
  for (int e = 0; e < NBLK * 16; ++e) {
    const int blk = e >> 4, i = (e >> 2) & 3, j = e & 3;
    int idx = -1;
    if (blk < 5) {                                   // dense 0: rows (w0x, w0t, b0)
      const int f = 4 * blk + j;
      idx = i == 0 ? nd.off_w[0] + f : i == 1 ? nd.off_w[0] + FW + f : i == 2 ? nd.off_b[0] + f : -1;
    } else if (blk < BLK_H) {                        // hidden layer d: 25 weight blocks (m, n), 5 bias blocks
      const int r = blk - 5, d = 1 + r / 30, mn = r - (d - 1) * 30;
      if (mn < 25) { const int m = mn / 5, n = mn - 5 * m; idx = nd.off_w[d] + (4 * m + i) * FW + 4 * n + j; }
      else if (i == 0) idx = nd.off_b[d] + 4 * (mn - 25) + j;
    } else if (j == 0) {                             // dense H: one output column
      const int m = blk - BLK_H;
      idx = m < 5 ? nd.off_w[H] + 4 * m + i : (i == 0 ? nd.off_b[H] : -1);
    }
    out[e] = idx;
  }
  fused20dh_slot_table(H, out + NBLK * 16);
}

Fused20dPlan fused20d_plan(int n_pad, int n_cu) {
  // k_fused20dh is OPT-IN (PINN_F64_HELPER=1): parity-green, and measured exactly as fast as k_fused20d at the metric's
  // N_f = 10 000 (41.90 vs 41.9 us per Adam step, profiles/r03_helper_wave.txt) -- see kernels_fused20dh.h for why
  static const bool on = [] { const char* e = getenv("PINN_F64_HELPER"); return e && e[0] == '1'; }();
  const int t48 = fused20dh_tiles(n_pad), t64 = n_pad / 64;
  if (on && t48 <= n_cu) return Fused20dPlan{t48, 1};
  return Fused20dPlan{t64 < n_cu ? t64 : n_cu, 0};
}

size_t fused20d_plan_lds_bytes(Fused20dPlan plan, int n_hidden, int n_theta) {
  return plan.helper ? fused20dh_lds_bytes(n_hidden, n_theta) : fused20d_lds_bytes(n_hidden, n_theta);
}

int fused20d_launch_any(int pde, const NetDesc& nd, const SetDesc& sd, const double* th, const double* xs,
                        const double* ts, const double* tgt, double lbx, double lbt, double sx, double st, double nu,
                        double* part, int R, Fused20dPlan plan, const int* row_index, hipStream_t stream,
                        long long* stamps, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (plan.helper) {
#define ARGS nd, sd, th, xs, ts, tgt, lbx, lbt, sx, st, nu, part, R, row_index, stream, stamps, ev_start, ev_stop
    switch (nd.n_hidden) {
      case 4: return pde == 1 ? fused20dh_launch<1, 4>(ARGS) : fused20dh_launch<0, 4>(ARGS);
      case 6: return pde == 1 ? fused20dh_launch<1, 6>(ARGS) : fused20dh_launch<0, 6>(ARGS);
      case 8: return pde == 1 ? fused20dh_launch<1, 8>(ARGS) : fused20dh_launch<0, 8>(ARGS);
      default: return (int)hipErrorInvalidValue;
    }
#undef ARGS
  }
  const int n_wg = plan.n_wg;
#define ARGS nd, sd, th, xs, ts, tgt, lbx, lbt, sx, st, nu, part, R, n_wg, row_index, stream, stamps, ev_start, ev_stop
  switch (nd.n_hidden) {     // the AGPR stash holds (H - 2) x 40 registers: depths up to 8 fit the 256 of a wave
    case 4: return pde == 1 ? fused20d_launch<1, 4>(ARGS) : fused20d_launch<0, 4>(ARGS);
    case 6: return pde == 1 ? fused20d_launch<1, 6>(ARGS) : fused20d_launch<0, 6>(ARGS);
    case 8: return pde == 1 ? fused20d_launch<1, 8>(ARGS) : fused20d_launch<0, 8>(ARGS);
    default: return (int)hipErrorInvalidValue;
  }
#undef ARGS
}

}  // namespace pinn
