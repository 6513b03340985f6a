// fused20d_unit.hip -- translation unit of k_fused20d (see fused20d_api.h for why it is separate).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -c fused20d_unit.hip
#include "kernels_fused20d.h"

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
}

int fused20d_launch_any(int pde, const NetDesc& nd, const SetDesc& sd, const double* th, const double* xs,
                        const double* ts, const double* tgt, double lbx, double lbt, double sx, double st, double nu,
                        double* part, int R, int n_wg, const int* row_index, hipStream_t stream,
                        long long* stamps, hipEvent_t ev_start, hipEvent_t ev_stop) {
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
