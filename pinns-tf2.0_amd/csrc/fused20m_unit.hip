// fused20m_unit.hip -- k_fused20m at hidden depths 4, 6 and 10 (see fused20m_api.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -c fused20m_unit.hip
#include "kernels_fused20m.h"
#include "fused20m_api.h"

namespace pinn {

template <int H>
static int launch_h(int pde, const NetDesc& nd, const SetDesc& sd, const float* th, const float* img, const float* xs,
                    const float* ts, const float* tgt, float lbx, float lbt, float sx, float st, float nu, float* part,
                    int R, int n_wg, hipStream_t stream, long long* stamps, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (pde == 1)
    return fused20m_launch<1, H>(nd, sd, th, img, xs, ts, tgt, lbx, lbt, sx, st, nu, part, R, n_wg, stream, stamps,
                                 ev_start, ev_stop);
  return fused20m_launch<0, H>(nd, sd, th, img, xs, ts, tgt, lbx, lbt, sx, st, nu, part, R, n_wg, stream, stamps,
                               ev_start, ev_stop);
}

int fused20m_launch_depth(int pde, const NetDesc& nd, const SetDesc& sd, const float* th, const float* img,
                          const float* xs, const float* ts, const float* tgt, float lbx, float lbt, float sx, float st,
                          float nu, float* part, int R, int n_wg, hipStream_t stream, long long* stamps,
                          hipEvent_t ev_start, hipEvent_t ev_stop) {
#define ARGS pde, nd, sd, th, img, xs, ts, tgt, lbx, lbt, sx, st, nu, part, R, n_wg, stream, stamps, ev_start, ev_stop
  switch (nd.n_hidden) {
    case 4: return launch_h<4>(ARGS);
    case 6: return launch_h<6>(ARGS);
    case 10: return launch_h<10>(ARGS);
    default: return (int)hipErrorInvalidValue;
  }
#undef ARGS
}

}  // namespace pinn
