// kernels_fused20.h -- fused width-20 Burgers loss+gradient kernel (placeholder until the
// LDS/MFMA kernel lands; reports "unsupported" so the generic kernels serve every shape).
#pragma once
#include "kernels_generic.h"

namespace pinn {

inline bool fused20_supported(const NetDesc&) { return false; }
inline int fused20_rows(const SetDesc&) { return 0; }

template <typename real, int PDE>
inline int fused20_launch(const NetDesc&, const SetDesc&, const real*, const real*, const real*,
                          const real*, real, real, real, real, real, real*, int, hipStream_t) {
  return (int)hipErrorNotSupported;
}

}  // namespace pinn
