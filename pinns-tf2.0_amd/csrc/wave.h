// wave.h -- wave64 cross-lane primitives for gfx950 (CDNA4).
//
// A wavefront is 64 lanes = 4 DPP rows of 16.  Reductions go through DPP row operations
// (no LDS traffic): two quad permutes, row_half_mirror, row_mirror reduce each row of 16;
// row_bcast15 / row_bcast31 carry the row sums across rows so that lane 63 holds the total.
#pragma once
#include <hip/hip_runtime.h>

namespace pinn {

// Host side: kernel attributes (dynamic LDS size) are per device, so "set once" flags are per device too.
// True exactly once per (flag, current device).
inline bool first_call_on_device(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

template <typename T> struct alignas(4 * sizeof(T)) vec4 { T x, y, z, w; };

// DPP control words (gfx9 encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
  return __int_as_float(r);
}

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Sum over the 16 lanes of a DPP row (four vector-ALU steps, no LDS traffic); every lane of the row gets it.
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
  v += dpp_mov<DPP_QUAD_XOR1>(v);
  v += dpp_mov<DPP_QUAD_XOR2>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}

// Sum over the 64 lanes of a wave; the result is wave-uniform (read back from lane 63).
// Fixed combination order -> bit-reproducible run to run.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  v += dpp_mov<DPP_QUAD_XOR1>(v);
  v += dpp_mov<DPP_QUAD_XOR2>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  // bound_ctrl=true writes 0 into lanes without a source: rows 1,3 receive row 0/2 totals
  v += dpp_mov<DPP_ROW_BCAST15, 0xA>(v);
  v += dpp_mov<DPP_ROW_BCAST31, 0xC>(v);
  return read_lane(v, 63);
}

}  // namespace pinn
