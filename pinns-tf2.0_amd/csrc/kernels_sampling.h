// kernels_sampling.h -- Latin hypercube collocation points generated on the device, in place.
//
// The reference draws its collocation set once on the host with pyDOE's classic LHS
// (1d-burgers/burgersutil.py:122, 1dcomplex-schrodinger/schrodingerutil.py:58): per dimension, stratum k of n
// holds one uniform sample and the strata are handed to the points through a random permutation.  That
// construction needs a shuffle (sequential on the host, a sort on a GPU).  Here both ingredients are
// counter-based, so point i of an n-point design is a pure function of (seed, dimension, i):
//     x_{i,d} = lb_d + (ub_d - lb_d) * (pi_d(i) + u_{i,d}) / n
//   pi_d  = keyed bijection of [0, n): a 6-round Feistel network on 2*h bits (2^(2h) >= n) with cycle walking
//   u     = Philox4x32-10 (Salmon et al., SC'11) -> 53-bit uniform in [0, 1)
// No sort, no global state, any shard [first, first + count) of the same design can be produced by any rank
// independently (data-parallel ranks build disjoint shards of ONE hypercube), and re-drawing the whole set with
// a new seed is one ~5 us launch -- which is what makes per-epoch resampling affordable.
// It is the same distribution as pyDOE's, not the same stream: numpy's MT19937 shuffle cannot be reproduced
// in parallel.  oracle/lhs.py restates the integer pipeline in numpy; tests compare bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pinn {

__host__ __device__ inline void philox4x32_10(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0], p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1, n3 = (uint32_t)p0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// 32-bit finaliser of MurmurHash3: the Feistel round function
__host__ __device__ inline uint32_t lhs_mix(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// keyed permutation of [0, n), n <= 2^62
__host__ __device__ inline uint64_t lhs_permute(uint64_t i, uint64_t n, int half_bits, uint32_t k0, uint32_t k1) {
  const uint64_t mask = (1ull << half_bits) - 1;
  uint64_t v = i;
  do {
    uint32_t L = (uint32_t)(v >> half_bits), Rr = (uint32_t)(v & mask);
    for (int r = 0; r < 6; ++r) {
      const uint32_t f = lhs_mix(Rr ^ (r & 1 ? k1 : k0) ^ (0x9E3779B9u * (uint32_t)(r + 1))) & (uint32_t)mask;
      const uint32_t t = L ^ f;
      L = Rr; Rr = t;
    }
    v = ((uint64_t)L << half_bits) | Rr;
  } while (v >= n);
  return v;
}

__host__ __device__ inline int lhs_half_bits(uint64_t n) {
  int bits = 1;
  while (bits < 62 && (1ull << bits) < n) ++bits;
  return (bits + 1) / 2;
}

// writes points [first, first + count) of the n-point, 2-dimensional design into xs/ts (compute dtype)
template <typename real>
__global__ __launch_bounds__(256) void k_lhs_fill(real* __restrict__ xs, real* __restrict__ ts, int64_t count,
                                                  uint64_t first, uint64_t n, int half_bits, uint32_t seed_lo,
                                                  uint32_t seed_hi, double lbx, double lbt, double rx, double rt) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= count) return;
  const uint64_t i = first + (uint64_t)t;
  uint32_t ctr[4] = {(uint32_t)i, (uint32_t)(i >> 32), 0u, 0x4C485321u};
  philox4x32_10(ctr, seed_lo, seed_hi);
  const double inv = 1.0 / 9007199254740992.0;             // 2^-53
  const double u0 = (double)((((uint64_t)ctr[0] << 32) | ctr[1]) >> 11) * inv;
  const double u1 = (double)((((uint64_t)ctr[2] << 32) | ctr[3]) >> 11) * inv;
  const uint64_t p0 = lhs_permute(i, n, half_bits, seed_lo ^ 0x243F6A88u, seed_hi ^ 0x85A308D3u);
  const uint64_t p1 = lhs_permute(i, n, half_bits, seed_lo ^ 0x13198A2Eu, seed_hi ^ 0x03707344u);
  const double dn = (double)n;
  // explicit roundings: no fused multiply-add, so that the numpy restatement is bit-identical
  const double x = __dadd_rn(lbx, __dmul_rn(rx, __ddiv_rn(__dadd_rn((double)p0, u0), dn)));
  const double tt = __dadd_rn(lbt, __dmul_rn(rt, __ddiv_rn(__dadd_rn((double)p1, u1), dn)));
  xs[t] = (real)x;
  ts[t] = (real)tt;
}

}  // namespace pinn
