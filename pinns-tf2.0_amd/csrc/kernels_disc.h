// kernels_disc.h -- discrete-time (implicit Runge-Kutta) Burgers models: 1-input tanh MLP with q(+1) outputs,
// loss = sum over stage sets of || U + N(U) M^T - target ||^2 with N = c1 U U_x - c2 U_xx on the first q outputs
// (reference: 1d-burgers/inf_disc_burgers.py:57-95, 1d-burgers/ide_disc_burgers.py:81-115).
//
// Unlike the continuous models this path is GEMM shaped (hidden 50, 501 outputs, a 500 x 501 IRK table) over a
// few hundred points, so every contraction runs on v_mfma_{f32,f64}_16x16x4 and the evaluation is four launches
// whose grids are (16-point groups) x (64-column chunks of the output dimension):
//
//   k_disc_fwd         hidden layers (3 Taylor channels value / d/dx / d2/dx2, recomputed per chunk, LDS resident),
//                      output-layer chunk, N chunk                              -> U3, Nn (+ stash for the reverse)
//   k_disc_irk         R = U + Nn M^T - target (K = q), sum R^2                 -> R, loss partials
//   k_disc_bwd_out     dN = 2 R M (K = n_out), output-channel adjoints G_c, dW_out / db_out partial rows,
//                      partial adjoint of the last hidden layer (K = this chunk's 64 columns)
//   k_disc_bwd_hidden  sums the chunk partials, reverse sweep through the hidden layers, dW_l partial rows
//
// then the ordinary k_reduce_rows over the per-group gradient rows.  The problem is a few hundred points, so the
// kernels are latency bound: weights are staged into LDS through registers one layer ahead (the global loads fly
// under the previous layer's MFMAs), the K ~ 500 contractions keep two 64-deep operand stages in registers, and
// LDS strides are chosen per access pattern (== 4 mod 32 words for [row = lane&15][col = lane>>4] reads, == 16
// mod 32 for [row = lane>>4][col = lane&15] reads) so that a wave's 64 addresses fall 2 per bank.
//
// Operand convention of the 16x16x4 MFMA as used below: lane = (m | n) + 16 g;  A operand = A[m][4 s + g],
// B operand = B[4 s + g][n], accumulator register r = D[out_row(lane, r)][n]  (out_row differs between the f32
// and the f64 instruction).  NT = hidden width rounded up to 16, in tiles (4: widths <= 64, 8: widths <= 128).
#pragma once
#include "kernels_fused20.h"

namespace pinn {

struct DiscDesc {
  int n_groups;      // 16-point groups over all stage sets (a group never straddles two sets)
  int n_pad;         // 16 * n_groups
  int ldo;           // row stride of every [point][column] array: n_out rounded up to 64
  int n_chunks;      // ldo / 64
  int q;             // outputs entering N  (q <= n_out)
  int wp;            // hidden width rounded up to 64 (= 16 NT)
  int identify;      // 1: c1 = theta[n_net], c2 = exp(theta[n_net + 1])
};

// group descriptor: stage set in bits 0..7, valid points (1..16) above
__host__ __device__ inline int disc_ginfo(int set, int n_valid) { return set | (n_valid << 8); }

template <typename real> __device__ __forceinline__ real exp_r(real z);
template <> __device__ __forceinline__ float exp_r<float>(float z) { return expf(z); }
template <> __device__ __forceinline__ double exp_r<double>(double z) { return exp(z); }

template <typename real>
__device__ __forceinline__ void disc_coefs(const NetDesc& nd, const DiscDesc& dd, const real* __restrict__ th,
                                           real c1_in, real c2_in, real& c1, real& c2) {
  c1 = c1_in; c2 = c2_in;
  if (dd.identify) { c1 = th[nd.n_net]; c2 = exp_r(th[nd.n_net + 1]); }
}

// LDS geometry shared by host and device
template <int NT> struct DiscGeo {
  static constexpr int WP = 16 * NT;
  static constexpr int LD = WP + 4;            // activations [point][feature]: read as [m][4s+g]
  static constexpr int CS = 16 * LD;           // one channel
  static constexpr int WLD = WP + 16;          // weights [k][j] read as [4s+g][n]   (WP is a multiple of 64)
  static constexpr int TLD = WP + 4;           // weights [k][j] read as [m (k)][4s+g (j)]
  static constexpr int NPRE = WP * WP / 256;   // register-staged weight values per thread
  static constexpr int NPRE_O = WP * 64 / 256; // ... for a [WP][64] output-layer chunk
};
template <int NT> inline size_t disc_fwd_lds(size_t rs) {
  return (size_t)(2 * 3 * DiscGeo<NT>::CS + DiscGeo<NT>::WP * DiscGeo<NT>::WLD) * rs;
}
template <int NT> inline size_t disc_out_lds(size_t rs) {
  return (size_t)(3 * 16 * 68 + 3 * DiscGeo<NT>::CS + DiscGeo<NT>::WP * 68 + 8) * rs;
}
template <int NT> inline size_t disc_hid_lds(size_t rs) {
  return (size_t)(3 * 3 * DiscGeo<NT>::CS + DiscGeo<NT>::WP * DiscGeo<NT>::TLD + 16) * rs;
}

// register staging of a [rows <= WP][cols] weight tile: thread t holds elements t, t + 256, ... of the padded
// [WP][COLS] tile (zero outside the valid rows / columns)
template <typename real, int NT, int COLS>
__device__ __forceinline__ void wt_load(real (&pre)[16 * NT * COLS / 256], const real* __restrict__ src, int ld,
                                        int rows, int cols, int tid) {
#pragma unroll
  for (int n = 0; n < 16 * NT * COLS / 256; ++n) {
    const int i = tid + 256 * n, k = i / COLS, j = i % COLS;
    pre[n] = (k < rows && j < cols) ? src[(size_t)k * ld + j] : real(0);
  }
}
template <typename real, int NT, int COLS>
__device__ __forceinline__ void wt_store(const real (&pre)[16 * NT * COLS / 256], real* __restrict__ dst, int stride,
                                         int tid) {
#pragma unroll
  for (int n = 0; n < 16 * NT * COLS / 256; ++n) {
    const int i = tid + 256 * n;
    dst[(i / COLS) * stride + (i % COLS)] = pre[n];
  }
}

// acc += A[16 x K] B[K x 16] for this wave's tile, both operands in global memory (L2 resident):
//   arow = &A[m][4 g] (row-major, 16-byte aligned), bcol = &B[4 g][n] (row stride ldb); K = 16 nblocks, nblocks a
//   multiple of 4 (the arrays are zero padded).  Two register stages of 4 blocks each: the loads of the next 64
//   k-values are in flight while the 16 MFMAs of the current stage issue.
template <typename real>
struct DiscStage {
  vec4<real> a[4];
  real b[4][4];
  __device__ __forceinline__ void load(const real* __restrict__ arow, const real* __restrict__ bcol, size_t ldb,
                                       int S) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const vec4<real>*>(arow + 16 * (S + u));
      const real* bp = bcol + (size_t)16 * (S + u) * ldb;
      b[u][0] = bp[0]; b[u][1] = bp[ldb]; b[u][2] = bp[2 * ldb]; b[u][3] = bp[3 * ldb];
    }
  }
  template <typename acc_t> __device__ __forceinline__ acc_t mma(acc_t acc) const {
    using TR = FusedTraits<real>;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = TR::mfma(a[u].x, b[u][0], acc);
      acc = TR::mfma(a[u].y, b[u][1], acc);
      acc = TR::mfma(a[u].z, b[u][2], acc);
      acc = TR::mfma(a[u].w, b[u][3], acc);
    }
    return acc;
  }
};

template <typename real>
__device__ __forceinline__ typename FusedTraits<real>::acc_t disc_gemm_k(const real* __restrict__ arow,
                                                                         const real* __restrict__ bcol,
                                                                         size_t ldb, int nblocks) {
  typename FusedTraits<real>::acc_t acc = {0, 0, 0, 0};
  DiscStage<real> s0, s1;
  s0.load(arow, bcol, ldb, 0);
  for (int S = 0; S < nblocks; S += 8) {
    if (S + 4 < nblocks) s1.load(arow, bcol, ldb, S + 4);
    acc = s0.mma(acc);
    if (S + 8 < nblocks) s0.load(arow, bcol, ldb, S + 8);
    if (S + 4 < nblocks) acc = s1.mma(acc);
  }
  return acc;
}
__host__ __device__ inline int disc_kblocks(int K) { return ((K + 15) / 16 + 3) / 4 * 4; }

// ---------------------------------------------------------------------------------------------------------
// forward
//   Ast [H][3][n_pad][wp]   per hidden layer (a, z_x, z_xx): tanh output and the derivative channels of the
//                           pre-activation -- what the reverse sweep needs (written by chunk 0 only)
//   A3  [3][n_pad][wp]      output channels (a, a_x, a_xx) of the last hidden layer
//   U3  [3][n_pad][ldo]     (U, U_x, U_xx) of the output layer;  Nn [n_pad][ldo] = c1 U U_x - c2 U_xx (0 beyond q)
// ---------------------------------------------------------------------------------------------------------
template <typename real, int NT>
__global__ __launch_bounds__(256) void k_disc_fwd(NetDesc nd, DiscDesc dd, const real* __restrict__ th,
                                                  const real* __restrict__ xs, real lbx, real sx, real c1_in,
                                                  real c2_in, real* __restrict__ Ast, real* __restrict__ A3,
                                                  real* __restrict__ U3, real* __restrict__ Nn, int write_stash) {
  using TR = FusedTraits<real>;
  using acc_t = typename TR::acc_t;
  using GEO = DiscGeo<NT>;
  constexpr int WP = GEO::WP, LD = GEO::LD, CS = GEO::CS, WLD = GEO::WLD;
  extern __shared__ __attribute__((aligned(16))) char disc_smem[];
  real* act = reinterpret_cast<real*>(disc_smem);          // [2][3][16][LD]
  real* wb = act + 2 * 3 * CS;                             // [WP][WLD] weights of the layer being applied
  const int G = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = nd.width, H = nd.n_hidden, NO = nd.n_out;
  const int n_pad = dd.n_pad, ldo = dd.ldo;
  const int m = lane & 15, g = lane >> 4;
  const bool stash = write_stash && ch == 0;
  real c1, c2;
  disc_coefs(nd, dd, th, c1_in, c2_in, c1, c2);

  real pre[GEO::NPRE];                                     // next layer's weights, in flight
  if (H > 1) wt_load<real, NT, WP>(pre, th + nd.off_w[1], W, W, W, tid);
  else wt_load<real, NT, WP>(pre, th + nd.off_w[H] + 64 * ch, NO, W, min(64, NO - 64 * ch), tid);

  // dense 0 (fan_in 1): h = sx (x - lb) - 1, h_x = sx, h_xx = 0
  for (int i = tid; i < 16 * WP; i += 256) {
    const int p = i / WP, j = i % WP;
    real a = 0, zp = 0, ax = 0, axx = 0;
    if (j < W) {
      const real w0 = th[nd.off_w[0] + j], b0 = th[nd.off_b[0] + j];
      const real h = sx * (xs[16 * G + p] - lbx) - real(1);
      zp = w0 * sx;
      a = tanh_r(w0 * h + b0);
      const real d1 = real(1) - a * a, d2 = real(-2) * a * d1;
      ax = d1 * zp;
      axx = d2 * zp * zp;
    }
    act[0 * CS + p * LD + j] = a;
    act[1 * CS + p * LD + j] = ax;
    act[2 * CS + p * LD + j] = axx;
    if (stash) {
      const size_t o = (size_t)(16 * G + p) * WP + j;
      Ast[(size_t)0 * n_pad * WP + o] = a;
      Ast[(size_t)1 * n_pad * WP + o] = zp;
      Ast[(size_t)2 * n_pad * WP + o] = 0;
    }
  }

  int cur = 0;
  const int ksteps = (W + 3) / 4;
  for (int l = 1; l < H; ++l) {
    __syncthreads();                                       // act[cur] complete, previous readers of wb done
    wt_store<real, NT, WP>(pre, wb, WLD, tid);
    __syncthreads();
    if (l + 1 < H) wt_load<real, NT, WP>(pre, th + nd.off_w[l + 1], W, W, W, tid);
    else wt_load<real, NT, WP>(pre, th + nd.off_w[H] + 64 * ch, NO, W, min(64, NO - 64 * ch), tid);
    const real* __restrict__ bl = th + nd.off_b[l];
    const real* in = act + cur * 3 * CS;
    real* out = act + (cur ^ 1) * 3 * CS;
    for (int ct = wave; ct < NT; ct += 4) {
      const int j = 16 * ct + m;
      const real bj = j < W ? bl[j] : real(0);
      acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
      for (int ks = 0; ks < ksteps; ++ks) {
        const int k = 4 * ks + g;
        const real b = wb[k * WLD + j];
        const real* ap = in + m * LD + k;
        acc0 = TR::mfma(ap[0], b, acc0);
        acc1 = TR::mfma(ap[CS], b, acc1);
        acc2 = TR::mfma(ap[2 * CS], b, acc2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = TR::out_row(lane, r);
        const real zp = acc1[r], zr = acc2[r];
        const real a = tanh_r(acc0[r] + bj);
        const real d1 = real(1) - a * a, d2 = real(-2) * a * d1;
        out[0 * CS + p * LD + j] = a;
        out[1 * CS + p * LD + j] = d1 * zp;
        out[2 * CS + p * LD + j] = d2 * zp * zp + d1 * zr;
        if (stash) {
          const size_t o = (size_t)(16 * G + p) * WP + j;
          Ast[((size_t)l * 3 + 0) * n_pad * WP + o] = a;
          Ast[((size_t)l * 3 + 1) * n_pad * WP + o] = zp;
          Ast[((size_t)l * 3 + 2) * n_pad * WP + o] = zr;
        }
      }
    }
    cur ^= 1;
  }
  __syncthreads();
  wt_store<real, NT, WP>(pre, wb, WLD, tid);               // output-layer chunk [W][64] (zero padded to [WP][WP])
  __syncthreads();
  const real* in = act + cur * 3 * CS;
  if (stash)
    for (int i = tid; i < 3 * 16 * WP; i += 256) {
      const int c = i / (16 * WP), rem = i % (16 * WP), p = rem / WP, j = rem % WP;
      A3[((size_t)c * n_pad + 16 * G + p) * WP + j] = in[c * CS + p * LD + j];
    }

  {  // output-layer chunk: columns 64 ch + 16 wave + (0..15)
    const int jl = 16 * wave + m, j = 64 * ch + jl;
    const real bj = j < NO ? th[nd.off_b[H] + j] : real(0);
    acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const int k = 4 * ks + g;
      const real b = wb[k * WLD + jl];
      const real* ap = in + m * LD + k;
      acc0 = TR::mfma(ap[0], b, acc0);
      acc1 = TR::mfma(ap[CS], b, acc1);
      acc2 = TR::mfma(ap[2 * CS], b, acc2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t gp = 16 * G + TR::out_row(lane, r);
      const real U = acc0[r] + bj, Ux = acc1[r], Uxx = acc2[r];
      U3[((size_t)0 * n_pad + gp) * ldo + j] = U;
      U3[((size_t)1 * n_pad + gp) * ldo + j] = Ux;
      U3[((size_t)2 * n_pad + gp) * ldo + j] = Uxx;
      Nn[gp * ldo + j] = j < dd.q ? c1 * U * Ux - c2 * Uxx : real(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// IRK stage:  R[p][j] = U[p][j] + sum_k Nn[p][k] MT[k][j] - target[p]   (MT = (dt-scaled table)^T, [ldo][ldo]
// zero padded; MT == nullptr: no table, R = U - target).  mode 1: prediction only (no target, no mask, no loss).
// ---------------------------------------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(256) void k_disc_irk(DiscDesc dd, int n_out, const int* __restrict__ ginfo,
                                                  const real* __restrict__ MT0, const real* __restrict__ MT1,
                                                  const real* __restrict__ Nn, const real* __restrict__ U,
                                                  const real* __restrict__ tgt, real* __restrict__ R,
                                                  real* __restrict__ lossp, int mode) {
  using TR = FusedTraits<real>;
  using acc_t = typename TR::acc_t;
  __shared__ real wsum[4];
  const int G = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4, ldo = dd.ldo;
  const int gi = ginfo[G], set = gi & 0xff, n_valid = gi >> 8;
  const real* __restrict__ MT = set == 0 ? MT0 : MT1;
  const int j = 64 * ch + 16 * wave + m;
  real uo[4], tg[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {                            // epilogue operands, in flight under the contraction
    const size_t gp = 16 * G + TR::out_row(lane, r);
    uo[r] = U[gp * ldo + j];
    tg[r] = mode == 0 ? tgt[gp] : real(0);
  }
  acc_t acc = {0, 0, 0, 0};
  if (MT)
    acc = disc_gemm_k<real>(Nn + (size_t)(16 * G + m) * ldo + 4 * g, MT + (size_t)(4 * g) * ldo + j, ldo,
                            disc_kblocks(dd.q));
  real l = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = TR::out_row(lane, r);
    real v = acc[r] + uo[r];
    if (mode == 0) {
      v = (p < n_valid && j < n_out) ? v - tg[r] : real(0);
      l += v * v;
    }
    R[(size_t)(16 * G + p) * ldo + j] = v;
  }
  if (mode == 0) {
    l = wave_sum(l);
    if (lane == 0) wsum[wave] = l;
    __syncthreads();
    if (tid == 0) lossp[G * dd.n_chunks + ch] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// reverse, output side.  part = per-group gradient rows [n_groups][R_cols]; dAp [n_chunks][3][n_pad][wp];
// lamp [n_groups * n_chunks][2] partial sums for the identification parameters.
// ---------------------------------------------------------------------------------------------------------
template <typename real, int NT>
__global__ __launch_bounds__(256) void k_disc_bwd_out(NetDesc nd, DiscDesc dd, const real* __restrict__ th,
                                                      const int* __restrict__ ginfo,
                                                      const real* __restrict__ M0, const real* __restrict__ M1,
                                                      const real* __restrict__ R, const real* __restrict__ U3,
                                                      const real* __restrict__ A3, real c1_in, real c2_in,
                                                      real* __restrict__ part, int R_cols,
                                                      real* __restrict__ dAp, real* __restrict__ lamp) {
  using TR = FusedTraits<real>;
  using acc_t = typename TR::acc_t;
  using GEO = DiscGeo<NT>;
  constexpr int WP = GEO::WP, LD = GEO::LD, CS = GEO::CS, GL = 68, OL = 68;
  extern __shared__ __attribute__((aligned(16))) char disc_smem[];
  real* Gs = reinterpret_cast<real*>(disc_smem);           // [3][16][GL] output-channel adjoints of this chunk
  real* a3s = Gs + 3 * 16 * GL;                            // [3][16][LD] last hidden layer's output channels
  real* wos = a3s + 3 * CS;                                // [WP][OL] output-layer weights of this chunk
  real* wsum = wos + WP * OL;                              // [4][2]
  const int G = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = nd.width, H = nd.n_hidden, NO = nd.n_out, ldo = dd.ldo, n_pad = dd.n_pad;
  const int m = lane & 15, g = lane >> 4;
  const int set = ginfo[G] & 0xff;
  const real* __restrict__ M = set == 0 ? M0 : M1;
  real c1, c2;
  disc_coefs(nd, dd, th, c1_in, c2_in, c1, c2);
  const int jl = 16 * wave + m, j = 64 * ch + jl;

  // operands of the later phases: issued now, parked in registers during the contraction
  real pa[3 * 16 * WP / 256], pw[GEO::NPRE_O];
#pragma unroll
  for (int n = 0; n < 3 * 16 * WP / 256; ++n) {
    const int i = tid + 256 * n, c = i / (16 * WP), rem = i % (16 * WP);
    pa[n] = A3[((size_t)c * n_pad + 16 * G + rem / WP) * WP + rem % WP];
  }
  wt_load<real, NT, 64>(pw, th + nd.off_w[H] + 64 * ch, NO, W, min(64, NO - 64 * ch), tid);
  real Rv[4], Uv[4], Uxv[4], Uxxv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t o = (size_t)(16 * G + TR::out_row(lane, r)) * ldo + j;
    Rv[r] = R[o]; Uv[r] = U3[o]; Uxv[r] = U3[(size_t)n_pad * ldo + o];
    Uxxv[r] = dd.identify ? U3[(size_t)2 * n_pad * ldo + o] : real(0);
  }

  {  // dN tile = 2 R M  (K = n_out), then the three output-channel adjoints
    acc_t acc = {0, 0, 0, 0};
    if (M && 64 * ch + 16 * wave < dd.q)
      acc = disc_gemm_k<real>(R + (size_t)(16 * G + m) * ldo + 4 * g, M + (size_t)(4 * g) * ldo + j, ldo,
                              disc_kblocks(NO));
    real l1p = 0, l2p = 0, bsum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = TR::out_row(lane, r);
      const real dN = real(2) * acc[r];
      const real g0 = real(2) * Rv[r] + dN * c1 * Uxv[r];
      Gs[(0 * 16 + p) * GL + jl] = g0;
      Gs[(1 * 16 + p) * GL + jl] = dN * c1 * Uv[r];
      Gs[(2 * 16 + p) * GL + jl] = -c2 * dN;
      bsum += g0;
      l1p += dN * Uv[r] * Uxv[r];
      l2p -= c2 * dN * Uxxv[r];
    }
    // db_out[j] = sum over the 16 points: 4 registers here, the other 12 in the lanes m + 16, 32, 48
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (g == 0 && j < NO) part[(size_t)G * R_cols + nd.off_b[H] + j] = bsum;
    if (dd.identify) {
      l1p = wave_sum(l1p);
      l2p = wave_sum(l2p);
      if (lane == 0) { wsum[wave * 2 + 0] = l1p; wsum[wave * 2 + 1] = l2p; }
    }
  }
#pragma unroll
  for (int n = 0; n < 3 * 16 * WP / 256; ++n) {
    const int i = tid + 256 * n, c = i / (16 * WP), rem = i % (16 * WP);
    a3s[c * CS + (rem / WP) * LD + rem % WP] = pa[n];
  }
  wt_store<real, NT, 64>(pw, wos, OL, tid);
  __syncthreads();
  if (dd.identify && tid < 2)
    lamp[(size_t)(G * dd.n_chunks + ch) * 2 + tid] = (wsum[tid] + wsum[2 + tid]) + (wsum[4 + tid] + wsum[6 + tid]);

  {  // dW_out[k][j] partial of this group: sum over 3 channels x 16 points of A3_c[p][k] G_c[p][j]
    acc_t acc[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) acc[rt] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int p = 4 * ks + g;
        const real b = Gs[(c * 16 + p) * GL + jl];
        const real* ap = a3s + c * CS + p * LD + m;
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) acc[rt] = TR::mfma(ap[16 * rt], b, acc[rt]);
      }
    real* __restrict__ dst = part + (size_t)G * R_cols + nd.off_w[H];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * rt + TR::out_row(lane, r);
        if (k < W && j < NO) dst[(size_t)k * NO + j] = acc[rt][r];
      }
  }

  // partial adjoint of the last hidden layer's output channels over this chunk's 64 columns
  for (int kt = wave; kt < NT; kt += 4) {
    const int k = 16 * kt + m;
    acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int jj = 4 * ks + g;
      const real b = wos[k * OL + jj];
      acc0 = TR::mfma(Gs[(0 * 16 + m) * GL + jj], b, acc0);
      acc1 = TR::mfma(Gs[(1 * 16 + m) * GL + jj], b, acc1);
      acc2 = TR::mfma(Gs[(2 * 16 + m) * GL + jj], b, acc2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t gp = 16 * G + TR::out_row(lane, r);
      dAp[(((size_t)ch * 3 + 0) * n_pad + gp) * WP + k] = acc0[r];
      dAp[(((size_t)ch * 3 + 1) * n_pad + gp) * WP + k] = acc1[r];
      dAp[(((size_t)ch * 3 + 2) * n_pad + gp) * WP + k] = acc2[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// reverse, hidden layers (one workgroup per 16-point group).  Adjoint algebra: SURVEY.md Appendix A.3 with the
// t channel dropped (oracle/mlp.py taylor_backward).
// ---------------------------------------------------------------------------------------------------------
template <typename real, int NT>
__global__ __launch_bounds__(256) void k_disc_bwd_hidden(NetDesc nd, DiscDesc dd, const real* __restrict__ th,
                                                         const int* __restrict__ ginfo,
                                                         const real* __restrict__ xs, real lbx, real sx,
                                                         const real* __restrict__ Ast,
                                                         const real* __restrict__ dAp,
                                                         const real* __restrict__ lossp,
                                                         const real* __restrict__ lamp,
                                                         real* __restrict__ part, int R_cols) {
  using TR = FusedTraits<real>;
  using acc_t = typename TR::acc_t;
  using GEO = DiscGeo<NT>;
  constexpr int WP = GEO::WP, LD = GEO::LD, CS = GEO::CS, TLD = GEO::TLD;
  constexpr int EPT = 16 * WP / 256;                       // (point, feature) elements per thread
  extern __shared__ __attribute__((aligned(16))) char disc_smem[];
  const int G = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = nd.width, H = nd.n_hidden, n_pad = dd.n_pad;
  const int m = lane & 15, g = lane >> 4;
  real* adj = reinterpret_cast<real*>(disc_smem);   // [3][16][LD] adjoint of the layer's output channels
  real* gz = adj + 3 * CS;                          // adjoint of the pre-activation channels
  real* inp = gz + 3 * CS;                          // the layer's input channels
  real* wt = inp + 3 * CS;                          // [WP][TLD] weights of the layer being reversed
  real* hs = wt + WP * TLD;                         // [16] normalised inputs of the group's points
  real* __restrict__ row = part + (size_t)G * R_cols;

  // in flight from the start: weights of the last hidden layer, its stash and the stash below it
  real pre[GEO::NPRE];
  if (H > 1) wt_load<real, NT, WP>(pre, th + nd.off_w[H - 1], W, W, W, tid);
  real sc[3][EPT], sp[3][EPT];                             // stash of layer l / layer l - 1
  auto load_stash = [&](real (&dst)[3][EPT], int l) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = tid + 256 * e;
        dst[c][e] = Ast[(((size_t)l * 3 + c) * n_pad + 16 * G + i / WP) * WP + i % WP];
      }
  };
  load_stash(sc, H - 1);
  if (H > 1) load_stash(sp, H - 2);

  {  // adjoint of the last hidden layer = sum of the chunk partials (fixed order), 4 chunks x all of this
     // thread's elements in flight per round
    const int nch = dd.n_chunks;
    const size_t cstride = (size_t)3 * n_pad * WP;
    real s[3][EPT];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < EPT; ++e) s[c][e] = 0;
    for (int c0 = 0; c0 < nch; c0 += 4) {
      real v[3][EPT][4];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const int i = tid + 256 * e;
          const real* src = dAp + ((size_t)c * n_pad + 16 * G + i / WP) * WP + i % WP + (size_t)c0 * cstride;
#pragma unroll
          for (int u = 0; u < 4; ++u) v[c][e][u] = c0 + u < nch ? src[(size_t)u * cstride] : real(0);
        }
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < EPT; ++e) s[c][e] += (v[c][e][0] + v[c][e][1]) + (v[c][e][2] + v[c][e][3]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = tid + 256 * e;
        adj[c * CS + (i / WP) * LD + i % WP] = s[c][e];
      }
    if (tid < 16) hs[tid] = sx * (xs[16 * G + tid] - lbx) - real(1);
  }
  if (tid == 0) {   // loss slot of this group's stage set; identification parameters
    real s = 0;
    for (int ch = 0; ch < dd.n_chunks; ++ch) s += lossp[G * dd.n_chunks + ch];
    const int set = ginfo[G] & 0xff;
    for (int k = 0; k < 4; ++k) row[nd.n_theta + k] = (k == (set == 0 ? 0 : 1)) ? s : real(0);
    if (dd.identify)
      for (int k = 0; k < 2; ++k) {
        real t = 0;
        for (int ch = 0; ch < dd.n_chunks; ++ch) t += lamp[(size_t)(G * dd.n_chunks + ch) * 2 + k];
        row[nd.n_net + k] = t;
      }
  }
  __syncthreads();

  const int ksteps = (W + 3) / 4;
  for (int l = H - 1; l >= 0; --l) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + 256 * e, p = i / WP, j = i % WP;
      const real a = sc[0][e], zp = sc[1][e], zr = sc[2][e];
      const real hb = adj[0 * CS + p * LD + j], pb = adj[1 * CS + p * LD + j], rb = adj[2 * CS + p * LD + j];
      const real d1 = real(1) - a * a, d2 = real(-2) * a * d1, d3 = real(-2) * d1 * (real(1) - real(3) * a * a);
      gz[0 * CS + p * LD + j] = d1 * hb + d2 * (zp * pb + zr * rb) + d3 * zp * zp * rb;
      gz[1 * CS + p * LD + j] = d1 * pb + real(2) * d2 * zp * rb;
      gz[2 * CS + p * LD + j] = d1 * rb;
      if (l >= 1) {
        const real a0 = sp[0][e], zp0 = sp[1][e], zr0 = sp[2][e];
        const real e1 = real(1) - a0 * a0, e2 = real(-2) * a0 * e1;
        inp[0 * CS + p * LD + j] = a0;
        inp[1 * CS + p * LD + j] = e1 * zp0;
        inp[2 * CS + p * LD + j] = e2 * zp0 * zp0 + e1 * zr0;
      }
    }
    if (l >= 1) wt_store<real, NT, WP>(pre, wt, TLD, tid);
    __syncthreads();
    // next iteration's operands
    if (l >= 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < EPT; ++e) sc[c][e] = sp[c][e];
      if (l >= 2) {
        load_stash(sp, l - 2);
        wt_load<real, NT, WP>(pre, th + nd.off_w[l - 1], W, W, W, tid);
      }
    }
    if (tid < W) {   // bias gradient; dense 0 has one input row (h, h_x = sx, h_xx = 0)
      real sb = 0, sw = 0;
      for (int p = 0; p < 16; ++p) {
        const real zb = gz[0 * CS + p * LD + tid];
        sb += zb;
        if (l == 0) sw += hs[p] * zb + sx * gz[1 * CS + p * LD + tid];
      }
      row[nd.off_b[l] + tid] = sb;
      if (l == 0) row[nd.off_w[0] + tid] = sw;
    }
    if (l >= 1) {
      for (int ct = wave; ct < NT; ct += 4) {   // dW_l[k][j], j in this wave's column tile
        const int j = 16 * ct + m;
        acc_t acc[NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) acc[rt] = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int p = 4 * ks + g;
            const real b = gz[c * CS + p * LD + j];
            const real* ap = inp + c * CS + p * LD + m;
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) acc[rt] = TR::mfma(ap[16 * rt], b, acc[rt]);
          }
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * rt + TR::out_row(lane, r);
            if (k < W && j < W) row[nd.off_w[l] + k * W + j] = acc[rt][r];
          }
      }
      for (int kt = wave; kt < NT; kt += 4) {   // adjoint of the input channels: gz_c W_l^T
        const int k = 16 * kt + m;
        acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
        for (int ks = 0; ks < ksteps; ++ks) {
          const int jj = 4 * ks + g;
          const real b = wt[k * TLD + jj];
          const real* ap = gz + m * LD + jj;
          acc0 = TR::mfma(ap[0], b, acc0);
          acc1 = TR::mfma(ap[CS], b, acc1);
          acc2 = TR::mfma(ap[2 * CS], b, acc2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = TR::out_row(lane, r);
          adj[0 * CS + p * LD + k] = acc0[r];
          adj[1 * CS + p * LD + k] = acc1[r];
          adj[2 * CS + p * LD + k] = acc2[r];
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace pinn
