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
// then the ordinary k_reduce_rows over the per-group gradient rows.  Operand convention of the 16x16x4 MFMA as
// used everywhere below: lane = (m | n) + 16 g;  A operand = A[m][4 s + g],  B operand = B[4 s + g][n],
// accumulator register r = D[out_row(lane, r)][n]  (out_row differs between the f32 and f64 instruction).
#pragma once
#include "kernels_fused20.h"

namespace pinn {

struct DiscDesc {
  int n_groups;      // 16-point groups over all stage sets (a group never straddles two sets)
  int n_pad;         // 16 * n_groups
  int ldo;           // row stride of every [point][column] array: n_out rounded up to 64
  int n_chunks;      // ldo / 64
  int q;             // outputs entering N  (q <= n_out)
  int wp;            // hidden width rounded up to 16
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

inline size_t disc_fwd_lds(int wp, size_t rs) { return (size_t)2 * 3 * 16 * (wp + 4) * rs; }
inline size_t disc_hid_lds(int wp, size_t rs) { return (size_t)3 * 3 * 16 * (wp + 4) * rs; }

// ---------------------------------------------------------------------------------------------------------
// forward
//   Ast [H][3][n_pad][wp]   per hidden layer (a, z_x, z_xx): tanh output and the derivative channels of the
//                           pre-activation -- what the reverse sweep needs (written by chunk 0 only)
//   A3  [3][n_pad][wp]      output channels (a, a_x, a_xx) of the last hidden layer
//   U3  [3][n_pad][ldo]     (U, U_x, U_xx) of the output layer;  Nn [n_pad][ldo] = c1 U U_x - c2 U_xx (0 beyond q)
// ---------------------------------------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(256) void k_disc_fwd(NetDesc nd, DiscDesc dd, const real* __restrict__ th,
                                                  const real* __restrict__ xs, real lbx, real sx, real c1_in,
                                                  real c2_in, real* __restrict__ Ast, real* __restrict__ A3,
                                                  real* __restrict__ U3, real* __restrict__ Nn, int write_stash) {
  using TR = FusedTraits<real>;
  using acc_t = typename TR::acc_t;
  extern __shared__ __attribute__((aligned(16))) char disc_smem[];
  real* act = reinterpret_cast<real*>(disc_smem);          // [2][3][16][LD]
  const int G = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = nd.width, H = nd.n_hidden, NO = nd.n_out, wp = dd.wp, LD = wp + 4, CS = 16 * LD;
  const int n_pad = dd.n_pad, ldo = dd.ldo;
  const int m = lane & 15, g = lane >> 4;
  const bool stash = write_stash && ch == 0;
  real c1, c2;
  disc_coefs(nd, dd, th, c1_in, c2_in, c1, c2);

  // dense 0 (fan_in 1): h = sx (x - lb) - 1, h_x = sx, h_xx = 0
  for (int i = tid; i < 16 * wp; i += 256) {
    const int p = i / wp, j = i - p * wp;
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
      const size_t o = (size_t)(16 * G + p) * wp + j;
      Ast[(size_t)0 * n_pad * wp + o] = a;
      Ast[(size_t)1 * n_pad * wp + o] = zp;
      Ast[(size_t)2 * n_pad * wp + o] = 0;
    }
  }
  __syncthreads();

  int cur = 0;
  const int ksteps = (W + 3) / 4;
  for (int l = 1; l < H; ++l) {
    const real* __restrict__ Wl = th + nd.off_w[l];
    const real* __restrict__ bl = th + nd.off_b[l];
    const real* in = act + cur * 3 * CS;
    real* out = act + (cur ^ 1) * 3 * CS;
    for (int ct = wave; ct < wp / 16; ct += 4) {
      const int j = 16 * ct + m;
      acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
      for (int ks = 0; ks < ksteps; ++ks) {
        const int k = 4 * ks + g;
        const real b = (k < W && j < W) ? Wl[k * W + j] : real(0);
        const real* ap = in + m * LD + k;
        acc0 = TR::mfma(ap[0], b, acc0);
        acc1 = TR::mfma(ap[CS], b, acc1);
        acc2 = TR::mfma(ap[2 * CS], b, acc2);
      }
      const real bj = j < W ? bl[j] : real(0);
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
          const size_t o = (size_t)(16 * G + p) * wp + j;
          Ast[((size_t)l * 3 + 0) * n_pad * wp + o] = a;
          Ast[((size_t)l * 3 + 1) * n_pad * wp + o] = zp;
          Ast[((size_t)l * 3 + 2) * n_pad * wp + o] = zr;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  const real* in = act + cur * 3 * CS;
  if (stash)
    for (int i = tid; i < 3 * 16 * wp; i += 256) {
      const int c = i / (16 * wp), rem = i - c * 16 * wp, p = rem / wp, j = rem - p * wp;
      A3[((size_t)c * n_pad + 16 * G + p) * wp + j] = in[c * CS + p * LD + j];
    }

  {  // output-layer chunk: columns 64 ch + 16 wave + (0..15)
    const real* __restrict__ Wo = th + nd.off_w[H];
    const real* __restrict__ bo = th + nd.off_b[H];
    const int j = 64 * ch + 16 * wave + m;
    acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const int k = 4 * ks + g;
      const real b = (k < W && j < NO) ? Wo[(size_t)k * NO + j] : real(0);
      const real* ap = in + m * LD + k;
      acc0 = TR::mfma(ap[0], b, acc0);
      acc1 = TR::mfma(ap[CS], b, acc1);
      acc2 = TR::mfma(ap[2 * CS], b, acc2);
    }
    const real bj = j < NO ? bo[j] : real(0);
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
  acc_t acc = {0, 0, 0, 0};
  if (MT) {
    const int ksteps = (dd.q + 15) / 16;
    const real* __restrict__ arow = Nn + (size_t)(16 * G + m) * ldo + 4 * g;
    const real* __restrict__ bcol = MT + (size_t)(4 * g) * ldo + j;
#pragma unroll 2
    for (int S = 0; S < ksteps; ++S) {
      const vec4<real> a4 = *reinterpret_cast<const vec4<real>*>(arow + 16 * S);
      const real* bp = bcol + (size_t)16 * S * ldo;
      const real b0 = bp[0], b1 = bp[ldo], b2 = bp[2 * ldo], b3 = bp[3 * ldo];
      acc = TR::mfma(a4.x, b0, acc);
      acc = TR::mfma(a4.y, b1, acc);
      acc = TR::mfma(a4.z, b2, acc);
      acc = TR::mfma(a4.w, b3, acc);
    }
  }
  real l = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = TR::out_row(lane, r);
    const size_t gp = 16 * G + p;
    real v = acc[r] + U[gp * ldo + j];
    if (mode == 0) {
      v = (p < n_valid && j < n_out) ? v - tgt[gp] : real(0);
      l += v * v;
    }
    R[gp * ldo + j] = v;
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
// lamp [n_groups * n_chunks][2] partial (dL/dc1-ish) sums for the identification parameters.
// ---------------------------------------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(256) void k_disc_bwd_out(NetDesc nd, DiscDesc dd, const real* __restrict__ th,
                                                      const int* __restrict__ ginfo,
                                                      const real* __restrict__ M0, const real* __restrict__ M1,
                                                      const real* __restrict__ R, const real* __restrict__ U3,
                                                      const real* __restrict__ A3, real c1_in, real c2_in,
                                                      real* __restrict__ part, int R_cols,
                                                      real* __restrict__ dAp, real* __restrict__ lamp) {
  using TR = FusedTraits<real>;
  using acc_t = typename TR::acc_t;
  constexpr int GL = 68;
  __shared__ real Gs[3][16][GL];
  __shared__ real wsum[4][2];
  const int G = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = nd.width, H = nd.n_hidden, NO = nd.n_out, wp = dd.wp, ldo = dd.ldo, n_pad = dd.n_pad;
  const int m = lane & 15, g = lane >> 4;
  const int set = ginfo[G] & 0xff;
  const real* __restrict__ M = set == 0 ? M0 : M1;
  real c1, c2;
  disc_coefs(nd, dd, th, c1_in, c2_in, c1, c2);
  const int jl = 16 * wave + m, j = 64 * ch + jl;

  {  // dN tile = 2 R M  (K = n_out), then the three output-channel adjoints
    acc_t acc = {0, 0, 0, 0};
    if (M && 64 * ch + 16 * wave < dd.q) {
      const int ksteps = (NO + 15) / 16;
      const real* __restrict__ arow = R + (size_t)(16 * G + m) * ldo + 4 * g;
      const real* __restrict__ bcol = M + (size_t)(4 * g) * ldo + j;
#pragma unroll 2
      for (int S = 0; S < ksteps; ++S) {
        const vec4<real> a4 = *reinterpret_cast<const vec4<real>*>(arow + 16 * S);
        const real* bp = bcol + (size_t)16 * S * ldo;
        const real b0 = bp[0], b1 = bp[ldo], b2 = bp[2 * ldo], b3 = bp[3 * ldo];
        acc = TR::mfma(a4.x, b0, acc);
        acc = TR::mfma(a4.y, b1, acc);
        acc = TR::mfma(a4.z, b2, acc);
        acc = TR::mfma(a4.w, b3, acc);
      }
    }
    real l1p = 0, l2p = 0, bsum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = TR::out_row(lane, r);
      const size_t o = (size_t)(16 * G + p) * ldo + j;
      const real Rv = R[o], U = U3[o], Ux = U3[(size_t)n_pad * ldo + o];
      const real dN = real(2) * acc[r];
      const real g0 = real(2) * Rv + dN * c1 * Ux;
      Gs[0][p][jl] = g0;
      Gs[1][p][jl] = dN * c1 * U;
      Gs[2][p][jl] = -c2 * dN;
      bsum += g0;
      if (dd.identify) {
        l1p += dN * U * Ux;
        l2p -= c2 * dN * U3[(size_t)2 * n_pad * ldo + o];
      }
    }
    // db_out[j] = sum over the 16 points: 4 registers here, the other 12 in the lanes m + 16, 32, 48
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (g == 0 && j < NO) part[(size_t)G * R_cols + nd.off_b[H] + j] = bsum;
    if (dd.identify) {
      l1p = wave_sum(l1p);
      l2p = wave_sum(l2p);
      if (lane == 0) { wsum[wave][0] = l1p; wsum[wave][1] = l2p; }
    }
  }
  __syncthreads();
  if (dd.identify && tid < 2)
    lamp[(size_t)(G * dd.n_chunks + ch) * 2 + tid] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);

  {  // dW_out[k][j] partial of this group: sum over 3 channels x 16 points of A3_c[p][k] G_c[p][j]
    const int nrt = wp / 16;
    acc_t acc[8];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) acc[rt] = acc_t{0, 0, 0, 0};
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int p = 4 * ks + g;
        const real b = Gs[c][p][jl];
        const real* __restrict__ ap = A3 + ((size_t)c * n_pad + 16 * G + p) * wp + m;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
          if (rt < nrt) acc[rt] = TR::mfma(ap[16 * rt], b, acc[rt]);
      }
    real* __restrict__ dst = part + (size_t)G * R_cols + nd.off_w[H];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
      if (rt < nrt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * rt + TR::out_row(lane, r);
          if (k < W && j < NO) dst[(size_t)k * NO + j] = acc[rt][r];
        }
  }

  {  // partial adjoint of the last hidden layer's output channels over this chunk's 64 columns
    const real* __restrict__ Wo = th + nd.off_w[H];
    for (int kt = wave; kt < wp / 16; kt += 4) {
      const int k = 16 * kt + m;
      acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int jj = 4 * ks + g, jg = 64 * ch + jj;
        const real b = (k < W && jg < NO) ? Wo[(size_t)k * NO + jg] : real(0);
        acc0 = TR::mfma(Gs[0][m][jj], b, acc0);
        acc1 = TR::mfma(Gs[1][m][jj], b, acc1);
        acc2 = TR::mfma(Gs[2][m][jj], b, acc2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t gp = 16 * G + TR::out_row(lane, r);
        dAp[(((size_t)ch * 3 + 0) * n_pad + gp) * wp + k] = acc0[r];
        dAp[(((size_t)ch * 3 + 1) * n_pad + gp) * wp + k] = acc1[r];
        dAp[(((size_t)ch * 3 + 2) * n_pad + gp) * wp + k] = acc2[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// reverse, hidden layers (one workgroup per 16-point group).  Adjoint algebra: SURVEY.md Appendix A.3 with the
// t channel dropped (oracle/mlp.py taylor_backward).
// ---------------------------------------------------------------------------------------------------------
template <typename real>
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
  extern __shared__ __attribute__((aligned(16))) char disc_smem[];
  const int G = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = nd.width, H = nd.n_hidden, wp = dd.wp, LD = wp + 4, CS = 16 * LD, n_pad = dd.n_pad;
  const int m = lane & 15, g = lane >> 4;
  real* adj = reinterpret_cast<real*>(disc_smem);   // [3][16][LD] adjoint of the layer's output channels
  real* gz = adj + 3 * CS;                          // adjoint of the pre-activation channels
  real* inp = gz + 3 * CS;                          // the layer's input channels
  real* __restrict__ row = part + (size_t)G * R_cols;

  for (int i = tid; i < 3 * 16 * wp; i += 256) {
    const int c = i / (16 * wp), rem = i - c * 16 * wp, p = rem / wp, k = rem - p * wp;
    real s = 0;
    for (int ch = 0; ch < dd.n_chunks; ++ch) s += dAp[(((size_t)ch * 3 + c) * n_pad + 16 * G + p) * wp + k];
    adj[c * CS + p * LD + k] = s;
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

  for (int l = H - 1; l >= 0; --l) {
    for (int i = tid; i < 16 * wp; i += 256) {
      const int p = i / wp, j = i - p * wp;
      const size_t o = (size_t)(16 * G + p) * wp + j;
      const real a = Ast[((size_t)l * 3 + 0) * n_pad * wp + o], zp = Ast[((size_t)l * 3 + 1) * n_pad * wp + o],
                 zr = Ast[((size_t)l * 3 + 2) * n_pad * wp + o];
      const real hb = adj[0 * CS + p * LD + j], pb = adj[1 * CS + p * LD + j], rb = adj[2 * CS + p * LD + j];
      const real d1 = real(1) - a * a, d2 = real(-2) * a * d1, d3 = real(-2) * d1 * (real(1) - real(3) * a * a);
      gz[0 * CS + p * LD + j] = d1 * hb + d2 * (zp * pb + zr * rb) + d3 * zp * zp * rb;
      gz[1 * CS + p * LD + j] = d1 * pb + real(2) * d2 * zp * rb;
      gz[2 * CS + p * LD + j] = d1 * rb;
      if (l >= 1) {
        const real a0 = Ast[((size_t)(l - 1) * 3 + 0) * n_pad * wp + o],
                   zp0 = Ast[((size_t)(l - 1) * 3 + 1) * n_pad * wp + o],
                   zr0 = Ast[((size_t)(l - 1) * 3 + 2) * n_pad * wp + o];
        const real e1 = real(1) - a0 * a0, e2 = real(-2) * a0 * e1;
        inp[0 * CS + p * LD + j] = a0;
        inp[1 * CS + p * LD + j] = e1 * zp0;
        inp[2 * CS + p * LD + j] = e2 * zp0 * zp0 + e1 * zr0;
      }
    }
    __syncthreads();
    if (tid < W) {   // bias gradient; dense 0 has one input row (h, h_x = sx, h_xx = 0)
      real sb = 0, sw = 0;
      for (int p = 0; p < 16; ++p) {
        const real zb = gz[0 * CS + p * LD + tid];
        sb += zb;
        if (l == 0) sw += (sx * (xs[16 * G + p] - lbx) - real(1)) * zb + sx * gz[1 * CS + p * LD + tid];
      }
      row[nd.off_b[l] + tid] = sb;
      if (l == 0) row[nd.off_w[0] + tid] = sw;
    }
    if (l >= 1) {
      const real* __restrict__ Wl = th + nd.off_w[l];
      const int nrt = wp / 16;
      for (int ct = wave; ct < nrt; ct += 4) {   // dW_l[k][j], j in this wave's column tile
        const int j = 16 * ct + m;
        acc_t acc[8];
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) acc[rt] = acc_t{0, 0, 0, 0};
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int p = 4 * ks + g;
            const real b = gz[c * CS + p * LD + j];
            const real* ap = inp + c * CS + p * LD + m;
#pragma unroll
            for (int rt = 0; rt < 8; ++rt)
              if (rt < nrt) acc[rt] = TR::mfma(ap[16 * rt], b, acc[rt]);
          }
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
          if (rt < nrt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * rt + TR::out_row(lane, r);
              if (k < W && j < W) row[nd.off_w[l] + k * W + j] = acc[rt][r];
            }
      }
      const int ksteps = (W + 3) / 4;
      for (int kt = wave; kt < nrt; kt += 4) {   // adjoint of the input channels: gz_c W_l^T
        const int k = 16 * kt + m;
        acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll 4
        for (int ks = 0; ks < ksteps; ++ks) {
          const int jj = 4 * ks + g;
          const real b = (k < W && jj < W) ? Wl[k * W + jj] : real(0);
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
