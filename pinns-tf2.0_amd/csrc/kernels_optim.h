// kernels_optim.h -- deterministic gradient reduction, Adam, device-resident L-BFGS.
// All optimiser state is float64 whatever the kernel compute dtype (the reference is
// float64 end to end, utils/neuralnetwork.py:24-26); the compute kernels read a `real`
// mirror of the weights that these kernels keep in sync.
#pragma once
#include "kernels_fused20r.h"

namespace pinn {

// gl[c] = sum over rows of part[row*R + c] in a fixed order (bit-reproducible).
// gl layout: [0, n_theta) gradient | n_theta+0..2 loss parts (residual, data, boundary).
// Block = 64 columns x 4 row-quarters (row r belongs to quarter r & 3); each thread keeps 8
// independent accumulators so that 8 loads are in flight, then the quarters are combined
// through LDS in index order.  Grid = ceil(R / 64) blocks of 256 threads.
constexpr int RED_COLS = 64;

template <typename real>
__device__ __forceinline__ double reduce_column(const real* __restrict__ part, int n_rows, int R,
                                                int c, int q, double (*sh)[RED_COLS]) {
  const int cl = threadIdx.x & 63;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  if (c < R) {
    const real* __restrict__ p = part + c;
    int r = q;
    for (; r + 28 < n_rows; r += 32) {
      a0 += (double)p[(size_t)(r + 0) * R];  a1 += (double)p[(size_t)(r + 4) * R];
      a2 += (double)p[(size_t)(r + 8) * R];  a3 += (double)p[(size_t)(r + 12) * R];
      a4 += (double)p[(size_t)(r + 16) * R]; a5 += (double)p[(size_t)(r + 20) * R];
      a6 += (double)p[(size_t)(r + 24) * R]; a7 += (double)p[(size_t)(r + 28) * R];
    }
    for (; r < n_rows; r += 4) a0 += (double)p[(size_t)r * R];
  }
  sh[q][cl] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  __syncthreads();
  return (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

template <typename real>
__global__ __launch_bounds__(256) void k_reduce_rows(const real* __restrict__ part, int n_rows,
                                                     int R, double* __restrict__ gl) {
  __shared__ double sh[4][RED_COLS];
  const int c = blockIdx.x * RED_COLS + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const double g = reduce_column(part, n_rows, R, c, q, sh);
  if (q == 0 && c < R) gl[c] = g;
}

// Single-GPU Adam step fused behind the reduction (no all-reduce in between): same arithmetic as
// k_reduce_rows followed by k_adam.  loss3 (may be null) <- the three loss parts of this step.
template <typename real>
__global__ __launch_bounds__(256) void k_reduce_adam(const real* __restrict__ part, int n_rows,
                                                     int R, double* __restrict__ gl, int n,
                                                     double* __restrict__ theta,
                                                     real* __restrict__ theta_r,
                                                     double* __restrict__ m, double* __restrict__ v,
                                                     double alpha, double b1, double b2, double eps,
                                                     double* __restrict__ loss3, NetDesc nd,
                                                     float* __restrict__ img) {
  __shared__ double sh[4][RED_COLS];
  const int c = blockIdx.x * RED_COLS + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const double g = reduce_column(part, n_rows, R, c, q, sh);
  if (q != 0 || c >= R) return;
  gl[c] = g;
  if (c < n) {
    const double mi = m[c] + (1.0 - b1) * (g - m[c]);
    const double vi = v[c] + (1.0 - b2) * (g * g - v[c]);
    m[c] = mi;
    v[c] = vi;
    const double t = theta[c] - alpha * mi / (sqrt(vi) + eps);
    theta[c] = t;
    theta_r[c] = (real)t;
    pack_store(nd, img, c, (float)t);
  } else if (loss3 && c < n + 3) {
    loss3[c - n] = g;
  }
}

// TF-2.0 ResourceApplyAdam (reference call site
// utils/neuralnetwork.py:114): m += (1-b1)(g-m); v += (1-b2)(g^2-v);
// theta -= alpha*m/(sqrt(v)+eps), alpha = lr*sqrt(1-b2^t)/(1-b1^t) computed by the host.
template <typename real>
__global__ void k_adam(int n, const double* __restrict__ gl, double* __restrict__ theta,
                       real* __restrict__ theta_r, double* __restrict__ m,
                       double* __restrict__ v, double alpha, double b1, double b2, double eps,
                       double* __restrict__ loss3, int n_theta, NetDesc nd,
                       float* __restrict__ img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 && loss3) loss3[i] = gl[n_theta + i];
  if (i >= n) return;
  const double g = gl[i];
  const double mi = m[i] + (1.0 - b1) * (g - m[i]);
  const double vi = v[i] + (1.0 - b2) * (g * g - v[i]);
  m[i] = mi;
  v[i] = vi;
  const double t = theta[i] - alpha * mi / (sqrt(vi) + eps);
  theta[i] = t;
  theta_r[i] = (real)t;
  pack_store(nd, img, i, (float)t);
}

template <typename real>
__global__ void k_cast_weights(int n, const double* __restrict__ theta, real* __restrict__ theta_r,
                               NetDesc nd, float* __restrict__ img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { theta_r[i] = (real)theta[i]; pack_store(nd, img, i, (float)theta[i]); }
}

// ---------------------------------------------------------------------------------------------
// L-BFGS (utils/custom_lbfgs.py:39-236), one 1024-thread workgroup, float64.
// ---------------------------------------------------------------------------------------------
struct LbfgsState {
  int n_iter;        // state.nIter
  int func_eval;     // currentFuncEval
  int hist_len;      // len(old_dirs)
  int hist_head;     // ring index of the oldest pair
  int done;          // 0 running | 1 nIter==maxIter | 2 gtd>-tolX | 3 maxEval | 4 tolFun | 5 step<tolX | 6 |df|<tolX | 7 initial tolFun
  int n_logged;      // log_fn calls so far
  int pad0, pad1;
  double Hdiag, t, f, f_old, final_loss;
};

constexpr int LB_THREADS = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
  // fixed-shape reduction: wave DPP sum, then the 16 wave totals in index order
  const double w = wave_sum(v);
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wid] = w;
  __syncthreads();
  double tot = 0;
#pragma unroll
  for (int i = 0; i < LB_THREADS / 64; ++i) tot += sh[i];
  return tot;
}

// Compute the search direction, step, and advance x (custom_lbfgs.py:84-174).
//   g    : gradient at the current x (= gl from the last evaluation)
//   x    : L-BFGS iterate;  theta/theta_r: model weights (set only when an evaluation follows,
//          i.e. when nIter != maxIter -- custom_lbfgs.py:176 + neuralnetwork.py:94)
template <typename real>
__global__ __launch_bounds__(LB_THREADS) void k_lbfgs_step(
    int n, int max_iter, int n_corr, double lr, double tol_x, LbfgsState* __restrict__ st,
    const double* __restrict__ g, double* __restrict__ x, double* __restrict__ theta,
    real* __restrict__ theta_r, double* __restrict__ d, double* __restrict__ g_old,
    double* __restrict__ Sh, double* __restrict__ Yh, double* __restrict__ ro,
    double* __restrict__ al, double* __restrict__ q, NetDesc nd, float* __restrict__ img) {
  __shared__ double sh[LB_THREADS / 64];
  __shared__ int s_flag;
  if (st->done) return;
  const int tid = threadIdx.x;
  const int n_iter = st->n_iter + 1;
  int hist_len = st->hist_len, head = st->hist_head;
  double Hdiag = st->Hdiag;
  const double t_prev = st->t;

  if (n_iter == 1) {                                   // :90-95
    for (int i = tid; i < n; i += LB_THREADS) d[i] = -g[i];
    hist_len = 0; head = 0; Hdiag = 1.0;
  } else {
    // y = g - g_old, s = d*t  (:98-100); staged in q (y) and al-free scratch d stays d
    double ys_p = 0, yy_p = 0;
    for (int i = tid; i < n; i += LB_THREADS) {
      const double y = g[i] - g_old[i], s = d[i] * t_prev;
      ys_p += y * s; yy_p += y * y;
    }
    const double ys = block_sum(ys_p, sh);
    const double yy = block_sum(yy_p, sh);
    if (ys > 1e-10) {                                  // :102-114
      int slot;
      if (hist_len == n_corr) { slot = head; head = (head + 1) % n_corr; }
      else { slot = (head + hist_len) % n_corr; hist_len += 1; }
      for (int i = tid; i < n; i += LB_THREADS) {
        Sh[(size_t)slot * n + i] = d[i] * t_prev;
        Yh[(size_t)slot * n + i] = g[i] - g_old[i];
      }
      if (tid == 0) ro[slot] = 1.0 / ys;               // 1/dot(old_stps[i], old_dirs[i]) (:121-123)
      Hdiag = ys / yy;
    }
    __syncthreads();
    // two-loop recursion (:126-141); logical index i -> ring slot (head+i)%n_corr
    for (int i = tid; i < n; i += LB_THREADS) q[i] = -g[i];
    __syncthreads();
    for (int li = hist_len - 1; li >= 0; --li) {
      const int slot = (head + li) % n_corr;
      double p = 0;
      for (int i = tid; i < n; i += LB_THREADS) p += Sh[(size_t)slot * n + i] * q[i];
      const double a = block_sum(p, sh) * ro[slot];
      if (tid == 0) al[slot] = a;
      for (int i = tid; i < n; i += LB_THREADS) q[i] -= a * Yh[(size_t)slot * n + i];
      __syncthreads();
    }
    for (int i = tid; i < n; i += LB_THREADS) q[i] *= Hdiag;          // r = q*Hdiag
    __syncthreads();
    for (int li = 0; li < hist_len; ++li) {
      const int slot = (head + li) % n_corr;
      double p = 0;
      for (int i = tid; i < n; i += LB_THREADS) p += Yh[(size_t)slot * n + i] * q[i];
      const double be = block_sum(p, sh) * ro[slot];
      const double co = al[slot] - be;
      for (int i = tid; i < n; i += LB_THREADS) q[i] += co * Sh[(size_t)slot * n + i];
      __syncthreads();
    }
    for (int i = tid; i < n; i += LB_THREADS) d[i] = q[i];
  }
  __syncthreads();
  // g_old = g, f_old = f  (:143-145); gtd = g.d (:151)
  double gtd_p = 0, ga_p = 0;
  for (int i = tid; i < n; i += LB_THREADS) {
    const double gi = g[i];
    g_old[i] = gi;
    gtd_p += gi * d[i];
    ga_p += fabs(gi);
  }
  const double gtd = block_sum(gtd_p, sh);
  const double gabs = block_sum(ga_p, sh);
  if (tid == 0) s_flag = (gtd > -tol_x) ? 1 : 0;       // :154-156
  __syncthreads();
  if (s_flag) {
    if (tid == 0) {
      st->n_iter = n_iter; st->hist_len = hist_len; st->hist_head = head; st->Hdiag = Hdiag;
      st->f_old = st->f; st->done = 2;
    }
    return;
  }
  double t;                                            // :159-163
  if (n_iter == 1) { const double inv = 1.0 / gabs; t = inv < 1.0 ? inv : 1.0; }
  else t = lr;
  const bool will_eval = (n_iter != max_iter);         // :176
  for (int i = tid; i < n; i += LB_THREADS) {
    const double xi = x[i] + t * d[i];                 // :174
    x[i] = xi;
    if (will_eval) { theta[i] = xi; theta_r[i] = (real)xi; pack_store(nd, img, i, (float)xi); }
  }
  if (tid == 0) {
    st->n_iter = n_iter; st->hist_len = hist_len; st->hist_head = head; st->Hdiag = Hdiag;
    st->t = t; st->f_old = st->f;
    if (!will_eval) st->done = 1;                      // :192 (nIter == maxIter)
  }
}

// After the re-evaluation: bookkeeping and the break tests of custom_lbfgs.py:185-224.
__global__ __launch_bounds__(LB_THREADS) void k_lbfgs_post(
    int n, int n_theta, int max_iter, double max_eval, double tol_fun, double tol_x,
    LbfgsState* __restrict__ st, const double* __restrict__ gl, const double* __restrict__ d,
    int* __restrict__ log_iters, double* __restrict__ log_losses, int initial) {
  __shared__ double sh[LB_THREADS / 64];
  if (st->done) return;
  const int tid = threadIdx.x;
  const double f = gl[n_theta] + gl[n_theta + 1] + gl[n_theta + 2];
  double ga_p = 0, sa_p = 0;
  const double t = st->t;
  for (int i = tid; i < n; i += LB_THREADS) {
    ga_p += fabs(gl[i]);
    if (!initial) sa_p += fabs(d[i] * t);
  }
  const double gabs = block_sum(ga_p, sh);
  const double sabs = block_sum(sa_p, sh);
  if (tid != 0) return;
  if (initial) {                                       // :65-76
    st->f = f; st->func_eval = 1;
    if (gabs <= tol_fun) st->done = 7;
    return;
  }
  const double f_old = st->f_old;
  st->f = f;
  st->func_eval += 1;
  const int n_iter = st->n_iter;
  if (n_iter == max_iter) { st->done = 1; return; }    // :192 (unreachable: step sets it)
  if ((double)st->func_eval >= max_eval) { st->done = 3; return; }   // :195
  if (gabs <= tol_fun) { st->done = 4; return; }       // :200-203
  if (sabs <= tol_x) { st->done = 5; return; }         // :206-209
  if (fabs(f - f_old) < tol_x) { st->done = 6; return; }   // :212-215
  const int k = st->n_logged;                          // :217-218
  log_iters[k] = n_iter;
  log_losses[k] = f;
  st->n_logged = k + 1;
  if (n_iter == max_iter - 1) st->final_loss = f;      // :223-224
}

// ---------------------------------------------------------------------------------------------
// Compact L-BFGS: same iteration as k_lbfgs_step, restructured so that no kernel contains a
// chain of dependent full-length reductions.
//   k_lbc_dots   one workgroup per history slot: every dot product this iteration needs
//                (s_a.y_c, s_c.y_a, y_a.y_c, s_a.g, y_a.g, y_c.s_c, y_c.y_c, g.g, |g|_1, |t d|_1) in
//                parallel; also materialises the candidate pair s_c = t d, y_c = g - g_old.
//   k_lbc_coef   (a) the bookkeeping/break tests of custom_lbfgs.py:185-224 for the evaluation
//                that precedes this iteration (k_lbfgs_post folded in, `do_post`);
//                (b) accepts the pair (y.s > 1e-10), maintains the Gram matrices
//                SY[a][b] = s_a.y_b, YY[a][b] = y_a.y_b, and runs the two-loop recursion of
//                custom_lbfgs.py:126-141 on *scalars* -- every vector of the recursion lives in
//                span{s_j, y_j, g}, so alpha_i/beta_i follow from the Gram entries alone.  One
//                lane per history slot; each lane carries the running sum it will need when its
//                turn comes, so a recursion step is: lane i finishes alpha_i -> broadcast ->
//                every lane does one fma.  No cross-lane reductions inside the loops.
//                Emits d = cg g + sum_j (cy_j y_j + cs_j s_j), gtd, the step t and the break flag.
//   k_lbc_apply  d, g_old, x += t d (and the model weights when an evaluation follows).
// Mathematically identical to the reference recursion; rounding differs at the 1e-16 level.
// The ring has m+1 slots so that the candidate never overwrites a pair that may still be needed.
// ---------------------------------------------------------------------------------------------
struct LbcExtra {
  int apply, will_eval, pad0, pad1;
  double cg, gtd;
};

constexpr int LBC_THREADS = 256;
constexpr int LBC_NSCAL = 5;       // trailing scalars of the dots array

__device__ __forceinline__ double block_sum256(double v, double* sh) {
  const double w = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// dots layout: [0,M1) s_a.y_c | [M1,2M1) s_c.y_a | [2M1,3M1) y_a.y_c | [3M1,4M1) s_a.g |
//              [4M1,5M1) y_a.g | 5M1+0 y_c.s_c | +1 y_c.y_c | +2 g.g | +3 |g|_1 | +4 |t d|_1
__global__ __launch_bounds__(LBC_THREADS) void k_lbc_dots(
    int n, int M1, const LbfgsState* __restrict__ st, const double* __restrict__ g,
    const double* __restrict__ g_old, const double* __restrict__ d, double* __restrict__ Sh,
    double* __restrict__ Yh, double* __restrict__ dots) {
  __shared__ double sh[4];
  if (st->done) return;
  const int head = st->hist_head, len = st->hist_len;
  const bool first = (st->n_iter == 0);
  int c = head + len; if (c >= M1) c -= M1;
  const int a = blockIdx.x;
  int pos = a - head; if (pos < 0) pos += M1;
  const double t = st->t;
  const int tid = threadIdx.x;
  if (a == c) {
    double ys = 0, yy = 0, sg = 0, yg = 0, gg = 0, ga = 0, sa = 0;
    for (int i = tid; i < n; i += LBC_THREADS) {
      const double gi = g[i];
      gg += gi * gi; ga += fabs(gi);
      if (!first) {
        const double y = gi - g_old[i], s = d[i] * t;
        Sh[(size_t)c * n + i] = s; Yh[(size_t)c * n + i] = y;
        ys += y * s; yy += y * y; sg += s * gi; yg += y * gi; sa += fabs(s);
      }
    }
    ys = block_sum256(ys, sh); yy = block_sum256(yy, sh); sg = block_sum256(sg, sh);
    yg = block_sum256(yg, sh); gg = block_sum256(gg, sh); ga = block_sum256(ga, sh);
    sa = block_sum256(sa, sh);
    if (tid == 0) {
      dots[5 * M1 + 0] = ys; dots[5 * M1 + 1] = yy; dots[5 * M1 + 2] = gg; dots[5 * M1 + 3] = ga;
      dots[5 * M1 + 4] = sa;
      dots[3 * M1 + c] = sg; dots[4 * M1 + c] = yg;
    }
  } else if (pos < len) {
    double say = 0, sya = 0, yya = 0, sg = 0, yg = 0;
    for (int i = tid; i < n; i += LBC_THREADS) {
      const double gi = g[i], sa = Sh[(size_t)a * n + i], ya = Yh[(size_t)a * n + i];
      const double yc = gi - g_old[i], sc = d[i] * t;
      say += sa * yc; sya += sc * ya; yya += ya * yc; sg += sa * gi; yg += ya * gi;
    }
    say = block_sum256(say, sh); sya = block_sum256(sya, sh); yya = block_sum256(yya, sh);
    sg = block_sum256(sg, sh); yg = block_sum256(yg, sh);
    if (tid == 0) {
      dots[a] = say; dots[M1 + a] = sya; dots[2 * M1 + a] = yya;
      dots[3 * M1 + a] = sg; dots[4 * M1 + a] = yg;
    }
  }
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
inline int lbc_ld(int M1) { return M1 | 1; }     // odd leading dimension: column walks hit distinct banks

__global__ __launch_bounds__(LBC_THREADS) void k_lbc_coef(
    int M1, int m, int max_iter, double lr, double tol_x, double tol_fun, double max_eval,
    int do_post, int n_theta, LbfgsState* __restrict__ st, LbcExtra* __restrict__ ex,
    const double* __restrict__ gl, const double* __restrict__ dots, double* __restrict__ SY,
    double* __restrict__ YY, double* __restrict__ ro, double* __restrict__ cs_out,
    double* __restrict__ cy_out, int* __restrict__ log_iters, double* __restrict__ log_losses) {
  extern __shared__ double lsh[];
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int LD = M1 | 1;
  double* sSY = lsh;
  double* sYY = lsh + M1 * LD;
  if (st->done) { if (tid == 0) ex->apply = 0; return; }
  // snapshot of the state (thread 0 writes it back after the barrier below)
  const int n_iter_prev = uni(st->n_iter), fe_prev = uni(st->func_eval), n_logged = uni(st->n_logged);
  int head = uni(st->hist_head), len = uni(st->hist_len);
  double Hdiag = st->Hdiag, f_cur = st->f;
  const double f_old_prev = st->f_old;
  const double ys = dots[5 * M1 + 0], yy = dots[5 * M1 + 1];
  const double gg = dots[5 * M1 + 2], gabs = dots[5 * M1 + 3], sabs = dots[5 * M1 + 4];
  int dn = 0;
  if (do_post) {                                   // custom_lbfgs.py:185-215 for the last evaluation
    f_cur = gl[n_theta] + gl[n_theta + 1] + gl[n_theta + 2];
    if (n_iter_prev == max_iter) dn = 1;
    else if ((double)(fe_prev + 1) >= max_eval) dn = 3;
    else if (gabs <= tol_fun) dn = 4;
    else if (sabs <= tol_x) dn = 5;
    else if (fabs(f_cur - f_old_prev) < tol_x) dn = 6;
  }
  for (int r = wave; r < M1; r += 4) {
    if (lane < M1) {
      sSY[r * LD + lane] = SY[r * M1 + lane];
      sYY[r * LD + lane] = YY[r * M1 + lane];
    }
  }
  __syncthreads();
  if (do_post && tid == 0) {
    st->f = f_cur; st->func_eval = fe_prev + 1;
    if (dn) { st->done = dn; ex->apply = 0; ex->will_eval = 0; }
    else {                                         // :217-224
      log_iters[n_logged] = n_iter_prev; log_losses[n_logged] = f_cur; st->n_logged = n_logged + 1;
      if (n_iter_prev == max_iter - 1) st->final_loss = f_cur;
    }
  }
  if (dn) return;

  const int n_iter = n_iter_prev + 1;
  const bool first = (n_iter == 1);
  int c = head + len; if (c >= M1) c -= M1;
  const bool accept = !first && ys > 1e-10;        // custom_lbfgs.py:102-114
  if (accept) {
    if (wave == 0 && lane < M1) {
      const double sya = (lane == c) ? ys : dots[M1 + lane];       // s_c . y_lane
      const double say = (lane == c) ? ys : dots[lane];            // s_lane . y_c
      const double yya = (lane == c) ? yy : dots[2 * M1 + lane];
      sSY[c * LD + lane] = sya; SY[c * M1 + lane] = sya;
      sSY[lane * LD + c] = say; SY[lane * M1 + c] = say;
      sYY[c * LD + lane] = yya; YY[c * M1 + lane] = yya;
      sYY[lane * LD + c] = yya; YY[lane * M1 + c] = yya;
    }
    if (tid == 0) ro[c] = 1.0 / ys;
    Hdiag = ys / yy;
    if (len == m) { head += 1; if (head >= M1) head -= M1; } else len += 1;
  }
  __syncthreads();
  if (wave != 0) return;

  int my_pos = lane - head; if (my_pos < 0) my_pos += M1;
  const bool my_in = lane < M1 && my_pos < len;
  const double my_sg = my_in ? dots[3 * M1 + lane] : 0.0;
  const double my_yg = my_in ? dots[4 * M1 + lane] : 0.0;
  double my_ro = 0.0;
  if (my_in) my_ro = (accept && lane == c) ? 1.0 / ys : ro[lane];
  const int row = (my_in ? lane : 0) * LD;

  // backward loop: al_i = ro_i * s_i.q_i, q_i = -g - sum_{j>i} al_j y_j.  Lane a carries
  // acc = sum_{j done} al_j (s_a.y_j); yq0 accumulates y_a.q_0 on the way.
  double acc = 0.0, yq0 = -my_yg, al = 0.0;
  for (int i = len - 1; i >= 0; --i) {
    int si = head + i; if (si >= M1) si -= M1;
    const double al_i = read_lane(my_ro * (-my_sg - acc), si);
    if (lane == si) al = al_i;
    acc += al_i * sSY[row + si];
    yq0 -= al_i * sYY[row + si];
  }
  // forward loop: be_i = ro_i * y_i.(Hdiag q_0 + sum_{j<i} cs_j s_j), cs_i = al_i - be_i.
  // Lane a carries acc2 = sum_{j done} cs_j (s_j.y_a).
  double acc2 = 0.0, cs = 0.0;
  for (int i = 0; i < len; ++i) {
    int si = head + i; if (si >= M1) si -= M1;
    const double c_i = read_lane(al - my_ro * (Hdiag * yq0 + acc2), si);
    if (lane == si) cs = c_i;
    acc2 += c_i * sSY[si * LD + (my_in ? lane : 0)];
  }
  const double cy = -Hdiag * al;
  const double cg = -Hdiag;
  const double gtd = cg * gg + wave_sum(my_in ? (cy * my_yg + cs * my_sg) : 0.0);
  if (lane < M1) { cs_out[lane] = my_in ? cs : 0.0; cy_out[lane] = my_in ? cy : 0.0; }
  if (lane == 0) {
    st->n_iter = n_iter; st->hist_len = len; st->hist_head = head; st->Hdiag = Hdiag;
    st->f_old = f_cur;
    ex->cg = cg; ex->gtd = gtd;
    if (gtd > -tol_x) {                          // custom_lbfgs.py:154-156
      st->done = 2; ex->apply = 0; ex->will_eval = 0;
    } else {
      double t;
      if (first) { const double inv = 1.0 / gabs; t = inv < 1.0 ? inv : 1.0; } else t = lr;
      st->t = t;
      ex->apply = 1;
      ex->will_eval = (n_iter != max_iter) ? 1 : 0;
      if (n_iter == max_iter) st->done = 1;      // :192
    }
  }
}

// d = cg g + sum_j (cy_j y_j + cs_j s_j); g_old = g; x += t d.  Block = 64 elements x 4
// history-quarters, combined through LDS in a fixed order.
template <typename real>
__global__ __launch_bounds__(256) void k_lbc_apply(
    int n, int M1, const LbfgsState* __restrict__ st, const LbcExtra* __restrict__ ex,
    const double* __restrict__ g, const double* __restrict__ Sh, const double* __restrict__ Yh,
    const double* __restrict__ cs, const double* __restrict__ cy, double* __restrict__ d,
    double* __restrict__ g_old, double* __restrict__ x, double* __restrict__ theta,
    real* __restrict__ theta_r, NetDesc nd, float* __restrict__ img) {
  __shared__ double sh[4][64];
  if (!ex->apply) return;
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  const int head = st->hist_head, len = st->hist_len;
  double p0 = 0.0, p1 = 0.0;
  if (i < n) {
    int p = q;
    for (; p + 4 < len; p += 8) {
      int s0 = head + p; if (s0 >= M1) s0 -= M1;
      int s1 = head + p + 4; if (s1 >= M1) s1 -= M1;
      p0 += cy[s0] * Yh[(size_t)s0 * n + i] + cs[s0] * Sh[(size_t)s0 * n + i];
      p1 += cy[s1] * Yh[(size_t)s1 * n + i] + cs[s1] * Sh[(size_t)s1 * n + i];
    }
    for (; p < len; p += 4) {
      int s0 = head + p; if (s0 >= M1) s0 -= M1;
      p0 += cy[s0] * Yh[(size_t)s0 * n + i] + cs[s0] * Sh[(size_t)s0 * n + i];
    }
  }
  sh[q][cl] = p0 + p1;
  __syncthreads();
  if (q != 0 || i >= n) return;
  const double gi = g[i];
  const double di = ex->cg * gi + ((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]));
  d[i] = di;
  g_old[i] = gi;
  const double xi = x[i] + st->t * di;
  x[i] = xi;
  if (ex->will_eval) { theta[i] = xi; theta_r[i] = (real)xi; pack_store(nd, img, i, (float)xi); }
}

}  // namespace pinn
