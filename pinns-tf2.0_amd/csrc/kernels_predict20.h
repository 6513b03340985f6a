// kernels_predict20.h -- forward-only sweeps for width-20 tanh MLPs on the matrix instructions, any depth, one
// output: what `self.model(X_star)` (utils/neuralnetwork.py:151-153, 1d-burgers/inf_cont_burgers.py:95-98) and
// f_model at caller-supplied points (1d-burgers/ide_cont_burgers.py:169-172) cost once training is done, plus the
// device-side relative L2 error of 1d-burgers/inf_cont_burgers.py:114-116 / utils/logger.py:56-60.
//
// Until round 3 these ran on k_forward (one lane per point, scalar-cache weights, HBM stash of every layer): 225 us
// (f32) / 286 us (f64) for the 25 600-point grid.  Here nothing is stashed (no reverse sweep follows) and the layer
// GEMVs use the two mappings of the training kernels:
//   float64  k_fwd20d  v_mfma_f64_4x4x4, lane = (feature slot, point), 16 points per wave, the result registers of
//                      a layer are the B operands of the next (kernels_fused20d.h): no exchange, no barrier
//   float32  k_fwd20f  v_mfma_f32_4x4x1_16B, lane = point, A = period-4 weight pattern, the four D registers are
//                      four output features of the same point (kernels_fused20m.h), one wave per workgroup
// NCH = 1 carries only the value channel (predict, error metric: 25 / 100 matrix instructions per layer and wave),
// NCH = 4 the Taylor channels (u, u_x, u_t, u_xx) for the residual.  Depth is a run-time loop: no register stash.
#pragma once
#include "kernels_fused20d.h"
#include "kernels_fused20m.h"

namespace pinn {

inline bool predict20_ok(const NetDesc& nd) { return nd.width == FW && nd.n_out == 1 && nd.n_hidden >= 1; }
inline size_t predict20_lds_bytes(const NetDesc& nd, size_t rs) { return ((size_t)nd.n_net * rs + 15) / 16 * 16; }

// ---- float64: 4 waves x 16 points per workgroup ---------------------------------------------------------------
// out4 (NCH == 4): [n_pad] vec4 (u, u_x, u_t, u_xx);  out1 (NCH == 1): [n_pad] double u
template <int NCH>
__global__ __launch_bounds__(256) void k_fwd20d(NetDesc nd, const double* __restrict__ th,
                                                const double* __restrict__ xs, const double* __restrict__ ts,
                                                int n_tiles, double lbx, double lbt, double sx, double st,
                                                vec4<double>* __restrict__ out4, double* __restrict__ out1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* const wl = reinterpret_cast<double*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, s = lane >> 4, i4 = lane & 3;
  const int pf = s * FW + i4;                      // forward pattern W[4m + s][4n + i4]
  const int H = nd.n_hidden;
  int tile = blockIdx.x;
  double x = 0.0, t = 0.0;
  if (tile < n_tiles) { x = xs[tile * 64 + wave * 16 + q]; t = ts[tile * 64 + wave * 16 + q]; }
  for (int i = tid; i < nd.n_net; i += 256) wl[i] = th[i];
  __syncthreads();
  for (; tile < n_tiles; tile += gridDim.x) {
    const int pt = tile * 64 + wave * 16 + q;
    const double hx = __builtin_fma(sx, x - lbx, -1.0), ht = __builtin_fma(st, t - lbt, -1.0);
    {
      const int nt = tile + gridDim.x;
      if (nt < n_tiles) { x = xs[nt * 64 + wave * 16 + q]; t = ts[nt * 64 + wave * 16 + q]; }
    }
    double in[NCH][5];
#pragma unroll
    for (int n = 0; n < 5; ++n) {                  // dense 0
      const int f = 4 * n + s;
      const double w0x = wl[nd.off_w[0] + f], w0t = wl[nd.off_w[0] + FW + f], b0 = wl[nd.off_b[0] + f];
      const double a = tanh_d(__builtin_fma(hx, w0x, __builtin_fma(ht, w0t, b0)));
      in[0][n] = a;
      if constexpr (NCH == 4) {
        double h;
        channels_d(a, sx * w0x, st * w0t, 0.0, h, in[1][n], in[2][n], in[3][n]);
      }
    }
    for (int d = 1; d < H; ++d) {
      const double* __restrict__ wd = wl + nd.off_w[d] + pf;
      const double* __restrict__ bd = wl + nd.off_b[d] + s;
      double acc[NCH][5];
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        acc[0][n] = bd[4 * n];
#pragma unroll
        for (int c = 1; c < NCH; ++c) acc[c][n] = 0.0;
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          const double A = wd[80 * m + 4 * n];
#pragma unroll
          for (int c = 0; c < NCH; ++c) acc[c][n] = mfma444(A, in[c][m], acc[c][n]);
        }
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const double a = tanh_d(acc[0][n]);
        in[0][n] = a;
        if constexpr (NCH == 4) {
          double h;
          channels_d(a, acc[1][n], acc[2][n], acc[3][n], h, in[1][n], in[2][n], in[3][n]);
        }
      }
    }
    double o[NCH];
    o[0] = wl[nd.off_b[H]];
#pragma unroll
    for (int c = 1; c < NCH; ++c) o[c] = 0.0;
#pragma unroll
    for (int m = 0; m < 5; ++m) {                  // linear output layer: every slot lane of a point gets the result
      const double A = wl[nd.off_w[H] + 4 * m + s];
#pragma unroll
      for (int c = 0; c < NCH; ++c) o[c] = mfma444(A, in[c][m], o[c]);
    }
    if (s == 0) {
      if constexpr (NCH == 4) out4[pt] = vec4<double>{o[0], o[1], o[2], o[3]};
      else out1[pt] = o[0];
    }
  }
}

// ---- float32: one wave = 64 points per workgroup; hidden matrices staged TRANSPOSED so that one ds_read_b128 holds
// the A patterns of four k:  wl[off_w[d] + j * 20 + k] = W_d[k][j]  (d >= 1; dense 0, biases and the output layer keep
// the flat layout)
template <int NCH>
__global__ __launch_bounds__(64) void k_fwd20f(NetDesc nd, const float* __restrict__ th,
                                               const float* __restrict__ xs, const float* __restrict__ ts,
                                               int n_tiles, float lbx, float lbt, float sx, float st,
                                               vec4<float>* __restrict__ out4, double* __restrict__ out1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int lane = threadIdx.x;
  const int H = nd.n_hidden;
  int tile = blockIdx.x;
  float x = 0.0f, t = 0.0f;
  if (tile < n_tiles) { x = xs[tile * 64 + lane]; t = ts[tile * 64 + lane]; }
  {
    const int lo = nd.off_w[1], per = FW * FW + FW, hi = H >= 2 ? nd.off_w[H] : lo;
    for (int i = lane; i < nd.n_net; i += 64) {
      int dst = i;
      if (i >= lo && i < hi) {
        const int d1 = (i - lo) / per, rem = (i - lo) - d1 * per;
        if (rem < FW * FW) { const int k = rem / FW, j = rem - k * FW; dst = lo + d1 * per + j * FW + k; }
      }
      wl[dst] = th[i];
    }
  }
  __syncthreads();
  for (; tile < n_tiles; tile += gridDim.x) {
    const int pt = tile * 64 + lane;
    const float hx = fmaf(sx, x - lbx, -1.0f), ht = fmaf(st, t - lbt, -1.0f);
    {
      const int nt = tile + gridDim.x;
      if (nt < n_tiles) { x = xs[nt * 64 + lane]; t = ts[nt * 64 + lane]; }
    }
    float in[NCH][FW];
#pragma unroll
    for (int j = 0; j < FW; ++j) {                 // dense 0 (wave-uniform weights)
      const float w0x = wl[nd.off_w[0] + j], w0t = wl[nd.off_w[0] + FW + j], b0 = wl[nd.off_b[0] + j];
      const v4f sv{tanh_r5(fmaf(hx, w0x, fmaf(ht, w0t, b0))), sx * w0x, st * w0t, 0.0f};
      const v4f ch = channels4(sv);
#pragma unroll
      for (int c = 0; c < NCH; ++c) in[c][j] = ch[c];
    }
    for (int d = 1; d < H; ++d) {
      const float* __restrict__ wt = wl + nd.off_w[d] + (lane & 3) * FW;   // rows 4g + lane%4 of W_d^T
      const float* __restrict__ bd = wl + nd.off_b[d];
      acc4 acc[NCH][5];
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        acc[0][g] = *reinterpret_cast<const v4f*>(bd + 4 * g);
#pragma unroll
        for (int c = 1; c < NCH; ++c) acc[c][g] = acc4{0, 0, 0, 0};
      }
#pragma unroll
      for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          const v4f a4 = *reinterpret_cast<const v4f*>(wt + g * 4 * FW + 4 * m);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
              acc[c][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[kk], in[c][4 * m + kk], acc[c][g], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v4f sv{tanh_r5(acc[0][g][i]), 0.0f, 0.0f, 0.0f};
          if constexpr (NCH == 4) { sv.y = acc[1][g][i]; sv.z = acc[2][g][i]; sv.w = acc[3][g][i]; }
          const v4f ch = channels4(sv);
#pragma unroll
          for (int c = 0; c < NCH; ++c) in[c][4 * g + i] = ch[c];
        }
      }
    }
    float o[NCH];
    o[0] = wl[nd.off_b[H]];
#pragma unroll
    for (int c = 1; c < NCH; ++c) o[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < FW; ++k) {
      const float w = wl[nd.off_w[H] + k];
#pragma unroll
      for (int c = 0; c < NCH; ++c) o[c] = fmaf(in[c][k], w, o[c]);
    }
    if constexpr (NCH == 4) out4[pt] = vec4<float>{o[0], o[1], o[2], o[3]};
    else out1[pt] = (double)o[0];
  }
}

// value channel of the Taylor outputs O[o * n_pad + pt] -> compact [n][n_out] float64 (the generic forward sweeps)
template <typename real>
__global__ void k_pick_values(const vec4<real>* __restrict__ O, int n, int n_pad, int n_out, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int o = 0; o < n_out; ++o) out[(size_t)i * n_out + o] = (double)O[(size_t)o * n_pad + i].x;
}

// ---- relative L2 error  ||ref - pred||_2 / ||ref||_2  (inf_cont_burgers.py:114-116, logger.py:56-60), two launches,
// fixed summation order: block b sums elements [b * ERR_SPAN, (b+1) * ERR_SPAN) (thread-strided, DPP wave sums, waves
// in index order), the second kernel adds the block partials in index order.
//   kind 0: pred, ref [n][n_out], element-wise (Frobenius);  kind 1: modulus sqrt(sum_o pred_o^2) against ref [n]
//   (1dcomplex-schrodinger/inf_cont_schrodinger.py:155-158)
constexpr int ERR_THREADS = 256, ERR_SPAN = 2048;
__global__ __launch_bounds__(ERR_THREADS) void k_err_partial(const double* __restrict__ pred, const double* __restrict__ ref,
                                                             long long n_elem, int n_out, int kind,
                                                             double* __restrict__ partial) {
  __shared__ double sh[2][ERR_THREADS / 64];
  double num = 0.0, den = 0.0;
  const long long lo = (long long)blockIdx.x * ERR_SPAN;
  for (long long i = lo + threadIdx.x; i < lo + ERR_SPAN && i < n_elem; i += ERR_THREADS) {
    double p;
    if (kind == 1) {
      double h2 = 0.0;
      for (int o = 0; o < n_out; ++o) { const double v = pred[i * n_out + o]; h2 = __builtin_fma(v, v, h2); }
      p = sqrt(h2);
    } else {
      p = pred[i];
    }
    const double r = ref[i], e = r - p;
    num = __builtin_fma(e, e, num); den = __builtin_fma(r, r, den);
  }
  const double wn = wave_sum(num), wd = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = wn; sh[1][threadIdx.x >> 6] = wd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < ERR_THREADS / 64; ++w) { a += sh[0][w]; b += sh[1][w]; }
    partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b;
  }
}
// result[0] = sqrt(num / den), [1] = num, [2] = den
__global__ void k_err_final(const double* __restrict__ partial, int n_blocks, double* __restrict__ result) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < n_blocks; ++i) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  result[0] = sqrt(a) / sqrt(b); result[1] = a; result[2] = b;
}

}  // namespace pinn
