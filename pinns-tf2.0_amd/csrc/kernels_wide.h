// kernels_wide.h -- float32 forward / reverse sweeps for WIDE tanh MLPs (hidden width W a multiple
// of 4, 64 < W <= 128: the Schrodinger net [2,100,100,100,100,2] of
// 1dcomplex-schrodinger/inf_cont_schrodinger.py:23-41), every contraction on v_mfma_f32_16x16x4_f32.
//
// At width 100 a layer is a real GEMM ([points x 4 channels] x [100 x 100]) and the 64-point tile of
// the width-20 kernels no longer fits LDS (100 features x 64 points x 4 channels x 4 B = 102 KB per
// tile, two are needed).  Mapping here:
//   workgroup = 4 waves = one group of 16 points at a time (persistent over groups);
//   MFMA "C layout": lane = (n = lane & 15: point, g = lane >> 4), VGPR r of output tile a holds
//   feature j = 16a + 4g + r  ->  all four Taylor channels of a (point, feature) sit in ONE lane,
//   so tanh and its derivative chain are lane-local;
//   wave w owns output tiles a in {w, w+4} (7 tiles of 16 cover 100 features, 89 % dense);
//   layer input  B[k][n]  = exchange tile T[k][n] (float4 = 4 channels; one ds_read_b128 serves the
//                           four channel MFMAs of a k-step), written by all waves, barrier per layer;
//   weights      A[j][k]  = the layer's matrix staged in LDS per layer ([k][112] floats, zero padded,
//                           bias as row W), 45 KB: one conflict-free ds_read_b32 per (k-step, tile);
//   stash (a, z_x, z_t, z_xx) goes to HBM in the generic kernels' layout
//   S[(layer*W + feature)*s_pad + point] (6.4 KB per point at 4x100: 130 MB per evaluation of the
//   reference configuration, ~50 us at HBM rate against >= 93 us of FP32 work).
// Forward and reverse are two launches (the periodic-boundary seeds of the Schrodinger loss couple
// pairs of points through the network outputs, inf_cont_schrodinger.py:107-129); they use the same
// S / O buffers as k_forward / k_backward (kernels_generic.h), so either half can be swapped for its
// generic counterpart -- which is how the tests pin them.
//
// Math: SURVEY.md Appendix A.
#pragma once
#include "kernels_fused20m.h"

namespace pinn {

template <int W>
struct WideCfg {
  static constexpr int NT = (W + 15) / 16;     // output tiles of 16 features
  static constexpr int WP = NT * 16;           // padded width
  static constexpr int KS = W / 4;             // k-steps of a W-long contraction
  static constexpr int TP = 17;                // exchange-tile row pitch in float4 (16 points + 1 pad)
  static constexpr int TILE = WP * TP;         // float4 per exchange tile
  static constexpr int IMG_RAW = (W + 1) * WP;  // floats per matrix (W rows + bias row)
  static constexpr int NCH = (IMG_RAW + 255) / 256;   // 1 KiB LDS-DMA pieces per matrix
  static constexpr int IMG = NCH * 256;         // floats per staged matrix, padded to whole pieces
  static_assert(W % 4 == 0 && W > 64 && W <= 128, "wide kernels serve 64 < W <= 128, W % 4 == 0");
};

// global weight image: per hidden dense layer d = 1..H-1 two matrices of IMG floats
//   F_d[k][WP] = W_d[k][j]  (+ row W = b_d)     forward A operands
//   R_d[j][WP] = W_d[k][j]                       reverse A operands (W_d^T)
template <int W>
inline size_t wide_image_floats(int n_hidden) { return (size_t)(n_hidden - 1) * 2 * WideCfg<W>::IMG; }
template <int W>
inline size_t wide_lds_bytes() {      // two weight buffers, two exchange tiles, output-layer partials [4 waves][16][8]
  return (size_t)2 * WideCfg<W>::IMG * 4 + (size_t)2 * WideCfg<W>::TILE * 16 + 4 * 16 * 8 * 4;
}

__device__ __forceinline__ void pack_store_wide(const NetDesc& nd, float* __restrict__ img, int i, float v) {
  const int W = nd.width, WP = (W + 15) / 16 * 16, IMG = ((W + 1) * WP + 255) / 256 * 256;
  const int lo = nd.off_w[1], hi = nd.off_w[nd.n_hidden];
  if (i < lo || i >= hi) return;
  const int per = W * W + W;
  const int d1 = (i - lo) / per, rem = (i - lo) - d1 * per;
  float* __restrict__ F = img + (size_t)d1 * 2 * IMG;
  float* __restrict__ Rm = F + IMG;
  if (rem < W * W) {
    const int k = rem / W, j = rem - k * W;
    F[k * WP + j] = v;
    Rm[j * WP + k] = v;
  } else {
    F[W * WP + rem - W * W] = v;
  }
}

// Called by every kernel that writes a weight: mirrors flat parameter i into whichever packed
// image the engine keeps for the active loss+grad kernel.
__device__ __forceinline__ void pack_store_any(const NetDesc& nd, float* __restrict__ img, int i, float v) {
  if (!img) return;
  if (nd.img_kind == 1) pack_store_m(nd, img, i, v);
  else if (nd.img_kind == 2) pack_store_wide(nd, img, i, v);
}

// Asynchronous global -> LDS copy of one matrix (global_load_lds_dwordx4: each wave-instruction
// moves 1 KiB, destination = wave-uniform base + lane * 16, no registers involved).  The copy of
// the NEXT layer's matrix is issued right after a layer's opening barrier and lands in the other
// LDS buffer while the MFMAs of the current layer run; the issuing wave drains it with
// s_waitcnt vmcnt(0) just before the next opening barrier.
template <int W>
__device__ __forceinline__ void wide_dma(float* __restrict__ wl_dst, const float* __restrict__ src,
                                         const int wave, const int lane) {
  using C = WideCfg<W>;
  for (int c = wave; c < C::NCH; c += 4)
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(src + c * 256 + lane * 4),
        (__attribute__((address_space(3))) void*)(wl_dst + c * 256), 16, 0, 0);
}
// The DMA is drained (s_waitcnt vmcnt(0)) right after the MFMA block it ran under -- by then it is
// long done and the stash stores of the previous layer are too, so the wait is free -- and NOT at
// the barriers, which would also wait for the acknowledgement of the stash stores just issued.
__device__ __forceinline__ void wide_dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// acc[t][c] += sum_k A[k][16 a_t + m] * Tin_c[k][n] for this wave's tiles a_t = wave + 4t, t < NTW.
// Straight-line over the k-steps (no branch inside: the accumulators must stay in the MFMA
// register file), operands of step s+1 fetched before the eight MFMAs of step s are issued.
template <int W, int NTW>
__device__ __forceinline__ void wide_gemm_n(acc4 (&acc)[2][4], const float* __restrict__ wl,
                                            const v4f* __restrict__ Tin, const int wave, const int lane) {
  using C = WideCfg<W>;
  const int n = lane & 15, g = lane >> 4;
  const float* __restrict__ a0 = wl + g * C::WP + 16 * wave + n;
  const v4f* __restrict__ b = Tin + g * C::TP + n;
  v4f Bn = b[0];
  float A0n = a0[0], A1n = NTW > 1 ? a0[64] : 0.0f;
#pragma unroll
  for (int s = 0; s < C::KS; ++s) {
    const v4f B = Bn;
    const float A0 = A0n, A1 = A1n;
    if (s + 1 < C::KS) {
      Bn = b[4 * (s + 1) * C::TP];
      A0n = a0[4 * (s + 1) * C::WP];
      if (NTW > 1) A1n = a0[4 * (s + 1) * C::WP + 64];
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, B.x, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, B.y, acc[0][1], 0, 0, 0);
    acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, B.z, acc[0][2], 0, 0, 0);
    acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, B.w, acc[0][3], 0, 0, 0);
    if (NTW > 1) {
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, B.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, B.y, acc[1][1], 0, 0, 0);
      acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, B.z, acc[1][2], 0, 0, 0);
      acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, B.w, acc[1][3], 0, 0, 0);
    }
  }
}
template <int W>
__device__ __forceinline__ void wide_gemm(acc4 (&acc)[2][4], const float* __restrict__ wl,
                                          const v4f* __restrict__ Tin, const int wave, const int lane) {
  if (wave + 4 < WideCfg<W>::NT) wide_gemm_n<W, 2>(acc, wl, Tin, wave, lane);   // wave-uniform
  else wide_gemm_n<W, 1>(acc, wl, Tin, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Forward sweep over points [base, base + 16*n_groups): fills S (stash) and O (outputs) exactly as
// k_forward does.
// ---------------------------------------------------------------------------------------------
template <int W, int NO>
__global__ __launch_bounds__(256) void k_wide_fwd(NetDesc nd, const float* __restrict__ th,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ xs,
                                                  const float* __restrict__ ts, int base, int n_pad,
                                                  int s_pad, int n_groups, float lbx, float lbt,
                                                  float sx, float st, vec4<float>* __restrict__ S,
                                                  vec4<float>* __restrict__ O) {
  using C = WideCfg<W>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wbuf = reinterpret_cast<float*>(lds_raw);           // two weight buffers of IMG floats
  v4f* const T0 = reinterpret_cast<v4f*>(wbuf + 2 * C::IMG);
  v4f* const T1 = T0 + C::TILE;
  float* const OP = reinterpret_cast<float*>(T1 + C::TILE);        // output-layer partials [4 waves][16 points][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int H = nd.n_hidden;
  v4f* const Sv = reinterpret_cast<v4f*>(S);
  v4f* const Ov = reinterpret_cast<v4f*>(O);
  static_assert(NO <= 2, "output-layer partials are laid out for at most two outputs");
  int wcur = 0;                                                    // buffer holding the matrix about to be used
  wide_dma<W>(wbuf, img, wave, lane);                              // F_1
  wide_dma_drain();

  // layer-0 parameters of this lane's features (tile t, register r): j = 16 (wave + 4t) + 4g + r
  float w0x[2][4], w0t[2][4], b0[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * (wave + 4 * t) + 4 * g + r;
      const bool ok = j < W;
      w0x[t][r] = ok ? th[nd.off_w[0] + j] : 0.0f;
      w0t[t][r] = ok ? th[nd.off_w[0] + W + j] : 0.0f;
      b0[t][r] = ok ? th[nd.off_b[0] + j] : 0.0f;
    }
  // output layer, K split over the four waves (wave w takes k-steps w, w+4, ...): A[m][k] = WL[k][m] for m < NO
  constexpr int KSW = (C::KS + 3) / 4;
  float aL[KSW];
#pragma unroll
  for (int i = 0; i < KSW; ++i) {
    const int s = wave + 4 * i;
    aL[i] = (s < C::KS && n < NO) ? th[nd.off_w[H] + (4 * s + g) * NO + n] : 0.0f;
  }
  const float bL0 = th[nd.off_b[H]], bL1 = NO > 1 ? th[nd.off_b[H] + 1] : 0.0f;

  float x_next = 0.0f, t_next = 0.0f;
  if (blockIdx.x < n_groups) { x_next = xs[base + blockIdx.x * 16 + n]; t_next = ts[base + blockIdx.x * 16 + n]; }
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int lp = grp * 16 + n, pt = base + lp;
    const float x = x_next, t = t_next;
    if (grp + (int)gridDim.x < n_groups) {       // the next group's inputs travel under this group's layers
      x_next = xs[pt + 16 * gridDim.x];
      t_next = ts[pt + 16 * gridDim.x];
    }
    const float hx = fmaf(sx, x - lbx, -1.0f), ht = fmaf(st, t - lbt, -1.0f);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int a = wave + 4 * tt;
      if (a < C::NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * a + 4 * g + r;
          const float z = fmaf(hx, w0x[tt][r], fmaf(ht, w0t[tt][r], b0[tt][r]));
          const v4f s{j < W ? tanh_r5(z) : 0.0f, sx * w0x[tt][r], st * w0t[tt][r], 0.0f};
          if (j < W) Sv[(size_t)j * s_pad + lp] = s;
          T0[j * C::TP + n] = channels4(s);
        }
      }
    }
    v4f* Tin = T0;
    v4f* Tout = T1;
    for (int d = 1; d < H; ++d) {
      lds_barrier();              // Tin published, F_d landed (drained below / in the prologue),
                                  // nobody still reads the other weight buffer
      const float* __restrict__ wl = wbuf + wcur * C::IMG;
      {                           // next matrix on its way: F_{d+1}, or F_1 for the next group
        const int dn = d + 1 < H ? d : 0;
        wide_dma<W>(wbuf + (wcur ^ 1) * C::IMG, img + (size_t)dn * 2 * C::IMG, wave, lane);
      }
      acc4 acc[2][4];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int a = (wave + 4 * tt < C::NT) ? wave + 4 * tt : wave;
        acc[tt][0] = *reinterpret_cast<const v4f*>(wl + W * C::WP + 16 * a + 4 * g);   // bias b_d[j]
        acc[tt][1] = acc[tt][2] = acc[tt][3] = acc4{0, 0, 0, 0};
      }
      wide_gemm<W>(acc, wl, Tin, wave, lane);
      wide_dma_drain();
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int a = wave + 4 * tt;
        if (a < C::NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * a + 4 * g + r;
            const v4f s{tanh_r5(acc[tt][0][r]), acc[tt][1][r], acc[tt][2][r], acc[tt][3][r]};
            if (j < W) Sv[((size_t)d * W + j) * s_pad + lp] = s;
            Tout[j * C::TP + n] = channels4(s);          // padded features: exact zeros
          }
        }
      }
      wcur ^= 1;
      v4f* tmp = Tin; Tin = Tout; Tout = tmp;
    }
    lds_barrier();                // last hidden layer's tile published
    {                             // linear output layer: every wave its share of the k-steps, rows m < NO of one tile
      acc4 ao[4];
      ao[0] = acc4{(wave == 3 && g == 0) ? bL0 : 0.0f, (wave == 3 && g == 0) ? bL1 : 0.0f, 0, 0};
      ao[1] = ao[2] = ao[3] = acc4{0, 0, 0, 0};
      const v4f* __restrict__ b = Tin + g * C::TP + n;
#pragma unroll
      for (int i = 0; i < KSW; ++i) {
        const int s = wave + 4 * i;
        const v4f B = b[4 * (s < C::KS ? s : 0) * C::TP];       // out-of-range steps carry a zero A operand
        ao[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aL[i], B.x, ao[0], 0, 0, 0);
        ao[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aL[i], B.y, ao[1], 0, 0, 0);
        ao[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aL[i], B.z, ao[2], 0, 0, 0);
        ao[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(aL[i], B.w, ao[3], 0, 0, 0);
      }
      if (g == 0) {               // rows 0..3 of the tile live in the g == 0 lanes: [output][channel] of point n
        float* __restrict__ dst = OP + (wave * 16 + n) * 8;
        *reinterpret_cast<v4f*>(dst) = v4f{ao[0][0], ao[1][0], ao[2][0], ao[3][0]};
        *reinterpret_cast<v4f*>(dst + 4) = v4f{ao[0][1], ao[1][1], ao[2][1], ao[3][1]};
      }
    }
    lds_barrier();                // partials published; every wave is done with the last tile
    if (wave == 3 && lane < 16 * NO) {          // lane = (output o, point n'): sum the four partials in wave order
      const int o = lane >> 4, np = lane & 15;
      v4f tot = *reinterpret_cast<const v4f*>(OP + (0 * 16 + np) * 8 + 4 * o);
#pragma unroll
      for (int w = 1; w < 4; ++w) tot += *reinterpret_cast<const v4f*>(OP + (w * 16 + np) * 8 + 4 * o);
      Ov[(size_t)o * n_pad + base + grp * 16 + np] = tot;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

// sum over the 16 lanes of a DPP row (the 16 points of a group); every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<DPP_QUAD_XOR1>(v);
  v += dpp_mov<DPP_QUAD_XOR2>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Reverse sweep over points [base, base + 16*n_groups), consuming S and O of the forward sweep:
// seeds (point_seeds, kernels_generic.h), adjoints through every layer, all weight gradients.
//   rev GEMM   in_bar[k][n] = sum_j W_d[k][j] z_bar[j][n]: the same wide_gemm on the staged W_d^T;
//   dW_d       49 output tiles (k-tile x j-tile, bias = the constant ones row k = W of the input
//              tile) x 16 MFMAs over the group's 64 (point, channel) rows, dealt round-robin to the
//              four waves; the accumulators (13 tiles x 3 layers = 156 registers) stay resident
//              across groups, and so do the per-lane partial sums of the first / last layer;
//   output     one partial-gradient row per workgroup (same `part` format as the other kernels).
// ---------------------------------------------------------------------------------------------
template <int W, int NO, int PDE, int H>
__global__ __launch_bounds__(256) void k_wide_bwd(NetDesc nd, SetDesc sd, const float* __restrict__ th,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ xs,
                                                  const float* __restrict__ ts,
                                                  const float* __restrict__ tgt, int base, int n_pad,
                                                  int s_pad, int n_groups, float lbx, float lbt,
                                                  float sx, float st, float nu,
                                                  const vec4<float>* __restrict__ S,
                                                  const vec4<float>* __restrict__ O,
                                                  float* __restrict__ part, int R, int accumulate) {
  using C = WideCfg<W>;
  constexpr int NTT = C::NT * C::NT;              // dW tiles per layer
  constexpr int NQ = (NTT + 3) / 4;               // tiles per wave (round-robin)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wbuf = reinterpret_cast<float*>(lds_raw);           // two weight buffers of IMG floats
  v4f* const TI = reinterpret_cast<v4f*>(wbuf + 2 * C::IMG);
  v4f* const TZ = TI + C::TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const v4f* const Sv = reinterpret_cast<const v4f*>(S);
  float* __restrict__ row = part + (size_t)blockIdx.x * R;
  int wcur = 0;
  wide_dma<W>(wbuf, img + (size_t)(H - 2) * 2 * C::IMG + C::IMG, wave, lane);     // W_{H-1}^T
  wide_dma_drain();

  float c1 = 1.0f, c2 = nu;
  if (PDE == 1) { c1 = th[nd.n_net]; c2 = __expf(th[nd.n_net + 1]); }
  // output-layer weights of this lane's features
  float wLo[2][4][NO];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * (wave + 4 * t) + 4 * g + r;
#pragma unroll
      for (int o = 0; o < NO; ++o) wLo[t][r][o] = j < W ? th[nd.off_w[H] + j * NO + o] : 0.0f;
    }

  // accumulators that live across groups
  acc4 dwacc[H - 1][NQ];                          // [d - 1][q]
#pragma unroll
  for (int d = 0; d < H - 1; ++d)
#pragma unroll
    for (int q = 0; q < NQ; ++q) dwacc[d][q] = acc4{0, 0, 0, 0};
  float g0x[2][4], g0t[2][4], g0b[2][4], gWL[2][4][NO];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      g0x[t][r] = g0t[t][r] = g0b[t][r] = agpr_put(0.0f);      // parked in AGPRs between groups
#pragma unroll
      for (int o = 0; o < NO; ++o) gWL[t][r][o] = agpr_put(0.0f);
    }
  float gbL[NO], lsum[3] = {0, 0, 0}, dlsum[2] = {0, 0};
#pragma unroll
  for (int o = 0; o < NO; ++o) gbL[o] = 0.0f;

  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    // the lane index is re-read through an opaque asm once per group: every per-lane address of the body is
    // loop-invariant, hipcc hoisted them out of the group loop and the 512-register wave spilled 54 of them (220 B of
    // scratch per lane, rounds 1-3); recomputed per group they cost a few integer instructions and no scratch
    int lane_o = tid & 63;
    asm volatile("" : "+v"(lane_o));
    const int lane = lane_o, n = lane_o & 15, g = lane_o >> 4;
  auto feat = [&](int t, int r) { return 16 * (wave + 4 * t) + 4 * g + r; };
    auto load_stash = [&](v4f (&dst)[2][4], int d, int lp) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * (wave + 4 * t) + 4 * (lane >> 4) + r;
          dst[t][r] = j < W ? Sv[((size_t)d * W + j) * s_pad + lp] : v4f{0, 0, 0, 0};
        }
    };

    const int lp = grp * 16 + n, pt = base + lp;
    const float x = xs[pt], t_ = ts[pt];
    const float hx = fmaf(sx, x - lbx, -1.0f), ht = fmaf(st, t_ - lbt, -1.0f);
    v4f s_cur[2][4], s_prev[2][4];
    load_stash(s_cur, H - 1, lp);
    load_stash(s_prev, H - 2, lp);

    vec4<float> sbv[2];
    float lt[3], dl[2];
    point_seeds<float, PDE>(sd, pt, n_pad, O, tgt, c1, c2, sbv, lt, dl);
    v4f sb[2];
    sb[0] = v4f{sbv[0].x, sbv[0].y, sbv[0].z, sbv[0].w};
    sb[1] = v4f{sbv[1].x, sbv[1].y, sbv[1].z, sbv[1].w};
    if (g == 0) {
      lsum[0] += lt[0]; lsum[1] += lt[1]; lsum[2] += lt[2];
      dlsum[0] += dl[0]; dlsum[1] += dl[1];
#pragma unroll
      for (int o = 0; o < NO; ++o) gbL[o] += sb[o].x;
    }
    // dense H (linear): z_bar = seeds; adjoint of this lane's layer-(H-1) outputs
    v4f ob[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const v4f in = channels4(s_cur[t][r]);
        v4f a = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          gWL[t][r][o] = agpr_put(agpr_get(gWL[t][r][o]) +
                                  fmaf(in.w, sb[o].w, fmaf(in.z, sb[o].z, fmaf(in.y, sb[o].y, in.x * sb[o].x))));
          a += sb[o] * wLo[t][r][o];
        }
        ob[t][r] = a;
      }

#pragma unroll
    for (int d = H - 1; d >= 1; --d) {
      // phase A: publish z_bar (layer d) and the layer-(d-1) output channels of this lane's features
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int a = wave + 4 * t;
        if (a < C::NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * a + 4 * g + r;
            TZ[j * C::TP + n] = j < W ? preact_adjoint4(s_cur[t][r], ob[t][r]) : v4f{0, 0, 0, 0};
            TI[j * C::TP + n] = j < W ? channels4(s_prev[t][r]) : (j == W ? v4f{1, 0, 0, 0} : v4f{0, 0, 0, 0});
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_cur[t][r] = s_prev[t][r];
      lds_barrier();              // TZ / TI published, W_d^T landed, the other weight buffer is free
      const float* __restrict__ wl = wbuf + wcur * C::IMG;
      {                           // next matrix on its way: W_{d-1}^T, or W_{H-1}^T for the next group
        const int dn = d >= 2 ? d - 1 : H - 1;
        wide_dma<W>(wbuf + (wcur ^ 1) * C::IMG, img + (size_t)(dn - 1) * 2 * C::IMG + C::IMG, wave, lane);
        wcur ^= 1;
      }
      // phase B: adjoint of the layer-(d-1) outputs, own tiles
      acc4 acc[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = acc4{0, 0, 0, 0};
      wide_gemm<W>(acc, wl, TZ, wave, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[t][r] = v4f{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
      // dW_d tiles tau = wave + 4q: rows k = 16 ti + ., columns j = 16 tj + .  No branch around the
      // MFMAs (the accumulators must stay in the matrix register file): a wave whose last slot has
      // no tile (tau >= 49) recomputes tile 48 there and the result is simply not stored.
      auto tile_ptrs = [&](int q, const v4f*& pa, const v4f*& pb) {
        const int tau = min(wave + 4 * q, NTT - 1);
        const int ti = tau / C::NT, tj = tau - ti * C::NT;
        pa = TI + (16 * ti + n) * C::TP + g;
        pb = TZ + (16 * tj + n) * C::TP + g;
      };
      // two tiles at a time, their MFMAs alternating: each accumulator is then touched every 64
      // cycles, above the 40-cycle dependent-issue latency of v_mfma_f32_16x16x4_f32
#pragma unroll
      for (int q = 0; q + 1 < NQ; q += 2) {
        const v4f *pa0, *pb0, *pa1, *pb1;
        tile_ptrs(q, pa0, pb0);
        tile_ptrs(q + 1, pa1, pb1);
        acc4 a0 = dwacc[d - 1][q], a1 = dwacc[d - 1][q + 1];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const v4f A0 = pa0[4 * s4], B0 = pb0[4 * s4], A1 = pa1[4 * s4], B1 = pb1[4 * s4];
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.x, B0.x, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.x, B1.x, a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.y, B0.y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.y, B1.y, a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.z, B0.z, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.z, B1.z, a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0.w, B0.w, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1.w, B1.w, a1, 0, 0, 0);
        }
        dwacc[d - 1][q] = a0; dwacc[d - 1][q + 1] = a1;
        __builtin_amdgcn_sched_barrier(0);        // keep one pair's operands live at a time
      }
      if (NQ & 1) {
        const v4f *pa, *pb;
        tile_ptrs(NQ - 1, pa, pb);
        acc4 a = dwacc[d - 1][NQ - 1];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const v4f A = pa[4 * s4], B = pb[4 * s4];
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(A.x, B.x, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(A.y, B.y, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(A.z, B.z, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w, B.w, a, 0, 0, 0);
        }
        dwacc[d - 1][NQ - 1] = a;
      }
      wide_dma_drain();           // next matrix landed (issued a whole phase B ago)
      if (d >= 2) load_stash(s_prev, d - 2, lp);
      lds_barrier();              // every wave is done with TI, TZ and this layer's weights
    }
    // dense 0: inputs (hx, ht), p0 = (sx, 0), q0 = (0, st); s_cur now holds the layer-0 stash
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const v4f zb = feat(t, r) < W ? preact_adjoint4(s_cur[t][r], ob[t][r]) : v4f{0, 0, 0, 0};
        g0x[t][r] = agpr_put(agpr_get(g0x[t][r]) + fmaf(hx, zb.x, sx * zb.y));
        g0t[t][r] = agpr_put(agpr_get(g0t[t][r]) + fmaf(ht, zb.x, st * zb.z));
        g0b[t][r] = agpr_put(agpr_get(g0b[t][r]) + zb.x);
      }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
  // ---- one partial-gradient row per workgroup
  auto put = [&](int idx, float v) { row[idx] = accumulate ? row[idx] + v : v; };
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * (wave + 4 * t) + 4 * (lane >> 4) + r;
      const float a = row16_sum(agpr_get(g0x[t][r])), b = row16_sum(agpr_get(g0t[t][r])),
                  c = row16_sum(agpr_get(g0b[t][r]));
      float e[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) e[o] = row16_sum(agpr_get(gWL[t][r][o]));
      if (n == 0 && j < W) {
        put(nd.off_w[0] + j, a); put(nd.off_w[0] + W + j, b); put(nd.off_b[0] + j, c);
#pragma unroll
        for (int o = 0; o < NO; ++o) put(nd.off_w[H] + j * NO + o, e[o]);
      }
    }
  if (wave == 0) {
    float v3[3], vb[NO], vd[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) v3[i] = row16_sum(lsum[i]);
#pragma unroll
    for (int o = 0; o < NO; ++o) vb[o] = row16_sum(gbL[o]);
    vd[0] = row16_sum(dlsum[0]); vd[1] = row16_sum(dlsum[1]);
    if (lane == 0) {
      put(nd.n_theta + 0, v3[0]); put(nd.n_theta + 1, v3[1]); put(nd.n_theta + 2, v3[2]);
#pragma unroll
      for (int o = 0; o < NO; ++o) put(nd.off_b[H] + o, vb[o]);
      if (PDE == 1) { put(nd.n_net, vd[0]); put(nd.n_net + 1, vd[1]); }
    }
  }
#pragma unroll
  for (int d = 1; d < H; ++d)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int tau = wave + 4 * q;
      if (tau < NTT) {
        const int ti = tau / C::NT, tj = tau - ti * C::NT;
        const int j = 16 * tj + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * ti + 4 * g + r;
          if (j < W && k < W) put(nd.off_w[d] + k * W + j, dwacc[d - 1][q][r]);
          else if (j < W && k == W) put(nd.off_b[d] + j, dwacc[d - 1][q][r]);
        }
      }
    }
}

}  // namespace pinn
