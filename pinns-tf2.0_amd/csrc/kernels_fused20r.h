// kernels_fused20r.h -- float32 loss+gradient kernel for width-20 tanh MLPs with the Taylor-channel
// stash held in registers (k_fused20r).  Same mathematics and the same workgroup mapping as
// k_fused20 (kernels_fused20.h: 4 waves, lane = point, wave w owns features [5w, 5w+5), layer
// GEMVs fed through LDS exchange tiles, weight gradient on the matrix pipe), re-cut around what
// the s_memtime timeline and the issue-rate microbenchmark (profiles/ubench) showed on gfx950:
// a workgroup of this shape runs one wave per SIMD, and a lone wave issues one instruction per
// ~5.3 cycles whatever it is, so the kernel is *instruction-issue* bound and the lever is the
// per-wave instruction count.
//   * GEMV inner products are v_pk_fma_f32 on channel pairs (h,p) and (q,r); the weight operand is
//     broadcast from one half of a register pair with op_sel, so nothing is moved or splatted:
//     200 packed FMAs per layer and direction instead of 400 scalar ones.
//   * the depth H is a template parameter and both layer loops are fully unrolled: the per-layer
//     stash (a, z_x, z_t, z_xx) of the wave's own 5 features stays in VGPRs (H x 5 x 4 = 160 of
//     the 512 a one-wave-per-SIMD kernel may use).  No HBM stash: the kernel's only global
//     traffic is the 8 B/point of coordinates, the packed weights and one gradient row.
//   * the hidden-layer weights arrive pre-packed per (layer, wave) from a float image that the
//     optimiser kernels keep current (pack_store below), staged with 16-byte copies and kept at
//     the bottom of LDS so every DS access uses an immediate offset; a wave pulls its 100 weights
//     of a layer into registers with 25 broadcast ds_read_b128.
//   * workgroups are persistent over tiles of 64 points: weight-gradient accumulators (MFMA C
//     registers) and the per-lane partial sums of the first/last layer and of the loss live in
//     registers across tiles; one gradient row per workgroup is written at the end, so the
//     reduction that follows sees at most one row per compute unit however large N_f is.
//
// Math: SURVEY.md Appendix A.1-A.3 == nested GradientTapes of inf_cont_burgers.py:65-90 under
// the outer tape of utils/neuralnetwork.py:55-59.
#pragma once
#include "kernels_fused20.h"

namespace pinn {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Packed weight image, one block of WBLK floats per (hidden dense layer d = 1..H-1, wave w):
//   [0, 100)    forward  W_d[k][5w+jj]   at k*5 + jj
//   [100, 105)  bias     b_d[5w+jj]
//   [108, 208)  reverse  W_d[5w+kk][j]   at 108 + j*5 + kk
constexpr int WBLK = 208;
constexpr int WBLK_REV = 108;

// + one block of padding: the second lane-distributed read of the last block runs past its end
inline size_t fused20r_image_floats(int n_hidden) { return (size_t)(n_hidden - 1) * 4 * WBLK + 64; }
inline size_t fused20r_lds_bytes(int n_hidden) {
  return fused20r_image_floats(n_hidden) * 4 + (size_t)4 * FROWS * 65 * 16;
}

// Called by every kernel that writes a weight: mirrors flat parameter i into the packed image.
__device__ __forceinline__ void pack_store(const NetDesc& nd, float* __restrict__ img, int i, float v) {
  if (!img) return;
  const int lo = nd.off_w[1], hi = nd.off_w[nd.n_hidden];
  if (i < lo || i >= hi) return;
  constexpr int PER = FW * FW + FW;
  const int d1 = (i - lo) / PER, rem = (i - lo) - d1 * PER;
  if (rem < FW * FW) {
    const int k = rem / FW, j = rem - k * FW;
    img[(d1 * 4 + j / FF) * WBLK + k * FF + j % FF] = v;
    img[(d1 * 4 + k / FF) * WBLK + WBLK_REV + j * FF + k % FF] = v;
  } else {
    const int j = rem - FW * FW;
    img[(d1 * 4 + j / FF) * WBLK + FW * FF + j % FF] = v;
  }
}

// acc += in * broadcast(low / high half of the scalar pair w)
__device__ __forceinline__ void pkfma_lo(v2f& acc, const v2f in, const v2f w) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(in), "s"(w));
}
__device__ __forceinline__ void pkfma_hi(v2f& acc, const v2f in, const v2f w) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(in), "s"(w));
}
// A wave's 100 weights of one layer and direction sit lane-distributed in two VGPRs (two
// lane-contiguous ds_read_b32 = 512 B through the LDS return path, instead of 25 KB of broadcast
// reads); weight f is pulled into an SGPR with v_readlane and feeds the packed FMAs as a scalar
// operand.  Measured on gfx950: LDS -> VGPR delivery is ~32 B/clk per SIMD, so operand bytes, not
// FMA issue, bound the VGPR-broadcast formulation.
__device__ __forceinline__ float lane_weight(const float wv0, const float wv1, const int f) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f < 64 ? wv0 : wv1), f & 63));
}
// acc[jj] (+)= in[k] * w[k][jj] for the 5x20 block, weights f = k*5 + jj taken two at a time.
// side(slot) is called twice per weight pair (slots 0..99, in program order): the reverse sweep
// drops its matrix-pipe work there, so that every MFMA is followed by >= 3 vector instructions
// (a lone wave issues in order; back-to-back MFMAs would hold it for the full 32-cycle passes).
template <typename IN_AT, typename SIDE>
__device__ __forceinline__ void gemv_5x20(v2f (&alo)[FF], v2f (&ahi)[FF], const float wv0, const float wv1,
                                          IN_AT in_at, SIDE side) {
#pragma unroll
  for (int p = 0; p < (FW * FF) / 2; ++p) {
    const v2f wp{lane_weight(wv0, wv1, 2 * p), lane_weight(wv0, wv1, 2 * p + 1)};
    {
      const int f = 2 * p, k = f / FF, jj = f - k * FF;
      const v4f in = in_at(k);
      pkfma_lo(alo[jj], __builtin_shufflevector(in, in, 0, 1), wp);
      pkfma_lo(ahi[jj], __builtin_shufflevector(in, in, 2, 3), wp);
    }
    side(2 * p);
    {
      const int f = 2 * p + 1, k = f / FF, jj = f - k * FF;
      const v4f in = in_at(k);
      pkfma_hi(alo[jj], __builtin_shufflevector(in, in, 0, 1), wp);
      pkfma_hi(ahi[jj], __builtin_shufflevector(in, in, 2, 3), wp);
    }
    side(2 * p + 1);
  }
}

// The per-layer stash lives in the accumulation half of the unified register file: parking it
// there explicitly (instead of letting the allocator spill to AGPRs) keeps it out of the VGPR
// pressure the scheduler reasons about, so LDS reads can be hoisted well ahead of their use.
__device__ __forceinline__ float agpr_put(const float x) {
  float a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x));
  return a;
}
__device__ __forceinline__ float agpr_get(const float a) {
  float x;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
  return x;
}
__device__ __forceinline__ v4f agpr_put4(const v4f s) {
  return v4f{agpr_put(s.x), agpr_put(s.y), agpr_put(s.z), agpr_put(s.w)};
}
__device__ __forceinline__ v4f agpr_get4(const v4f a) {
  return v4f{agpr_get(a.x), agpr_get(a.y), agpr_get(a.z), agpr_get(a.w)};
}

// tanh(x) = 1 - 2 / (1 + e^{2x}); absolute error ~1 ulp of 1.0 (cf. tanh_bf)
__device__ __forceinline__ float tanh_r5(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// layer-output channels (h, p, q, r) from a stash entry s = (a, zp, zq, zr)
__device__ __forceinline__ v4f channels4(const v4f s) {
  const float a = s.x, d1 = fmaf(-a, a, 1.0f);
  const float t = (-2.0f * a) * s.y;
  return v4f{a, d1 * s.y, d1 * s.z, d1 * fmaf(t, s.y, s.w)};
}

// adjoint of the pre-activation channels (A.3)
__device__ __forceinline__ v4f preact_adjoint4(const v4f s, const v4f ob) {
  const float a = s.x, a2 = a * a, d1 = 1.0f - a2;
  const float d2 = (-2.0f * a) * d1;
  const float d3 = (-2.0f * d1) * fmaf(-3.0f, a2, 1.0f);
  const float zpw = s.y * ob.w;
  const float dot = fmaf(s.w, ob.w, fmaf(s.z, ob.z, s.y * ob.y));
  v4f zb;
  zb.x = fmaf(d3 * s.y, zpw, fmaf(d2, dot, d1 * ob.x));
  zb.y = fmaf(d2 + d2, zpw, d1 * ob.y);
  zb.z = d1 * ob.z;
  zb.w = d1 * ob.w;
  return zb;
}

template <int PDE, int H>
__global__ __launch_bounds__(256) void k_fused20r(NetDesc nd, SetDesc sd,
                                                  const float* __restrict__ th,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ xs,
                                                  const float* __restrict__ ts,
                                                  const float* __restrict__ tgt, float lbx, float lbt,
                                                  float sx, float st, float nu,
                                                  float* __restrict__ part, int R, int n_tiles,
                                                  long long* __restrict__ stamps) {
  constexpr int RS4 = 65;
  constexpr int BUFV = FROWS * RS4;                 // v4f elements per exchange buffer
  constexpr int NW = (H - 1) * 4 * WBLK + 64;       // floats of packed weights (+ read-past padding)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  v4f* const xb = reinterpret_cast<v4f*>(wl + NW);

  STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = wave * FF;
  float* __restrict__ row = part + (size_t)blockIdx.x * R;

  // ---- stage the packed hidden-layer weights (16-byte copies) and the two "ones" rows
  for (int i = tid; i < NW / 4; i += 256)
    reinterpret_cast<v4f*>(wl)[i] = reinterpret_cast<const v4f*>(img)[i];
  if (wave == 0) {
    xb[0 * BUFV + FW * RS4 + lane] = v4f{1, 0, 0, 0};
    xb[2 * BUFV + FW * RS4 + lane] = v4f{1, 0, 0, 0};
  }
  // first / last dense layer parameters of this wave (wave-uniform)
  float w0x[FF], w0t[FF], b0[FF], wLo[FF], wL[FW];
#pragma unroll
  for (int jj = 0; jj < FF; ++jj) {
    w0x[jj] = th[nd.off_w[0] + j0 + jj];
    w0t[jj] = th[nd.off_w[0] + FW + j0 + jj];
    b0[jj] = th[nd.off_b[0] + j0 + jj];
    wLo[jj] = th[nd.off_w[H] + j0 + jj];
  }
#pragma unroll
  for (int k = 0; k < FW; ++k) wL[k] = th[nd.off_w[H] + k];
  const float bL = th[nd.off_b[H]];
  float c1 = 1.0f, c2 = nu;
  if (PDE == 1) { c1 = th[nd.n_net]; c2 = __expf(th[nd.n_net + 1]); }
  const float inv_nf = (float)sd.inv_nf, inv_nu = (float)sd.inv_nu;

  // accumulators that live across tiles
  typedef float acc4 __attribute__((ext_vector_type(4)));
  acc4 dw[H][2];          // dw[d][0..1], d = 1..H-1: this wave's 16x16 tile of dW_d, split over two
                          // accumulators (channels h,q / p,r) so consecutive MFMAs never depend
#pragma unroll
  for (int d = 0; d < H; ++d) dw[d][0] = dw[d][1] = acc4{0, 0, 0, 0};
  float g0x[FF], g0t[FF], g0b[FF], gH[FF];
#pragma unroll
  for (int jj = 0; jj < FF; ++jj) g0x[jj] = g0t[jj] = g0b[jj] = gH[jj] = 0.0f;
  float gHb = 0.0f, l_res = 0.0f, l_dat = 0.0f, dl0 = 0.0f, dl1 = 0.0f;

  const int ti = wave >> 1, tj = wave & 1;                    // this wave's 16x16 tile of dW
  const int fa = min(16 * ti + (lane & 15), FW);              // A row: input feature (20 = ones)
  const int fb = min(16 * tj + (lane & 15), FW - 1);          // B column: output feature
  const int kq = (lane >> 4) * 16;                            // this lane group's 16 points
  __syncthreads();                                            // weights staged
  STAMP(1);

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int pt = tile * 64 + lane;
    const float x = xs[pt], t = ts[pt];
    const float hx = fmaf(sx, x - lbx, -1.0f), ht = fmaf(st, t - lbt, -1.0f);
    v4f stash[H][FF];                            // AGPR-resident (agpr_put4 / agpr_get4)

    // ------------------------------------------------------------------ forward
#pragma unroll
    for (int jj = 0; jj < FF; ++jj) {            // dense 0: p0 = (sx, 0), q0 = (0, st), r0 = 0
      const float z = fmaf(hx, w0x[jj], fmaf(ht, w0t[jj], b0[jj]));
      const v4f s{tanh_r5(z), sx * w0x[jj], st * w0t[jj], 0.0f};
      stash[0][jj] = agpr_put4(s);
      xb[(j0 + jj) * RS4 + lane] = channels4(s);
    }
    lds_barrier();
#pragma unroll
    for (int d = 1; d < H; ++d) {
      const v4f* __restrict__ Xin = xb + ((d - 1) & 1) * BUFV;
      v4f* __restrict__ Xout = xb + (d & 1) * BUFV;
      const float* __restrict__ wsrc = wl + ((d - 1) * 4 + wave) * WBLK;
      const float wv0 = wsrc[lane], wv1 = wsrc[64 + lane];    // [0,100) weights, [100,105) bias
      v4f xin[FW];
#pragma unroll
      for (int k = 0; k < FW; ++k) xin[k] = Xin[k * RS4 + lane];
      __builtin_amdgcn_sched_barrier(0);         // all 22 LDS reads in flight before the first FMA
      v2f axy[FF], azw[FF];
#pragma unroll
      for (int jj = 0; jj < FF; ++jj) {
        axy[jj] = v2f{lane_weight(wv0, wv1, FW * FF + jj), 0};
        azw[jj] = v2f{0, 0};
      }
      gemv_5x20(axy, azw, wv0, wv1, [&](int k) { return xin[k]; }, [](int) {});
#pragma unroll
      for (int jj = 0; jj < FF; ++jj) {
        const v4f s{tanh_r5(axy[jj].x), axy[jj].y, azw[jj].x, azw[jj].y};
        stash[d][jj] = agpr_put4(s);
        Xout[(j0 + jj) * RS4 + lane] = channels4(s);
      }
      lds_barrier();
      STAMP(1 + d);
    }
    // linear output layer (every wave computes it) -> o = (u, u_x, u_t, u_xx)
    v4f o{bL, 0, 0, 0};
    {
      const v4f* __restrict__ Xin = xb + ((H - 1) & 1) * BUFV;
      v4f xin[FW];
#pragma unroll
      for (int k = 0; k < FW; ++k) xin[k] = Xin[k * RS4 + lane];
#pragma unroll
      for (int k = 0; k < FW; ++k) {
        o.x = fmaf(xin[k].x, wL[k], o.x); o.y = fmaf(xin[k].y, wL[k], o.y);
        o.z = fmaf(xin[k].z, wL[k], o.z); o.w = fmaf(xin[k].w, wL[k], o.w);
      }
    }

    // ------------------------------------------------------------------ seeds + loss parts
    v4f sb{0, 0, 0, 0};
    {
      const int cls = point_class(sd, pt);
      const bool res = (PDE == 0) ? (cls == CLS_COL) : (cls == CLS_DATA);
      if (res) {
        const float wgt = (PDE == 0) ? inv_nf : inv_nu;
        const float f = o.z + c1 * o.x * o.y - c2 * o.w;
        const float fbar = 2.0f * f * wgt;
        l_res += f * f * wgt;
        sb = v4f{fbar * c1 * o.y, fbar * c1 * o.x, fbar, -c2 * fbar};
        if (PDE == 1) { dl0 += fbar * o.x * o.y; dl1 -= fbar * c2 * o.w; }
      }
      if (cls == CLS_DATA) {
        const float dd = o.x - tgt[pt];
        l_dat += dd * dd * inv_nu;
        sb.x += 2.0f * dd * inv_nu;
      }
    }

    // ------------------------------------------------------------------ reverse sweep
    v4f ob[FF];                              // adjoint of the outputs of the layer below, own features
    {  // dense H (linear): z_bar = sb
      gHb += sb.x;
#pragma unroll
      for (int kk = 0; kk < FF; ++kk) {
        const v4f in = channels4(agpr_get4(stash[H - 1][kk]));
        gH[kk] += fmaf(in.w, sb.w, fmaf(in.z, sb.z, fmaf(in.y, sb.y, in.x * sb.x)));
        ob[kk] = sb * wLo[kk];
      }
    }
    lds_barrier();          // every wave is done reading the forward tile before it is overwritten
    STAMP(H + 1);
#pragma unroll
    for (int d = H - 1; d >= 1; --d) {
      const int pair = (H - 1 - d) & 1;
      v4f* __restrict__ IN = xb + (2 * pair) * BUFV;
      v4f* __restrict__ ZB = xb + (2 * pair + 1) * BUFV;
      // phase A: publish own z_bar (layer d) and own layer-(d-1) output channels
#pragma unroll
      for (int kk = 0; kk < FF; ++kk) {
        ZB[(j0 + kk) * RS4 + lane] = preact_adjoint4(agpr_get4(stash[d][kk]), ob[kk]);
        IN[(j0 + kk) * RS4 + lane] = channels4(agpr_get4(stash[d - 1][kk]));
      }
      lds_barrier();
      // phase B: vector pipe -- adjoint of own layer-(d-1) outputs = sum_j z_bar_j W_d[k][j];
      //          matrix pipe -- dW_d tile += IN^T . ZB over the 256 (point,channel) rows, one
      //          MFMA dropped into every side slot of the GEMV (64 of its 100 slots)
      const float* __restrict__ wsrc = wl + ((d - 1) * 4 + wave) * WBLK + WBLK_REV;
      const float wv0 = wsrc[lane], wv1 = wsrc[64 + lane];
      v4f zin[FW];
#pragma unroll
      for (int j = 0; j < FW; ++j) zin[j] = ZB[j * RS4 + lane];
      v4f a4[2], b4[2];                        // MFMA operand ring: group j in slot j & 1
      a4[0] = IN[fa * RS4 + kq + 0]; b4[0] = ZB[fb * RS4 + kq + 0];
      a4[1] = IN[fa * RS4 + kq + 1]; b4[1] = ZB[fb * RS4 + kq + 1];
      __builtin_amdgcn_sched_barrier(0);
      v2f oxy[FF], ozw[FF];
#pragma unroll
      for (int kk = 0; kk < FF; ++kk) oxy[kk] = ozw[kk] = v2f{0, 0};
      acc4 acc0 = dw[d][0], acc1 = dw[d][1];
      gemv_5x20(oxy, ozw, wv0, wv1, [&](int j) { return zin[j]; }, [&](int slot) {
        // slots 0..99; MFMA m = 0..63 goes to slot m + (m * 9) / 16 (spread: 64 of 100)
        // inverse: slot hosts an MFMA iff it equals m + (9m)/16 for some m
        const int m = (slot * 16 + 24) / 25;                   // candidate
        if (m < 64 && m + (m * 9) / 16 == slot) {
          const int j = m >> 2, c = m & 3;
          const v4f a = a4[j & 1], b = b4[j & 1];
          if (c == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
          if (c == 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
          if (c == 2) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
          if (c == 3) {
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
            if (j + 2 < 16) {                                  // refill this ring slot two groups ahead
              a4[j & 1] = IN[fa * RS4 + kq + j + 2];
              b4[j & 1] = ZB[fb * RS4 + kq + j + 2];
            }
          }
        }
      });
      dw[d][0] = acc0; dw[d][1] = acc1;
#pragma unroll
      for (int kk = 0; kk < FF; ++kk) ob[kk] = v4f{oxy[kk].x, oxy[kk].y, ozw[kk].x, ozw[kk].y};
      STAMP(2 * H + 1 - d);
    }
    {  // dense 0: inputs (hx, ht), p0 = (sx, 0), q0 = (0, st)
#pragma unroll
      for (int kk = 0; kk < FF; ++kk) {
        const v4f zb = preact_adjoint4(agpr_get4(stash[0][kk]), ob[kk]);
        g0x[kk] += fmaf(hx, zb.x, sx * zb.y);
        g0t[kk] += fmaf(ht, zb.x, st * zb.z);
        g0b[kk] += zb.x;
      }
    }
    lds_barrier();          // the next tile's first layer overwrites an exchange buffer
  }
  STAMP(2 * H + 1);

  // -------------------------------------------------------------------- one gradient row per workgroup
  {
    float kx = 0, kt = 0, kb = 0, kh = 0;
#pragma unroll
    for (int kk = 0; kk < FF; ++kk) {
      const float a = wave_sum(g0x[kk]), b = wave_sum(g0t[kk]), c = wave_sum(g0b[kk]), e = wave_sum(gH[kk]);
      if (lane == kk) { kx = a; kt = b; kb = c; kh = e; }
    }
    if (lane < FF) {
      row[nd.off_w[0] + j0 + lane] = kx;
      row[nd.off_w[0] + FW + j0 + lane] = kt;
      row[nd.off_b[0] + j0 + lane] = kb;
      row[nd.off_w[H] + j0 + lane] = kh;
    }
    if (wave == 0) {
      const float a = wave_sum(l_res), b = wave_sum(l_dat), g = wave_sum(gHb);
      if (lane == 0) {
        row[nd.n_theta + 0] = a; row[nd.n_theta + 1] = b; row[nd.n_theta + 2] = 0.0f;
        row[nd.off_b[H]] = g;
      }
      if (PDE == 1) {
        const float g1 = wave_sum(dl0), g2 = wave_sum(dl1);
        if (lane == 0) { row[nd.n_net] = g1; row[nd.n_net + 1] = g2; }
      }
    }
    const int jg = 16 * tj + (lane & 15);
    if (jg < FW) {
#pragma unroll
      for (int d = 1; d < H; ++d) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ig = 16 * ti + (lane >> 4) * 4 + r;
          const float g = dw[d][0][r] + dw[d][1][r];
          if (ig < FW) row[nd.off_w[d] + ig * FW + jg] = g;
          else if (ig == FW) row[nd.off_b[d] + jg] = g;
        }
      }
    }
  }
  STAMP(2 * H + 2);
}

// returns a hipError_t (0 = ok)
template <int PDE, int H>
inline int fused20r_launch(const NetDesc& nd, const SetDesc& sd, const float* th, const float* img,
                           const float* xs, const float* ts, const float* tgt, float lbx, float lbt,
                           float sx, float st, float nu, float* part, int R, int n_wg,
                           hipStream_t stream, long long* stamps = nullptr) {
  const size_t lds = fused20r_lds_bytes(H);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_fused20r<PDE, H>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_fused20r<PDE, H>), dim3(n_wg), dim3(256), lds, stream, nd, sd, th, img, xs,
                     ts, tgt, lbx, lbt, sx, st, nu, part, R, sd.n_pad / 64, stamps);
  return (int)hipGetLastError();
}

}  // namespace pinn
