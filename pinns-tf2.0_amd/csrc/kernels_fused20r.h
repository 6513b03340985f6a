// kernels_fused20r.h -- k_fused20r: k_fused20m (kernels_fused20m.h: float32, width 20, every contraction on the matrix
// instructions, lane = point, 64-point tiles, register stash) for the THROUGHPUT regime, with recompute instead of stash
// so that TWO workgroups share a CU.
//
// Why (SURVEY 7.3-3 "measure both").  k_fused20m holds 176 VGPR + 160 AGPR (the stash of
// 8 layers x 5 features x 4 channels) and 116 KB of LDS: one workgroup = one wave per SIMD per CU, matrix pipe busy 29 %
// of the wave cycles, a third of them parked on barriers / s_waitcnt, nothing to overlap with.  The upper bound was
// measured first (-DPINN_ABL=8, profiles/r03_ablate_two_wg.txt: half the stash dropped without paying for it, one tile
// pair, two workgroups per CU): N_f = 10^6 1312 -> 842 us per Adam step.  This kernel pays for it:
//   * stash only the EVEN layers in AGPRs (4 x 20 registers), the last hidden layer live in VGPRs; an odd layer is
//     recomputed in the reverse sweep from the even layer under it -- one forward GEMV (100 matrix instructions, two
//     barriers) per recomputed layer, three per tile at H = 8: +13 % matrix work;
//   * ONE (IN, ZBAR) exchange-tile pair (the group-4 barrier that ends a layer already orders its reads against the
//     next layer's writes), epilogue in rounds of four layers inside the same 72 KB;
//   * __launch_bounds__(256, 2), grid = min(tiles, 2 x CUs).
// RESULT (round 3, MI355X, profiles/r03_fused20r.txt): parity-green on the first run (the multi-tile float32 tests of
// tests/test_gpu_parity.py, Burgers and identification) and SLOWER than k_fused20m: N_f = 10^6 1576 vs 1308 us per Adam
// step, 125 000 points 236 vs 187 us.  hipcc splits the 256 registers of a two-waves-per-SIMD kernel 128 VGPR / 128
// AGPR as soon as AGPRs are named (no source-level attribute reaches amdgpu-agpr-alloc); the working set of the reverse
// sweep (64 registers of cross-tile gradient accumulators, the 80-register operand block of a GEMV, adjoints) needs ~176
// VGPRs, so 772 B per lane go to scratch (the upper-bound build, without the recomputed layer's operands, had 312 B
// and ran at 842 us).  OPT-IN (PINN_F32_RECOMPUTE=1, fused20m_plan); k_fused20m stays the product path.  What it would
// take: a 176 / 80 register split (assembler-level kernel descriptor), or gradient accumulators that are not pinned
// across tiles.
//
// Arithmetic is k_fused20m's, operation for operation (the recomputed layer repeats the forward layer's instructions
// on the same inputs, so it reproduces the stashed values bit for bit); only the summation grouping over workgroups
// differs (twice as many partial rows).  Selected by the engine when a workgroup has at least a few tiles
// (fused20r_pays): the metric's N_f = 10 000 stays on k_fused20m's one-tile variant.
#pragma once
#include "kernels_fused20m.h"

namespace pinn {

inline size_t fused20r_lds_bytes(int n_hidden) {
  return fused20m_image_floats(n_hidden) * 4 + (size_t)(2 * FROWS + 4) * 65 * 16;
}

template <int PDE, int H>
__global__ __launch_bounds__(256, 2) void k_fused20r(NetDesc nd, SetDesc sd,
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
  constexpr int NW = ((H - 1) * WIMG + 1023) / 1024 * 1024;   // floats of weight image (whole rounds of 4 DMA pieces)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  v4f* const xb = reinterpret_cast<v4f*>(wl + NW);
  v4f* const Q = xb + 2 * BUFV;                     // group-4 meeting point: [4][RS4] float4
  float* const Qf = reinterpret_cast<float*>(Q);

  STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* __restrict__ row = part + (size_t)blockIdx.x * R;
  // this wave's features: local jj = 0..3 -> 4*wave + jj, jj = 4 -> 16 + wave
  auto feat = [&](int jj) { return jj < 4 ? 4 * wave + jj : 16 + wave; };

  // first tile's coordinates
  int tile = blockIdx.x;
  float x = 0.0f, t = 0.0f;
  if (tile < n_tiles) { x = xs[tile * 64 + lane]; t = ts[tile * 64 + lane]; }

  // first hidden layer of the first tile: its patterns and biases come straight from the global
  // image into registers, in the same memory round trip as the coordinates -- the LDS image is only
  // needed from the second hidden layer on
  v4f pre_o[5], pre_g[5], pre_b, pre_bg;
  load_patterns(img, wave, lane, pre_o, pre_g);
  pre_b = *reinterpret_cast<const v4f*>(img + 2 * FW * FW + 4 * wave);
  pre_bg = *reinterpret_cast<const v4f*>(img + 2 * FW * FW + 16);

  STAMP(19);
  if (wave == 0) {
    xb[0 * BUFV + FW * RS4 + lane] = v4f{1, 0, 0, 0};     // the ones row of the IN slot (bias gradients)
  }
  // first / last dense layer parameters of this wave (wave-uniform)
  float w0x[FF], w0t[FF], b0[FF], wLo[FF], wL[FW];
#pragma unroll
  for (int jj = 0; jj < FF; ++jj) {
    w0x[jj] = th[nd.off_w[0] + feat(jj)];
    w0t[jj] = th[nd.off_w[0] + FW + feat(jj)];
    b0[jj] = th[nd.off_b[0] + feat(jj)];
    wLo[jj] = th[nd.off_w[H] + feat(jj)];
  }
#pragma unroll
  for (int k = 0; k < FW; ++k) wL[k] = th[nd.off_w[H] + k];
  const float bL = th[nd.off_b[H]];
  float c1 = 1.0f, c2 = nu;
  if (PDE == 1) { c1 = th[nd.n_net]; c2 = __expf(th[nd.n_net + 1]); }
  const float inv_nf = (float)sd.inv_nf, inv_nu = (float)sd.inv_nu;

  // accumulators that live across tiles
  // dW_d (+ db_d as row 20) = IN^T . ZB over the tile's 256 (point,channel) rows, as
  //   dwm[d]: the 16x16 block (k < 16, j < 16) on v_mfma_f32_16x16x4 -- this wave's quarter of the rows
  //           (points 16q + 4w .. +3 of every lane group q), 16 MFMAs;
  //   dwf[d]: the fringe (k = 16..20 or j = 16..19; 164 entries) as 14 of the 16 independent 4x4
  //           blocks of v_mfma_f32_4x4x1_16B, one (point,channel) row per instruction -- this wave's
  //           16 points, 64 MFMAs.
  // (Padding the fringe out to three more 16x16 tiles, one tile per wave, cost 64 x 32-cycle
  // MFMAs per wave and layer; this split costs 16 x 32 + 64 x 8.)  The four waves' partial sums meet
  // once per kernel in the epilogue.
  acc4 dwm[H], dwf[H];
#pragma unroll
  for (int d = 0; d < H; ++d) dwm[d] = dwf[d] = acc4{0, 0, 0, 0};
  float g0x[FF], g0t[FF], g0b[FF], gH[FF];
#pragma unroll
  for (int jj = 0; jj < FF; ++jj) g0x[jj] = g0t[jj] = g0b[jj] = gH[jj] = 0.0f;
  float gHb = 0.0f, l_res = 0.0f, l_dat = 0.0f, dl0 = 0.0f, dl1 = 0.0f;

  // main block operands: A row = input feature lane%16, B column = output feature lane%16,
  // lane group q = lane/16 supplies point 16q + 4w + jj at step jj
  const int mrow = (lane & 15) * RS4 + (lane >> 4) * 16 + 4 * wave;
  // fringe block b = lane/4 -> (input-feature group kg, output-feature group jg):
  //   b 0..3: (4, b)   b 4..7: (5 = ones row, b-4)   b 8..11: (b-8, 4)   b 12: (4, 4)   b 13..15: (5, 4)
  const int fblk = lane >> 2;
  const int fkg = fblk < 4 ? 4 : fblk < 8 ? 5 : fblk < 12 ? fblk - 8 : fblk == 12 ? 4 : 5;
  const int fjg = fblk < 8 ? (fblk & 3) : 4;
  const int farow = min(4 * fkg + (lane & 3), FW) * RS4 + 16 * wave;     // rows past the ones row repeat it
  const int fbrow = (4 * fjg + (lane & 3)) * RS4 + 16 * wave;
  const int qw = 4 * lane + wave;                             // float index of (point, channel = wave) in a Q row
  STAMP(1);
  bool image_pending = true;

  for (; tile < n_tiles; tile += gridDim.x) {
    const int pt = tile * 64 + lane;
    const float hx = fmaf(sx, x - lbx, -1.0f), ht = fmaf(st, t - lbt, -1.0f);
    {  // next tile's coordinates: in flight during this tile
      const int nt = tile + gridDim.x;
      if (nt < n_tiles) { x = xs[nt * 64 + lane]; t = ts[nt * 64 + lane]; }
    }
    // Stash: EVEN layers only, in AGPRs (H/2 x 20 registers); the last hidden layer stays live in VGPRs (top); an odd
    // layer below is recomputed in the reverse sweep, from the even layer under it, just before it is needed (rc)
    v4f stash[H][FF];
    v4f top[FF], rc[FF];

    // ------------------------------------------------------------------ forward
#pragma unroll
    for (int jj = 0; jj < FF; ++jj) {            // dense 0: p0 = (sx, 0), q0 = (0, st), r0 = 0
      const float z = fmaf(hx, w0x[jj], fmaf(ht, w0t[jj], b0[jj]));
      const v4f s{tanh_r5(z), sx * w0x[jj], st * w0t[jj], 0.0f};
      stash[0][jj] = agpr_put4(s);
      xb[feat(jj) * RS4 + lane] = channels4(s);
    }
    if (image_pending) {                         // the pre-loaded patterns have arrived (same round trip as x, t)
#pragma unroll
      for (int m = 0; m < 5; ++m) { consume4(pre_o[m]); consume4(pre_g[m]); }
      consume4(pre_b); consume4(pre_bg);
      STAMP(22);
    }
    lds_barrier();
    if (image_pending) {
      // ---- weight image -> LDS by asynchronous LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
      // instruction, no registers), in flight during the first hidden layer and drained before the
      // second.  Written as inline assembly on purpose: hipcc makes every LDS access that follows a
      // DMA it knows about wait for vmcnt(0) (it cannot tell the image from the exchange tiles),
      // which would put the whole fetch latency back in front of the first layer.  Unseen, the DMA
      // can only make the compiler's own vmcnt waits stricter (counters retire in order), and all
      // its earlier loads have been consumed by now.
#pragma unroll
      for (int m = 0; m < NW / 1024; ++m)
        lds_dma_b128(img + (4 * m + wave) * 256 + lane * 4, wl + (4 * m + wave) * 256);
      STAMP(23);
    }
#pragma unroll
    for (int d = 1; d < H; ++d) {
      const v4f* __restrict__ Xin = xb + ((d - 1) & 1) * BUFV;
      v4f* __restrict__ Xout = xb + (d & 1) * BUFV;
      const float* __restrict__ wimg = wl + (d - 1) * WIMG;
      v4f xin[FW];
#pragma unroll
      for (int k = 0; k < FW; ++k) xin[k] = Xin[k * RS4 + lane];
      acc4 acc_own[4], acc_g4;
      v4f ao[5], ag[5];
      if (d == 1 && image_pending) {
#pragma unroll
        for (int m = 0; m < 5; ++m) { ao[m] = pre_o[m]; ag[m] = pre_g[m]; }
        acc_own[0] = pre_b; acc_g4 = pre_bg;
      } else {
        load_patterns(wimg, wave, lane, ao, ag);
        acc_own[0] = *reinterpret_cast<const v4f*>(wimg + 2 * FW * FW + 4 * wave);   // bias b_d[4w..4w+3]
        acc_g4 = *reinterpret_cast<const v4f*>(wimg + 2 * FW * FW + 16);
      }
      acc_own[1] = acc_own[2] = acc_own[3] = acc4{0, 0, 0, 0};
      if (wave != 0) acc_g4 = acc4{0, 0, 0, 0};                                      // bias rides on channel h
      if (d == 4) STAMP(24);
      // group 4: this wave's channel of features 16..19 is published as soon as it is complete
      gemv_mfma(acc_own, acc_g4, ao, ag, wave, lane, xin,
                [&](const acc4& g) {
#pragma unroll
                  for (int i = 0; i < 4; ++i) Qf[i * RS4 * 4 + qw] = g[i];
                },
                [](int) {});
      if (d == 4) STAMP(25);
      if (d == 1 && image_pending) STAMP(20);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const v4f s{tanh_r5(acc_own[0][jj]), acc_own[1][jj], acc_own[2][jj], acc_own[3][jj]};
        if (d == H - 1) top[jj] = s; else if ((d & 1) == 0) stash[d][jj] = agpr_put4(s);
        Xout[(4 * wave + jj) * RS4 + lane] = channels4(s);
      }
      if (d == 4) STAMP(26);
      lds_barrier();
      if (d == 4) STAMP(27);
      {
        const v4f z4 = Q[wave * RS4 + lane];      // feature 16+wave: (h, p, q, r) pre-activations
        const v4f s{tanh_r5(z4.x), z4.y, z4.z, z4.w};
        if (d == H - 1) top[4] = s; else if ((d & 1) == 0) stash[d][4] = agpr_put4(s);
        Xout[(16 + wave) * RS4 + lane] = channels4(s);
      }
      if (d == 1 && image_pending) {             // first tile only: this wave's DMA pieces have landed
        STAMP(21);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        image_pending = false;
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
        const v4f in = channels4(top[kk]);
        gH[kk] += fmaf(in.w, sb.w, fmaf(in.z, sb.z, fmaf(in.y, sb.y, in.x * sb.x)));
        ob[kk] = sb * wLo[kk];
      }
    }
    lds_barrier();          // every wave is done reading the forward tile before it is overwritten
    STAMP(H + 1);
#pragma unroll
    for (int d = H - 1; d >= 1; --d) {
      constexpr int pair = 0;                    // ONE (IN, ZBAR) tile pair: the group-4 barrier at the end of a
                                                 // layer already orders its reads against the next layer's writes
      v4f* __restrict__ IN = xb + (2 * pair) * BUFV;
      v4f* __restrict__ ZB = xb + (2 * pair + 1) * BUFV;
      // ---- recompute: layer d-1 is an odd layer whose stash was not kept (d even).  Its inputs are the output channels
      // of layer d-2 (kept): published into the IN slot (free: the previous layer's phase B ended behind a barrier),
      // one forward GEMV with the group-4 exchange, tanh -> rc, which serves this reverse layer (as its inputs) and the
      // next one (as its pre-activations).  +100 matrix instructions and two barriers per recomputed layer.
      if ((d & 1) == 0 && d - 1 >= 1) {
#pragma unroll
        for (int kk = 0; kk < FF; ++kk) IN[feat(kk) * RS4 + lane] = channels4(agpr_get4(stash[d - 2][kk]));
        lds_barrier();
        const float* __restrict__ wimg2 = wl + (d - 2) * WIMG;           // image of dense d-1: forward patterns + bias
        v4f xin[FW];
#pragma unroll
        for (int k = 0; k < FW; ++k) xin[k] = IN[k * RS4 + lane];
        acc4 acc_own[4], acc_g4;
        v4f ao[5], ag[5];
        load_patterns(wimg2, wave, lane, ao, ag);
        acc_own[0] = *reinterpret_cast<const v4f*>(wimg2 + 2 * FW * FW + 4 * wave);
        acc_g4 = *reinterpret_cast<const v4f*>(wimg2 + 2 * FW * FW + 16);
        acc_own[1] = acc_own[2] = acc_own[3] = acc4{0, 0, 0, 0};
        if (wave != 0) acc_g4 = acc4{0, 0, 0, 0};
        gemv_mfma(acc_own, acc_g4, ao, ag, wave, lane, xin,
                  [&](const acc4& g) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Qf[i * RS4 * 4 + qw] = g[i];
                  },
                  [](int) {});
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          rc[jj] = v4f{tanh_r5(acc_own[0][jj]), acc_own[1][jj], acc_own[2][jj], acc_own[3][jj]};
        lds_barrier();                           // Q complete; every wave has its xin: the IN slot may be rewritten
        {
          const v4f z4 = Q[wave * RS4 + lane];
          rc[4] = v4f{tanh_r5(z4.x), z4.y, z4.z, z4.w};
        }
      }
      // phase A: publish own z_bar (layer d) and own layer-(d-1) output channels
#pragma unroll
      for (int kk = 0; kk < FF; ++kk) {
        const v4f sd = d == H - 1 ? top[kk] : (d & 1) == 0 ? agpr_get4(stash[d][kk]) : rc[kk];
        const v4f sm = ((d - 1) & 1) == 0 ? agpr_get4(stash[d - 1][kk]) : rc[kk];
        ZB[feat(kk) * RS4 + lane] = preact_adjoint4(sd, ob[kk]);
        IN[feat(kk) * RS4 + lane] = channels4(sm);
      }
      if (d == 4) STAMP(28);
      lds_barrier();
      if (d == 4) STAMP(29);
      // phase B: adjoint of own layer-(d-1) outputs  in_bar[k] = sum_j z_bar_j W_d[k][j]  (4x4x1 MFMAs)
      //          and dW_d tile += IN^T . ZB over the 256 (point,channel) rows (16x16x4 MFMAs),
      //          one of the latter after every pair of the former
      const float* __restrict__ wimg = wl + (d - 1) * WIMG + FW * FW;
      v4f zin[FW];
#pragma unroll
      for (int j = 0; j < FW; ++j) zin[j] = ZB[j * RS4 + lane];
      // operand rings, refilled two steps ahead: main block step jj = 0..3 (4 MFMAs each, one per
      // channel), fringe step p = 0..15 (one point, 4 MFMAs)
      v4f ma[2], mb[2], fa4[2], fb4[2];
      ma[0] = IN[mrow + 0]; mb[0] = ZB[mrow + 0];
      ma[1] = IN[mrow + 1]; mb[1] = ZB[mrow + 1];
      fa4[0] = IN[farow + 0]; fb4[0] = ZB[fbrow + 0];
      fa4[1] = IN[farow + 1]; fb4[1] = ZB[fbrow + 1];
      acc4 acc_own[4], acc_g4 = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 4; ++c) acc_own[c] = acc4{0, 0, 0, 0};
      acc4 accm = dwm[d], accf = dwf[d];
      auto dw_mfma = [&](int s) {              // s = 0..79, one after every own-group GEMV MFMA
        if (s % 5 == 4) {                      // 16 main-block MFMAs: step jj = m/4, channel m%4
          const int m = s / 5, jj = m >> 2, c = m & 3;
          const v4f a = ma[jj & 1], b = mb[jj & 1];
          accm = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], accm, 0, 0, 0);
          if (c == 3 && jj + 2 < 4) { ma[jj & 1] = IN[mrow + jj + 2]; mb[jj & 1] = ZB[mrow + jj + 2]; }
        } else {                               // 64 fringe MFMAs: point p = m/4, channel m%4
          const int m = s - s / 5, p = m >> 2, c = m & 3;
          const v4f a = fa4[p & 1], b = fb4[p & 1];
          accf = __builtin_amdgcn_mfma_f32_4x4x1f32(a[c], b[c], accf, 0, 0, 0);
          if (c == 3 && p + 2 < 16) { fa4[p & 1] = IN[farow + p + 2]; fb4[p & 1] = ZB[fbrow + p + 2]; }
        }
      };
      v4f ao[5], ag[5];
      load_patterns(wimg, wave, lane, ao, ag);
      gemv_mfma(acc_own, acc_g4, ao, ag, wave, lane, zin,
                [&](const acc4& g) {
#pragma unroll
                  for (int i = 0; i < 4; ++i) Qf[i * RS4 * 4 + qw] = g[i];
                },
                [&](int slot) { dw_mfma(slot); });   // slots 0..79
      dwm[d] = accm; dwf[d] = accf;
      if (d == 4) STAMP(30);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) ob[kk] = v4f{acc_own[0][kk], acc_own[1][kk], acc_own[2][kk], acc_own[3][kk]};
      lds_barrier();
      ob[4] = Q[wave * RS4 + lane];
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
    // As k_fused20m, but inside 72 KB of LDS: the whole allocation (the weight image is dead now) holds the four
    // waves' 25 row-sum rows and, four layers at a time, the weight-gradient partials of all waves.
    constexpr int NV = 25, RSF = 68;                 // rows; padded row stride in floats (16-B aligned)
    float* __restrict__ red = wl + wave * (NV * RSF);
    constexpr int PSR = 4 * 2 * 64;                  // float4 per wave and round: four layers x (main | fringe) x lanes
    v4f* const psum = reinterpret_cast<v4f*>(wl) + NV * RSF;
    static_assert(NV * RSF + 4 * PSR <= NW / 4 + 2 * BUFV + 4 * RS4, "epilogue fits in the workgroup's LDS");
#pragma unroll
    for (int kk = 0; kk < FF; ++kk) {
      red[(0 + kk) * RSF + lane] = g0x[kk];
      red[(5 + kk) * RSF + lane] = g0t[kk];
      red[(10 + kk) * RSF + lane] = g0b[kk];
      red[(15 + kk) * RSF + lane] = gH[kk];
    }
    red[20 * RSF + lane] = l_res; red[21 * RSF + lane] = l_dat; red[22 * RSF + lane] = gHb;
    red[23 * RSF + lane] = dl0;   red[24 * RSF + lane] = dl1;
    const int v = lane & 31, half = lane >> 5;
    float tot = 0.0f;
    if (v < NV) {
      const v4f* __restrict__ src = reinterpret_cast<const v4f*>(red + v * RSF + 32 * half);
      v4f q[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) q[m] = src[m];
      v4f s4 = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
      tot = (s4.x + s4.y) + (s4.z + s4.w);
    }
    tot += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(tot)));
    if (lane < 20) {
      const int grp = lane / 5, kk = lane - grp * 5;
      const int f = kk < 4 ? 4 * wave + kk : 16 + wave;
      const int base = grp == 0 ? nd.off_w[0] : grp == 1 ? nd.off_w[0] + FW : grp == 2 ? nd.off_b[0] : nd.off_w[H];
      row[base + f] = tot;
    } else if (wave == 0 && lane < NV) {
      if (lane == 20) { row[nd.n_theta + 0] = tot; row[nd.n_theta + 2] = 0.0f; }
      if (lane == 21) row[nd.n_theta + 1] = tot;
      if (lane == 22) row[nd.off_b[H]] = tot;
      if (PDE == 1 && lane == 23) row[nd.n_net] = tot;
      if (PDE == 1 && lane == 24) row[nd.n_net + 1] = tot;
    }
    STAMP(31);
    // dW / db in rounds of four layers: every wave publishes its partial blocks of layers 4 r + 1 .. 4 r + 4, wave w
    // adds up layer 4 r + w + 1 over the four waves (fixed order).  b_d sits right behind W_d in the flat layout, so
    // input-feature row 20 (the ones row) lands on the bias.
    const int km = 4 * (lane >> 4), jm = lane & 15;              // main block: VGPR r -> input feature km + r
    const int kf = 4 * fkg, jf = 4 * fjg + (lane & 3);           // fringe block: VGPR r -> input feature kf + r
    const bool okf = fblk <= 13;
#pragma unroll
    for (int rnd = 0; rnd < (H - 1 + 3) / 4; ++rnd) {
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int d = 4 * rnd + l + 1;
        if (d < H) {
          psum[wave * PSR + (l * 2 + 0) * 64 + lane] = dwm[d];
          psum[wave * PSR + (l * 2 + 1) * 64 + lane] = dwf[d];
        }
      }
      lds_barrier();
      const int d1 = 4 * rnd + wave;
      if (d1 < H - 1) {
        float* __restrict__ dst = row + nd.off_w[1] + d1 * (FW * FW + FW);
        const v4f* __restrict__ src = psum + (wave * 2) * 64 + lane;
        const v4f m4 = (src[0 * PSR] + src[1 * PSR]) + (src[2 * PSR] + src[3 * PSR]);
        const v4f f4 = (src[0 * PSR + 64] + src[1 * PSR + 64]) + (src[2 * PSR + 64] + src[3 * PSR + 64]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[(km + r) * FW + jm] = m4[r];
          if (okf && kf + r <= FW) dst[(kf + r) * FW + jf] = f4[r];
        }
      }
      lds_barrier();                                 // the next round overwrites the partials
    }
  }
  STAMP(2 * H + 2);
}

// returns a hipError_t (0 = ok); one partial gradient row per workgroup
template <int PDE, int H>
inline int fused20r_launch(const NetDesc& nd, const SetDesc& sd, const float* th, const float* img,
                           const float* xs, const float* ts, const float* tgt, float lbx, float lbt,
                           float sx, float st, float nu, float* part, int R, int n_wg,
                           hipStream_t stream, long long* stamps = nullptr, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr) {
  static_assert(H % 2 == 0, "the keep-even / recompute-odd schedule is written for an even number of hidden layers");
  const size_t lds = fused20r_lds_bytes(H);
  static unsigned long long attr_set = 0;
  if (first_call_on_device(attr_set)) {
    const hipError_t e = hipFuncSetAttribute((const void*)k_fused20r<PDE, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_tiles = sd.n_pad / 64;
  if (ev_start && ev_stop)
    hipExtLaunchKernelGGL((k_fused20r<PDE, H>), dim3(n_wg), dim3(256), lds, stream, ev_start, ev_stop, 0, nd, sd,
                          th, img, xs, ts, tgt, lbx, lbt, sx, st, nu, part, R, n_tiles, stamps);
  else
    hipLaunchKernelGGL((k_fused20r<PDE, H>), dim3(n_wg), dim3(256), lds, stream, nd, sd, th, img, xs,
                       ts, tgt, lbx, lbt, sx, st, nu, part, R, n_tiles, stamps);
  return (int)hipGetLastError();
}

}  // namespace pinn
