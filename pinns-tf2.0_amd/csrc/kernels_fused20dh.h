// kernels_fused20dh.h -- k_fused20dh: the float64 loss+gradient kernel of kernels_fused20d.h for launches that are ONE
// tile deep (the metric's N_f = 10 000: fewer tiles than compute units), with the weight-gradient work of three
// 16-point waves shared with a fourth, otherwise idle wave of the same workgroup.
//
// Why.  k_fused20d puts 16 points on a wave and the waves never talk to each other, so at N_f = 10 000 + 100 points a
// launch is 632 waves on the 1024 SIMDs of the chip and lasts as long as ONE wave's dependent chain (87 k cycles,
// profiles/r02_stamps_fused20d_f64.txt); 392 SIMDs idle.  Per reverse layer that chain is 7.4 k cycles of which ~4 k
// serve the weight gradient dW_d = IN^T . ZBAR only (105 of the 205 matrix instructions, the 80 lane rotations that
// bring both operands into point-major form, 30 block folds + LDS stores; ablation profiles/r02_ablate_fused20d.txt:
// fold + accumulate 5.3 us, rotations 4.5 us of a 42.6 us step).  dW_d needs nothing but the layer's input channels and
// pre-activation adjoints, and nothing downstream in the sweep needs dW_d: it can run anywhere, any time later.
//
// What.  Workgroup = 3 "main" waves (16 points each: tiles of 48 points, <= 256 tiles) + 1 helper wave:
//   * the mains run the forward sweep, the seeds, the adjoint GEMVs and, of every hidden layer's 30 gradient blocks,
//     the 10 of in-groups m = 0, 1 and the 5 bias blocks (plus dense 0 / dense H);
//   * per hidden layer a main parks its 20 pre-activation adjoints and the 12 input channels of in-groups m = 2, 3, 4
//     in LDS (32 x ds_write_b64) and raises a flag; the helper reads them back ROTATED -- lane l reads the slot of lane
//     rot(l): the lane rotation that costs the mains two ds_bpermute per value is free in the read address -- and
//     accumulates the 15 blocks (m = 2, 3, 4) over the three mains in registers: one fold + store per block and layer
//     instead of three;
//   * the mains get their own rotated adjoints the same way (read back what they just wrote) instead of 40 bpermutes;
//   * no s_barrier inside the sweeps: two LDS flags per layer (data ready: one per main; buffer consumed: helper), polled
//     with s_sleep; in steady state nobody waits (the helper frees the buffer ~2.6 k cycles after it was filled, the
//     mains refill it ~3.8 k cycles later).
// Gradient blocks are stored (never accumulated: every block is produced once per launch) in compact per-wave lists;
// the epilogue adds the three mains' copies in fixed order / takes the helper's -> one gradient row per workgroup, no
// atomics, bit-reproducible.  Arithmetic per block = k_fused20d's (same instruction, same operand order over the
// channels) except that the helper sums a block over 48 points in one accumulator chain where k_fused20d adds three
// 16-point partials through LDS: results agree to rounding (tests/test_gpu_parity.py, 1e-11).
//
// RESULT (round 3, MI355X): correct on the first run (every float64 parity test of tests/test_gpu_parity.py passes on it,
// 1e-11 against goldens and oracle) and NOT faster: 41.90 us per Adam step against 41.9 for k_fused20d at N_f = 10 000
// (helper taking 2 / 3 / 4 of the five in-groups: 43.1 / 41.9 / 44.8 us; profiles/r03_helper_wave.txt).  The timeline
// (profiles/r03_stamps_fused20dh.txt) says why: a main's reverse layer is still 6.5-7.4 k cycles (7.2 k before) --
// parking + reading back the adjoints 2.0 k, its remaining 45 gradient-block matrix instructions with their folds
// 2.65 k (3.7 x their pipe time), GEMV 2.0 k -- and the helper needs 6.8 k per layer for 180 matrix instructions
// (2.9 k of pipe): both roles are bound by the same thing as k_fused20d itself, a single wave's in-order issue, where
// hipcc emits the matrix instructions back to back (1574 of 2190 gaps between them are empty) and everything else in
// ~100 clumps, so that matrix pipe and vector ALU never overlap.  Moving instructions to another wave does not create
// that overlap; asking the scheduler for it (sched_group_barrier pipelines per reverse layer, amdgpu-sched-strategy
// max-ilp / iterative-ilp) left the instruction order unchanged or did not finish compiling within 15 minutes.
// The kernel therefore stays OPT-IN (PINN_F64_HELPER=1, fused20d_plan) and k_fused20d remains the product path.
//
// Math and references: as kernels_fused20d.h (SURVEY Appendix A == the nested tapes of
// 1d-burgers/inf_cont_burgers.py:65-90 under utils/neuralnetwork.py:55-59; identification ide_cont_burgers.py:56-91).
#pragma once
#include "kernels_fused20d.h"

namespace pinn {

constexpr int DH_MAINS = 3, DH_TILE = 16 * DH_MAINS;            // 48 points per workgroup
#ifndef PINN_DH_HM
#define PINN_DH_HM 3                                            // in-groups (of 5) whose gradient blocks the helper takes
#endif
constexpr int DH_HM = PINN_DH_HM, DH_MM = 5 - DH_HM;            // helper: in-groups DH_MM .. 4; mains: 0 .. DH_MM-1 + biases
constexpr int DH_XV = 20 + 4 * DH_HM;                           // exchanged values per main lane and layer
// compact block lists: a main holds dense 0 (5), per hidden layer in-groups 0, 1 (10) + bias (5), dense H (6);
// the helper per hidden layer in-groups 2, 3, 4 (15)
constexpr int DH_MB = 5 * DH_MM + 5;                            // a main's blocks per hidden layer (+ 5 bias blocks)
constexpr int dh_main_blocks(int H) { return 5 + (H - 1) * DH_MB + 6; }
constexpr int dh_help_blocks(int H) { return (H - 1) * 5 * DH_HM; }
// LDS offset (in doubles) of entry e of k_fused20d's block list: >= 0 in a main's compact list, -(offset+1) in the helper's
inline void fused20dh_slot_table(int H, int* out) {
  const int NBLK = fused20d_blocks(H), BLK_H = 5 + (H - 1) * 30;
  for (int e = 0; e < NBLK * 16; ++e) {
    const int blk = e >> 4, w16 = e & 15;
    int slot, hslot = -1;
    if (blk < 5) slot = blk;
    else if (blk >= BLK_H) slot = 5 + (H - 1) * DH_MB + (blk - BLK_H);
    else {
      const int r = blk - 5, d1 = r / 30, mn = r - d1 * 30;
      if (mn >= 25) slot = 5 + d1 * DH_MB + 5 * DH_MM + (mn - 25);
      else {
        const int m = mn / 5, n = mn - 5 * m;
        if (m < DH_MM) slot = 5 + d1 * DH_MB + m * 5 + n;
        else { slot = 0; hslot = d1 * 5 * DH_HM + (m - DH_MM) * 5 + n; }
      }
    }
    out[e] = hslot >= 0 ? -(hslot * 16 + w16) - 1 : slot * 16 + w16;
  }
}
inline size_t fused20dh_lds_bytes(int n_hidden, int n_theta) {
  return (fused20d_weight_doubles(n_theta) + (size_t)(DH_MAINS * dh_main_blocks(n_hidden) + dh_help_blocks(n_hidden)) * 16 +
          (size_t)DH_MAINS * 4 * 64 /* loss slots */ + (size_t)DH_MAINS * DH_XV * 64 /* exchange */ + 16 /* flags */) * sizeof(double);
}
inline int fused20dh_tiles(int n_pad) { return (n_pad + DH_TILE - 1) / DH_TILE; }

template <int PDE, int H>
__global__ __launch_bounds__(256) void k_fused20dh(NetDesc nd, SetDesc sd, const double* __restrict__ th,
                                                   const double* __restrict__ xs, const double* __restrict__ ts,
                                                   const double* __restrict__ tgt, double lbx, double lbt, double sx,
                                                   double st, double nu, double* __restrict__ part, int R,
                                                   const int* __restrict__ row_index, long long* __restrict__ stamps) {
  constexpr int NBLK = fused20d_blocks(H);
  constexpr int BLK_H = 5 + (H - 1) * 30;
  constexpr int NMB = dh_main_blocks(H), NHB = dh_help_blocks(H);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* const wl = reinterpret_cast<double*>(lds_raw);
  const int nwp = (nd.n_theta + 127) / 128 * 128;
  double* const gm_all = wl + nwp;                               // [3][NMB * 16]
  double* const gh = gm_all + DH_MAINS * NMB * 16;               // [NHB * 16]
  double* const lacc_all = gh + NHB * 16;                        // [3][4][64]
  double* const xch = lacc_all + DH_MAINS * 4 * 64;              // [3][DH_XV][64]
  volatile int* const flags = reinterpret_cast<volatile int*>(xch + DH_MAINS * DH_XV * 64);   // ready[3], consumed

  STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, s = lane >> 4, i4 = lane & 3;
  const int pf = s * FW + i4, pr = i4 * FW + s;
  const int src = ((lane >> 2) | (lane << 4)) & 63;              // the lane whose value this lane holds after the rotation
  const int rot4 = src << 2;
  const int ge = s * 4 + i4;
  const bool helper = wave == DH_MAINS;

  const int pt = blockIdx.x * DH_TILE + wave * 16 + q;           // mains only
  double x = lbx, t = lbt;
  if (!helper && pt < sd.n_pad) { x = xs[pt]; t = ts[pt]; }

  for (int c = wave; c < nwp / 128; c += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(th + c * 128 + lane * 2),
                                     (__attribute__((address_space(3))) void*)(wl + c * 128), 16, 0, 0);
  for (int i = tid; i < DH_MAINS * 4 * 64; i += 256) lacc_all[i] = 0.0;
  if (tid < 4) flags[tid] = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STAMP(1);

  const double onesA = i4 == 0 ? 1.0 : 0.0;

  if (helper) {
    // ---------------------------------------------------------------- helper: blocks (m = 2, 3, 4) x n of every layer
#ifdef PINN_STAMPS
    for (int i = 2; i <= H + 1; ++i) STAMP(i);        // (the helper has no forward phases)
#endif
    // One step = (layer d, main i).  The operands of the NEXT step are requested before the matrix instructions of the
    // current one whenever its flag is already up (the three mains of a layer park their data at about the same time),
    // so the LDS latency is exposed once per layer, not three times.
    double zbT[2][4][5], inT[2][4][DH_HM];
    auto fetch = [&](const int i, const int buf) {
      const double* __restrict__ xb = xch + (i * DH_XV) * 64 + src;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int n = 0; n < 5; ++n) zbT[buf][c][n] = xb[(c * 5 + n) * 64];
#pragma unroll
        for (int mm = 0; mm < DH_HM; ++mm) inT[buf][c][mm] = xb[(20 + c * DH_HM + mm) * 64];
      }
    };
    bool pre = false;                                  // the current step's operands are already in flight / in registers
#pragma unroll
    for (int d = H - 1; d >= 1; --d) {
      double D[DH_HM][5];
#pragma unroll
      for (int mm = 0; mm < DH_HM; ++mm) {
#pragma unroll
        for (int n = 0; n < 5; ++n) D[mm][n] = 0.0;
      }
#pragma unroll
      for (int i = 0; i < DH_MAINS; ++i) {
        const int buf = (i + (H - 1 - d) * DH_MAINS) & 1;
        if (!pre) {
          if (d == 4 && i == 0) STAMP(20);
          while (flags[i] != d) __builtin_amdgcn_s_sleep(1);     // main i has parked layer d (flags count H-1 ... 1)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (d == 4 && i == 0) STAMP(21);
          fetch(i, buf);
        }
        if (d == 4 && i == 1) STAMP(22);
        if (d == 4 && i == 2) STAMP(23);
        pre = false;
        if (i + 1 < DH_MAINS) {                                  // next step: main i+1 of this layer
          if (flags[i + 1] == d) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            fetch(i + 1, buf ^ 1);
            pre = true;
          }
        }
        if (i == DH_MAINS - 1 || (i == DH_MAINS - 2 && pre)) {
          // every operand of this layer has been requested: once they have landed the mains may refill the buffer
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) flags[3] = d;
        }
#pragma unroll
        for (int mm = 0; mm < DH_HM; ++mm) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int n = 0; n < 5; ++n) D[mm][n] = mfma444(inT[buf][c][mm], zbT[buf][c][n], D[mm][n]);
          }
        }
      }
      if (d == 4) STAMP(24);
#pragma unroll
      for (int mm = 0; mm < DH_HM; ++mm) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          double v = D[mm][n];
          v += dpp_mov<DPP_ROW_ROR8>(v);
          v += dpp_mov<DPP_ROW_ROR4>(v);
          gh[((d - 1) * 5 * DH_HM + mm * 5 + n) * 16 + ge] = v;
        }
      }
      STAMP(2 * H + 1 - d);
    }
#ifdef PINN_STAMPS
    STAMP(2 * H);                                       // (no dense-0 phase either)
#endif
  } else {
    // ---------------------------------------------------------------- mains
    double* const gacc = gm_all + wave * (NMB * 16);
    double* const lacc = lacc_all + wave * 256 + lane;
    double* const xw = xch + (wave * DH_XV) * 64;                // this main's exchange slots [v][lane]
    auto grad_store = [&](double D, const int slot) {
      D += dpp_mov<DPP_ROW_ROR8>(D);
      D += dpp_mov<DPP_ROW_ROR4>(D);
      gacc[slot * 16 + ge] = D;
    };
    double c1 = 1.0, c2 = nu;
    if (PDE == 1) { c1 = wl[nd.n_net]; c2 = exp(wl[nd.n_net + 1]); }
    const double inv_nf = sd.inv_nf, inv_nu = sd.inv_nu;
    const double hx = __builtin_fma(sx, x - lbx, -1.0), ht = __builtin_fma(st, t - lbt, -1.0);

    // ------------------------------------------------------------------ forward (as k_fused20d)
    double in[4][5], a0[5];
    agd stash[H][5][4];
    double top[5][4];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      const int f = 4 * n + s;
      const double w0x = wl[nd.off_w[0] + f], w0t = wl[nd.off_w[0] + FW + f], b0 = wl[nd.off_b[0] + f];
      const double a = tanh_d(__builtin_fma(hx, w0x, __builtin_fma(ht, w0t, b0)));
      a0[n] = a;
      channels_d(a, sx * w0x, st * w0t, 0.0, in[0][n], in[1][n], in[2][n], in[3][n]);
    }
#pragma unroll
    for (int d = 1; d < H; ++d) {
      const double* __restrict__ wd = wl + nd.off_w[d] + pf;
      double acc[4][5];
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        acc[0][n] = wl[nd.off_b[d] + 4 * n + s];
        acc[1][n] = acc[2][n] = acc[3][n] = 0.0;
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          const double A = wd[80 * m + 4 * n];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c][n] = mfma444(A, in[c][m], acc[c][n]);
        }
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const double a = tanh_d(acc[0][n]);
        channels_d(a, acc[1][n], acc[2][n], acc[3][n], in[0][n], in[1][n], in[2][n], in[3][n]);
        if (d < H - 1) {
          stash[d][n][0] = agd_put(a); stash[d][n][1] = agd_put_after(acc[1][n], in[1][n]);
          stash[d][n][2] = agd_put_after(acc[2][n], in[2][n]); stash[d][n][3] = agd_put_after(acc[3][n], in[3][n]);
        } else {
          top[n][0] = a; top[n][1] = acc[1][n]; top[n][2] = acc[2][n]; top[n][3] = acc[3][n];
        }
      }
      STAMP(1 + d);
    }
    double o[4] = {wl[nd.off_b[H]], 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const double A = wl[nd.off_w[H] + 4 * m + s];
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = mfma444(A, in[c][m], o[c]);
    }

    // ------------------------------------------------------------------ seeds + loss parts
    double sb[4] = {0.0, 0.0, 0.0, 0.0};
    {
      const int cls = point_class(sd, pt);
      const bool res = (PDE == 0) ? (cls == CLS_COL) : (cls == CLS_DATA);
      if (res) {
        const double wgt = (PDE == 0) ? inv_nf : inv_nu;
        const double f = o[2] + c1 * o[0] * o[1] - c2 * o[3];
        const double fbar = 2.0 * f * wgt;
        if (s == 0) {
          lacc[0] += f * f * wgt;
          if (PDE == 1) { lacc[128] += fbar * o[0] * o[1]; lacc[192] -= fbar * c2 * o[3]; }
        }
        sb[0] = fbar * c1 * o[1]; sb[1] = fbar * c1 * o[0]; sb[2] = fbar; sb[3] = -c2 * fbar;
      }
      if (cls == CLS_DATA) {
        const double dd = o[0] - tgt[pt];
        if (s == 0) lacc[64] += dd * dd * inv_nu;
        sb[0] += 2.0 * dd * inv_nu;
      }
    }

    // ------------------------------------------------------------------ reverse sweep
    double ob[4][5];
    {  // dense H (linear, one output): all six blocks stay with the main
      double sbT[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) sbT[c] = lane_fetch(s == 0 ? sb[c] : 0.0, rot4);
      double D[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int m = 0; m < 5; ++m) D[m] = mfma444(lane_fetch(in[c][m], rot4), sbT[c], D[m]);
      }
      D[5] = mfma444(onesA, sbT[0], 0.0);
#pragma unroll
      for (int m = 0; m < 6; ++m) grad_store(D[m], 5 + (H - 1) * DH_MB + m);
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const double w = wl[nd.off_w[H] + 4 * n + s];
#pragma unroll
        for (int c = 0; c < 4; ++c) ob[c][n] = sb[c] * w;
      }
    }
    STAMP(H + 1);
#pragma unroll
    for (int d = H - 1; d >= 1; --d) {
      double zb[4][5];
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        double a, zp, zq, zr;
        if (d == H - 1) { a = top[n][0]; zp = top[n][1]; zq = top[n][2]; zr = top[n][3]; }
        else { a = agd_get(stash[d][n][0]); zp = agd_get(stash[d][n][1]); zq = agd_get(stash[d][n][2]); zr = agd_get(stash[d][n][3]); }
        preact_adjoint_d(a, zp, zq, zr, ob[0][n], ob[1][n], ob[2][n], ob[3][n], zb[0][n], zb[1][n], zb[2][n], zb[3][n]);
      }
      // output channels of layer d-1, all five in-groups: 0, 1 for this wave's own blocks, 2..4 for the helper
      double inc[4][5];
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        double a_, zp_, zq_, zr_;
        if (d - 1 == 0) {
          const int f_ = 4 * m + s;
          a_ = a0[m]; zp_ = sx * wl[nd.off_w[0] + f_]; zq_ = st * wl[nd.off_w[0] + FW + f_]; zr_ = 0.0;
        } else {
          a_ = agd_get(stash[d - 1][m][0]); zp_ = agd_get(stash[d - 1][m][1]);
          zq_ = agd_get(stash[d - 1][m][2]); zr_ = agd_get(stash[d - 1][m][3]);
        }
        channels_d(a_, zp_, zq_, zr_, inc[0][m], inc[1][m], inc[2][m], inc[3][m]);
      }
      if (d == 4) STAMP(20);
      // the exchange buffer still holds layer d+1 until the helper has it in registers
      if (d < H - 1) {
        while (flags[3] != d + 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      if (d == 4) STAMP(21);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int n = 0; n < 5; ++n) xw[(c * 5 + n) * 64 + lane] = zb[c][n];
#pragma unroll
        for (int mm = 0; mm < DH_HM; ++mm) xw[(20 + c * DH_HM + mm) * 64 + lane] = inc[c][DH_MM + mm];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) flags[wave] = d;
      if (d == 4) STAMP(22);
      // own rotated adjoints: read back through the rotation (the slots this wave has just written)
      double zbT[4][5];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int n = 0; n < 5; ++n) zbT[c][n] = xw[(c * 5 + n) * 64 + src];
      }
      if (d == 4) STAMP(23);
      // own gradient blocks: in-groups 0, 1 and the biases (before the adjoint GEMV: their operands die here, which
      // keeps the wave inside its 256 VGPRs)
      const int base = 5 + (d - 1) * DH_MB;
#pragma unroll
      for (int m = 0; m < DH_MM; ++m) {
        double cur[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = lane_fetch(inc[c][m], rot4);
        double D[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int n = 0; n < 5; ++n) D[n] = mfma444(cur[c], zbT[c][n], D[n]);
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) grad_store(D[n], base + m * 5 + n);
      }
      {
        double D[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) D[n] = mfma444(onesA, zbT[0][n], 0.0);
#pragma unroll
        for (int n = 0; n < 5; ++n) grad_store(D[n], base + 5 * DH_MM + n);
      }
      if (d == 4) STAMP(24);
      // adjoint of the layer-(d-1) outputs
      const double* __restrict__ wd = wl + nd.off_w[d] + pr;
#pragma unroll
      for (int m = 0; m < 5; ++m) ob[0][m] = ob[1][m] = ob[2][m] = ob[3][m] = 0.0;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          const double A = wd[80 * m + 4 * n];
#pragma unroll
          for (int c = 0; c < 4; ++c) ob[c][m] = mfma444(A, zb[c][n], ob[c][m]);
        }
      }
      STAMP(2 * H + 1 - d);
    }
    {  // dense 0
      const double hxT = lane_fetch(hx, rot4), htT = lane_fetch(ht, rot4);
      const double Ah = i4 == 0 ? hxT : i4 == 1 ? htT : i4 == 2 ? 1.0 : 0.0;
      const double Ap = i4 == 0 ? sx : 0.0, Aq = i4 == 1 ? st : 0.0;
      double bT[3][5];
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const int f = 4 * n + s;
        double bh, bp, bq, br;
        preact_adjoint_d(a0[n], sx * wl[nd.off_w[0] + f], st * wl[nd.off_w[0] + FW + f], 0.0, ob[0][n], ob[1][n],
                         ob[2][n], ob[3][n], bh, bp, bq, br);
        bT[0][n] = lane_fetch(bh, rot4); bT[1][n] = lane_fetch(bp, rot4); bT[2][n] = lane_fetch(bq, rot4);
      }
      double D[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) D[n] = mfma444(Ah, bT[0][n], 0.0);
#pragma unroll
      for (int n = 0; n < 5; ++n) D[n] = mfma444(Ap, bT[1][n], D[n]);
#pragma unroll
      for (int n = 0; n < 5; ++n) D[n] = mfma444(Aq, bT[2][n], D[n]);
#pragma unroll
      for (int n = 0; n < 5; ++n) grad_store(D[n], n);
    }
  }
  STAMP(2 * H + 1);

  // -------------------------------------------------------------------- one gradient row per workgroup
  {
    constexpr int NE = NBLK * 16, NIT = (NE + 255) / 256;
    int idx[NIT], off[NIT];                            // parameter index / LDS offset of entry e (host tables, fetched early)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 256 * it;
      idx[it] = e < NE ? row_index[e] : -1;
      off[it] = e < NE ? row_index[NE + e] : 0;
    }
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (!helper) {
      const double* const lacc = lacc_all + wave * 256 + lane;
      t0 = wave_sum(lacc[0]); t1 = wave_sum(lacc[64]);
      t2 = PDE == 1 ? wave_sum(lacc[128]) : 0.0; t3 = PDE == 1 ? wave_sum(lacc[192]) : 0.0;
    }
    __syncthreads();                                   // every block of every wave is stored
    double* const scal = wl;                           // the weight copy is dead: 3 x 4 loss / lambda partials
    if (lane == 0 && !helper) { scal[wave * 4 + 0] = t0; scal[wave * 4 + 1] = t1; scal[wave * 4 + 2] = t2; scal[wave * 4 + 3] = t3; }
    __syncthreads();
    double* __restrict__ row = part + (size_t)blockIdx.x * R;
    double v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int o = off[it];
      const int om = o >= 0 ? o : 0, oh = o < 0 ? -o - 1 : 0;
      const double vm = (gm_all[om] + gm_all[NMB * 16 + om]) + gm_all[2 * NMB * 16 + om];
      const double vh = gh[oh];
      v[it] = o >= 0 ? vm : vh;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (idx[it] >= 0) row[idx[it]] = v[it];
    if (tid < 4) {
      const double v = (scal[tid] + scal[4 + tid]) + scal[8 + tid];
      if (tid == 0) { row[nd.n_theta + 0] = v; row[nd.n_theta + 2] = 0.0; }
      if (tid == 1) row[nd.n_theta + 1] = v;
      if (PDE == 1 && tid == 2) row[nd.n_net] = v;
      if (PDE == 1 && tid == 3) row[nd.n_net + 1] = v;
    }
  }
  STAMP(2 * H + 2);
}

// returns a hipError_t (0 = ok); n_wg = fused20dh_tiles(sd.n_pad) workgroups, one partial row each; row_index holds
// fused20d_row_index (NBLK x 16 entries) followed by fused20dh_slot_table (the same count)
template <int PDE, int H>
inline int fused20dh_launch(const NetDesc& nd, const SetDesc& sd, const double* th, const double* xs, const double* ts,
                            const double* tgt, double lbx, double lbt, double sx, double st, double nu, double* part,
                            int R, const int* row_index, hipStream_t stream, long long* stamps = nullptr,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  const size_t lds = fused20dh_lds_bytes(H, nd.n_theta);
  static unsigned long long attr_set = 0;
  if (first_call_on_device(attr_set)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_fused20dh<PDE, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_wg = fused20dh_tiles(sd.n_pad);
  if (ev_start && ev_stop)
    hipExtLaunchKernelGGL((k_fused20dh<PDE, H>), dim3(n_wg), dim3(256), lds, stream, ev_start, ev_stop, 0, nd, sd, th, xs,
                          ts, tgt, lbx, lbt, sx, st, nu, part, R, row_index, stamps);
  else
    hipLaunchKernelGGL((k_fused20dh<PDE, H>), dim3(n_wg), dim3(256), lds, stream, nd, sd, th, xs, ts, tgt, lbx, lbt, sx,
                       st, nu, part, R, row_index, stamps);
  return (int)hipGetLastError();
}

}  // namespace pinn
