// kernels_fused20d.h -- float64 loss+gradient kernel for width-20 tanh MLPs (k_fused20d): the reference's own
// arithmetic (utils/neuralnetwork.py:24-26 is float64) with every contraction on v_mfma_f64_4x4x4_4b_f64 and no
// inter-wave exchange at all.
//
// Why this shape.  Measured on gfx950 (profiles/r02_ubench_mfma_f64_4x4x4.txt): the instruction retires 4 blocks x
// (4x4x4) = 256 MACs in 16.3 cycles = the FP64 rate of the SIMD (a v_fma_f64 costs 4-5 cycles for 64 MACs; the two
// share the pipe, their times add), dependent-accumulator latency 20 cycles, and its lane maps are
//     A[i][k] : lane 16k + 4b + i      B[k][j] : lane 16k + 4b + j      D[i][j] : lane 16i + 4b + j     (b = block)
// i.e. D and B have the SAME map with the output row i in the place of the contraction index k.  Put a point on the
// low four lane bits (q = 4b + j: 16 points per wave) and a feature slot s on the two high bits, five registers per
// Taylor channel for the five groups of four features (feature = 4n + s), and a layer is
//     out_c[n] += W-pattern[m][n] (A)  x  in_c[m] (B)            25 instructions per channel, 20 = 5 x 4: no padding
// whose result registers ARE the next layer's B operands: lane = (slot, point) in, lane = (slot, point) out, for the
// forward GEMV and (with the transposed pattern) for the reverse GEMV.  No LDS exchange tile and no
// cross-lane traffic in either sweep; tanh and the Taylor / adjoint algebra are lane-local.  The weight patterns
// are plain ds_read_b64 of the flat weight vector (copied into LDS once per workgroup): lane (s, i = lane & 3) reads
// W[4m+s][4n+i] (forward) or W[4m+i][4n+s] (reverse) -- conflict-free, no packed image needed.
//
// The weight gradient dW_d[k][j] = sum over (point, channel) IN_c[k] ZBAR_c[j] contracts over POINTS, so both
// operands need point bits where the instruction contracts (lane bits 4-5) and the feature slot on bits 0-1.  Since
// round 6 that move is ONE matrix instruction with the 4 x 4 identity as B -- it swaps the slot field with the two LOW
// point bits (PINN_TO_POINTS below; rounds 2-5: a ds_bpermute pair per value, 642 per tile on the LDS pipe), 40 values
// per layer.  The four blocks then hold partial sums over the points with the same HIGH point bits.
//   tile loop (more tiles than workgroups): the blocks are folded with two DPP row rotations and added into a per-wave
//     accumulator in LDS (221 blocks x 16 values = 28 KB per wave, persistent over the workgroup's tiles; summed over the
//     four waves in fixed order at the end);
//   one tile per workgroup: nothing is accumulated -- every lane parks its partial in a double-buffered staging area and
//     the whole workgroup adds the 4 waves x 4 blocks of every entry once per reverse layer, straight into the gradient row.
// Either way one gradient row per workgroup, no atomics, bit-reproducible.  First and last dense layer run through the
// same block machinery (in-group (h_x, h_t, 1) / single output column), so there is no per-lane gradient bookkeeping.
//
// Stash: (a, z_x, z_t, z_xx) of the 5 own features x 6 middle layers = 240 registers parked in AGPRs; layer 0 keeps
// only a (its other channels are weight constants), the last hidden layer stays live in VGPRs.  One wave per SIMD,
// four waves = 64 points per workgroup, persistent over tiles (grid = min(tiles, CUs)).
//
// Work per 64-point tile and hidden layer: 100 (forward) + 100 (reverse GEMV) + 105 (dW) + 40 (operand moves) matrix
// instructions per wave = 5.6k cycles of matrix pipe (305 of them algorithmic); algorithmic FLOP and bytes as k_fused20m (68 640 FLOP and 16 B per collocation point).
//
// Math: SURVEY.md Appendix A.1-A.3 == nested GradientTapes of 1d-burgers/inf_cont_burgers.py:65-90 under the outer
// tape of utils/neuralnetwork.py:55-59; identification (PDE == 1): 1d-burgers/ide_cont_burgers.py:56-91.
#pragma once
#include <hip/hip_ext.h>
#include "kernels_fused20.h"
#include "fused20d_api.h"

// 1: the tile loop re-reads the lane index through an opaque asm once per tile (what stopped the address hoisting of
// k_wide_bwd / k_t16_fused).  Here it takes the tile-loop variants from 256 VGPRs + 6-11 AGPR spill slots to 238-242 / 0
// and is 1 % SLOWER (same-box A/B, N_f = 10^6: 1991 / 2000 vs 1971 / 1978 us per Adam step): off.
#ifndef PINN_PATTERN_AHEAD
#define PINN_PATTERN_AHEAD 1     // the GEMV loops pinned step by step (sched_barrier), weight patterns requested in pairs two
                                 // steps ahead (0: hipcc's placement -- it sinks every ds_read next to its consumer).  Round 4:
                                 // 40.8 -> 40.5 us per Adam step at N_f = 10^4, and 1.6 % SLOWER in the tile-loop variants
                                 // (PINN_PA_LOOP below turned that round 6)
#endif
#ifndef PINN_ROT_IN_GEMV
#define PINN_ROT_IN_GEMV 2       // where a reverse layer's 40 operand moves (PINN_TO_POINTS) and the one-tile phase sum stand:
                                 //  0  moves in front of the sum, the whole sum in front of the GEMV
                                 //  1  moves inside the GEMV (one per pinned step), barrier + reads in front of it, adds behind
                                 //  2  as 1, barrier + reads in front of the adjoint arithmetic (pure VALU) as well
                                 // same-box: 36.9 / 35.6 / 35.0 us with the ds_bpermute moves (profiles/r06_ab_onetile_v3, _v4)
#endif
#ifndef PINN_FOLD_STAGES
#define PINN_FOLD_STAGES 1       // tile loop: DPP fold stages in front of the ds_add_f64.  1 = the blocks are folded once (b with
                                 // b ^ 2) and the lanes of blocks 0 and 1 add into the accumulator -- two lanes per address in one
                                 // LDS instruction (resolved in lane order: run-to-run bit equality is asserted by the tile-loop
                                 // tests).  Same box, N_f = 10^6: 2 stages 1778 us, 1 stage 1738 us, 0 stages (four lanes per
                                 // address) 2110 us (profiles/r06_ab_foldstages.txt)
#endif
#ifndef PINN_GACC_ATOMIC
#define PINN_GACC_ATOMIC 1       // tile loop: gradient accumulators updated by ds_add_f64 from the lanes of block 0 (0: read-add-write)
#endif
#ifndef PINN_PA_LOOP
#define PINN_PA_LOOP 1           // the pinned GEMV loops in the tile-loop variants too: N_f = 10^6 1868 -> 1851 us, same box
                                 // (profiles/r06_ab_loopvariants.txt; with the ds_bpermute rotations of rounds 2-5 this was a loss)
#endif
#ifndef PINN_OPAQUE_TILE_D
#define PINN_OPAQUE_TILE_D 0
#endif

// finer timeline inside reverse layer 4 and forward layer 4 (slots 20..30 of the wave's 32), profiling build -DPINN_STAMPS2 only
#if defined(PINN_STAMPS) && defined(PINN_STAMPS2)
#define STAMP2(cond, i) do { if (cond) STAMP(i); } while (0)
#else
#define STAMP2(cond, i) do { } while (0)
#endif

namespace pinn {

// (Ablation builds -- one ingredient compiled out at a time, wrong results by construction, only times are read -- are not part
// of the product sources since round 5: `git apply -R profiles/ablation_scaffolding.patch` puts the -DPINN_ABL / -DPINN_ABLD /
// -DT16_ABL switches back for profiles/ablate_*.py; their results are under profiles/*ablate*.txt.)

// a double parked in the accumulation half of the register file (two 32-bit AGPRs)
struct agd { int lo, hi; };
__device__ __forceinline__ agd agd_put(const double x) {
  agd a;
  const int lo = __double2loint(x), hi = __double2hiint(x);
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.lo) : "v"(lo));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.hi) : "v"(hi));
  return a;
}
// The same for a value that comes straight out of a matrix instruction.  hipcc's hazard recogniser does not look
// inside inline asm, so nothing would keep the v_accvgpr_write the required wait states behind the MFMA that
// produces x (observed: stale low words, 1e-8 relative errors).  `after` must be the result of a compiler-visible
// VALU instruction that reads x: the extra operand orders the asm behind that instruction, whose own hazard wait
// the compiler did insert.  tests/helpers/isa_lint.py checks the built code for exactly this (every accvgpr move of
// the library against every matrix instruction in front of it, along the control-flow graph).
__device__ __forceinline__ agd agd_put_after(const double x, const double after) {
  agd a;
  const int lo = __double2loint(x), hi = __double2hiint(x), dep = __double2hiint(after);
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.lo) : "v"(lo), "v"(dep));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.hi) : "v"(hi), "v"(dep));
  return a;
}
__device__ __forceinline__ double agd_get(const agd a) {
  int lo, hi;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a.lo));
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a.hi));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double mfma444(const double a, const double b, const double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// value of lane (src4 >> 2) -- any permutation of the wave, 2 x ds_bpermute_b32
__device__ __forceinline__ double lane_fetch(const double x, const int src4) {
  const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(x));
  return __hiloint2double(hi, lo);
}

constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

// The weight-gradient blocks contract over POINTS, so both their operands need the point index where the matrix instruction
// contracts (lane bits 4-5) and the feature slot on bits 0-1.  Rounds 2-5 moved every value there with a ds_bpermute pair
// (a rotation of the lane index by two bits): 642 of them per tile, ~31 cycles of the LDS pipe each with four waves on the
// CU -- a quarter of the kernel's time (profiles/r06_stamps2_new.txt).  The matrix instruction itself can do the move: with
// the 4 x 4 identity as B (lane 16k + 4b + j holds [k == j]),
//     D[i][j] = sum_k A[i][k] [k == j] = A[i][j]:   lane 16i + 4b + j  <-  lane 16j + 4b + i,
// i.e. the slot field and the low two point bits of the lane index change places inside every block -- exactly a valid
// operand layout for the gradient blocks (contraction over the low point bits, one block per high-point-bit value; both
// operands and the "ones" / (hx, ht, 1) patterns of the bias and first-layer blocks use the same convention).  Products
// with 1.0 and sums with zeros are exact, so the moved value is bit-identical to the bpermute's.  One instruction of the
// matrix pipe (16 cycles, no LDS round trip, no address register) instead of two of the LDS pipe.
#ifndef PINN_ROT_MFMA
#define PINN_ROT_MFMA 1          // 0: the ds_bpermute pair
#endif

// tanh(x) = sign(x) (1 - t) / (1 + t), t = e^{-2|x|}: the denominator lies in (1, 2], so the quotient needs none of
// the scaling / fix-up of an IEEE division: v_rcp_f64 seed + two Newton steps (relative error < 1e-30 before the
// final rounding), 5 instructions instead of 11.
__device__ __forceinline__ double tanh_d(const double x) {
  const double t = exp(-2.0 * fabs(x));
  const double y = 1.0 + t;
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  const double num = 1.0 - t;
  double qv = num * r;
  qv = __builtin_fma(__builtin_fma(-y, qv, num), r, qv);      // one correction of the quotient itself
  return copysign(qv, x);
}

// layer-output channels (h, p, q, r) of a stash entry (a, zp, zq, zr)           (A.1)
__device__ __forceinline__ void channels_d(const double a, const double zp, const double zq, const double zr,
                                           double& h, double& p, double& q, double& r) {
  const double d1 = __builtin_fma(-a, a, 1.0);
  const double t = (-2.0 * a) * zp;
  h = a; p = d1 * zp; q = d1 * zq; r = d1 * __builtin_fma(t, zp, zr);
}

// adjoint of the pre-activation channels                                          (A.3)
__device__ __forceinline__ void preact_adjoint_d(const double a, const double zp, const double zq, const double zr,
                                                 const double oh, const double op, const double oq, const double orr,
                                                 double& bh, double& bp, double& bq, double& br) {
  const double a2 = a * a, d1 = 1.0 - a2;
  const double d2 = (-2.0 * a) * d1;
  const double d3 = (-2.0 * d1) * __builtin_fma(-3.0, a2, 1.0);
  const double zpw = zp * orr;
  const double dot = __builtin_fma(zr, orr, __builtin_fma(zq, oq, zp * op));
  bh = __builtin_fma(d3 * zp, zpw, __builtin_fma(d2, dot, d1 * oh));
  bp = __builtin_fma(d2 + d2, zpw, d1 * op);
  bq = d1 * oq;
  br = d1 * orr;
}

// ONE_TILE: the launch has at least as many workgroups as tiles: every gradient block is produced exactly once, so
// it is stored, not accumulated (no LDS read-modify-write), and there is no loop-carried coordinate prefetch.
template <int PDE, int H, bool ONE_TILE>
__global__ __launch_bounds__(256) void k_fused20d(const double* __restrict__ th, const double* __restrict__ xs,
                                                  const double* __restrict__ ts, const double* __restrict__ tgt,
                                                  double* __restrict__ part, const int* __restrict__ row_index, int R,
                                                  int n_tiles, double lbx, double lbt, double sx, double st, double nu,
                                                  SetDesc sd, long long* __restrict__ stamps, W20Desc nd_arg) {
  // weight offsets: compile-time constants in the one-tile variant (immediate operands; Adam step 41.9 -> 40.8 us with
  // the preloaded pointers); the tile-loop variant keeps them in SGPRs -- with immediates its schedule came out 9 %
  // slower (N_f = 10^6: 2104 vs 1930 us per step, same box)
  constexpr W20Desc nd_const = w20_desc(H, PDE == 1);
  const W20Desc nd = ONE_TILE ? nd_const : nd_arg;
  constexpr int NBLK = fused20d_blocks(H);
  constexpr bool PA = PINN_PATTERN_AHEAD && (ONE_TILE || PINN_PA_LOOP);   // the pinned GEMV loops (patterns two steps ahead)
  constexpr int BLK_H = 5 + (H - 1) * 30;            // first block of dense H
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* const wl = reinterpret_cast<double*>(lds_raw);
  const int nwp = (nd.n_theta + 127) / 128 * 128;
  double* const gacc_all = wl + nwp;                  // tile loop: 4 x NBLK x 16 accumulators; one tile: 2 staging buffers
  double* const lacc_all = gacc_all + (ONE_TILE ? 2 * FUSED20D_STAGE_BUF : 4 * NBLK * 16);

#if PINN_ROT_MFMA
#define PINN_TO_POINTS(X) mfma444((X), ident, 0.0)
#else
#define PINN_TO_POINTS(X) lane_fetch((X), rot4)
#endif
  STAMP(0);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // per-lane indices as a macro: the tile loop re-derives them from an opaque copy of the lane index once per tile
  // (PINN_OPAQUE_LANE), so that hipcc does not carry the ~30 per-lane LDS addresses they feed across the loop
#define PINN_LANE_INDICES(L)                                                                                          \
  const int lane = (L);                                                                                                \
  const int q = lane & 15;                 /* point of this wave's 16 */                                              \
  const int s = lane >> 4;                 /* feature slot: feature = 4 * group + s */                                \
  const int i4 = lane & 3;                 /* row / column slot of the A patterns and of the gradient blocks */       \
  const int pf = s * FW + i4;              /* forward pattern:  W[4m + s][4n + i4] */                                 \
  const int pr = i4 * FW + s;              /* reverse pattern:  W[4m + i4][4n + s] */                                 \
  const int rot4 = (((lane >> 2) | (lane << 4)) & 63) << 2;   /* lane-index rotation by two bits (bpermute address) */ \
  const double ident = s == i4 ? 1.0 : 0.0;                    /* 4 x 4 identity as a B operand: the transposing matrix instruction */ \
  const int ge = s * 4 + i4;               /* this lane's entry (i, j) of a gradient block */                         \
  double* const lacc = lacc_all + wave * 256 + lane;                        /* [k * 64]: l_res, l_dat, dl0, dl1 */     \
  const int sput = (s * 4 + i4) * 4 + ((lane >> 2) & 3);   /* one-tile staging slot: entry-major, the four blocks of an entry adjacent */ \
  const double onesA = i4 == 0 ? 1.0 : 0.0;                     /* rotated "ones" in-group: row 0 = 1 (bias gradients) */ \
  (void)q; (void)pf; (void)pr; (void)rot4; (void)ident; (void)ge; (void)lacc; (void)onesA; (void)s; (void)sput
  PINN_LANE_INDICES(tid & 63);
  double* const gacc = gacc_all + wave * (NBLK * 16);

  // first tile's coordinates: issued ahead of the weight staging, so the two round trips overlap
  int tile = blockIdx.x;
  double x = 0.0, t = 0.0;
  if (tile < n_tiles) { x = xs[tile * 64 + wave * 16 + q]; t = ts[tile * 64 + wave * 16 + q]; }

  // ---- flat weight vector -> LDS by asynchronous LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no
  // registers; the engine pads the vector's allocation to whole pieces), gradient accumulators <- 0 meanwhile
  for (int c = wave; c < nwp / 128; c += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(th + c * 128 + lane * 2),
                                     (__attribute__((address_space(3))) void*)(wl + c * 128), 16, 0, 0);
  {
    typedef double d2 __attribute__((ext_vector_type(2)));
    if (ONE_TILE) {                                    // the staging buffers are written before they are read: loss parts only
      d2* const z = reinterpret_cast<d2*>(lacc_all);
      for (int i = tid; i < 2 * 256; i += 256) z[i] = d2{0.0, 0.0};
    } else {
      d2* const z = reinterpret_cast<d2*>(gacc_all);
      for (int i = tid; i < 2 * NBLK * 16 + 2 * 256; i += 256) z[i] = d2{0.0, 0.0};   // + the loss-part slots behind them
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  double c1 = 1.0, c2 = nu;
  if (PDE == 1) { c1 = wl[nd.n_net]; c2 = exp(wl[nd.n_net + 1]); }
  const double inv_nf = sd.inv_nf, inv_nu = sd.inv_nu;
  // per-lane partial sums of the loss parts and of the two lambda gradients (slot-0 lanes only) live in LDS, four
  // slots per lane behind the gradient accumulators: one read-modify-write per tile instead of eight registers held
  // across the whole kernel (which cost the identification variant 20 B of scratch per lane)

  // Gradient blocks.  D = this lane's partial sum over the four points of its block b; the four blocks of an entry
  // are folded with two DPP row rotations -- commutative, so the four lanes of an entry end with bit-identical
  // totals and may all store (same value, same address: no exec masking, no branch).  The old accumulator values
  // are fetched BEFORE the matrix instructions that produce D (grad_fetch), so no LDS round trip is exposed.
  // (Tried: ds_add_f64 with the four lanes of an entry hitting one address, no fold, no read-modify-write --
  //  221 instructions instead of ~3000, bit-reproducible over 200 runs, and 14 % SLOWER: 49.8 vs 43.7 us per step.)

  STAMP(1);
#ifdef PINN_STAGGER
  // experiment: the four waves of a workgroup run the same instruction stream in step and meet at the LDS pipe; wave w starts
  // w x PINN_STAGGER x 64 cycles late
  if (wave == 1) __builtin_amdgcn_s_sleep(PINN_STAGGER);
  if (wave == 2) { __builtin_amdgcn_s_sleep(PINN_STAGGER); __builtin_amdgcn_s_sleep(PINN_STAGGER); }
  if (wave == 3) { __builtin_amdgcn_s_sleep(PINN_STAGGER); __builtin_amdgcn_s_sleep(PINN_STAGGER); __builtin_amdgcn_s_sleep(PINN_STAGGER); }
#endif

  for (; tile < n_tiles; tile += gridDim.x) {
    // (PINN_OPAQUE_TILE_D, off by default: see the macro)
    int lane_o = tid & 63;
    if (!ONE_TILE && PINN_OPAQUE_TILE_D) asm volatile("" : "+v"(lane_o));
    PINN_LANE_INDICES(lane_o);
    auto grad_fetch = [&](const int blk) { return (ONE_TILE || PINN_GACC_ATOMIC) ? 0.0 : gacc[blk * 16 + ge]; };
    // One tile per workgroup: nothing is accumulated, so the four blocks are not folded in registers (2 x 2 DPP moves + 2
    // adds per block, 1 300 instructions per tile): every lane parks its own partial in the phase's staging buffer
    // (entry-major: the four blocks of an entry adjacent), and phase_sum adds blocks and waves with the whole workgroup.
    int phase_first = 0;
    double* stage_w = gacc_all + wave * FUSED20D_STAGE_WAVE;
    // Tile loop (PINN_GACC_ATOMIC): the folded total sits in all four lanes of an entry; the lanes of block 0 add it into the
    // wave's accumulator with ONE ds_add_f64 (16 distinct addresses, one adder per address and tile: no conflict,
    // bit-reproducible) instead of a ds_read of the old value long before, a v_add_f64 and a ds_write.  A group's stores
    // share one hand-set execution-mask region: as `if (block 0) atomic` every store became its own basic block (two scalar
    // instructions each, and the stash reads the compiler shares between a layer's rotated inputs and the next layer's
    // adjoints were issued twice: 9 233 -> 9 766 instructions).  LDS operations the compiler does not see only make its own
    // lgkmcnt waits conservative (the queue is in order).  (The UNFOLDED form -- four lanes per address -- was 14 % slower
    // in round 2; this one follows the fold.)
    double pend_D[6] = {0, 0, 0, 0, 0, 0};
    int pend_off[6] = {0, 0, 0, 0, 0, 0}, pend_n = 0;
    auto gacc_flush = [&]() {
#if PINN_GACC_ATOMIC
      if constexpr (!ONE_TILE) {
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(gacc + ge);
        unsigned long long saved;
        if (pend_n == 6)
          asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[mk]\n\t"
                       "ds_add_f64 %[a], %[d0] offset:%[o0]\n\tds_add_f64 %[a], %[d1] offset:%[o1]\n\t"
                       "ds_add_f64 %[a], %[d2] offset:%[o2]\n\tds_add_f64 %[a], %[d3] offset:%[o3]\n\t"
                       "ds_add_f64 %[a], %[d4] offset:%[o4]\n\tds_add_f64 %[a], %[d5] offset:%[o5]\n\t"
                       "s_mov_b64 exec, %[sv]"
                       : [sv] "=&s"(saved)
                       : [a] "v"(addr), [mk] "s"(PINN_FOLD_STAGES == 2 ? 0x000f000f000f000full : PINN_FOLD_STAGES == 1 ? 0x00ff00ff00ff00ffull : 0xffffffffffffffffull), [d0] "v"(pend_D[0]), [d1] "v"(pend_D[1]),
                         [d2] "v"(pend_D[2]), [d3] "v"(pend_D[3]), [d4] "v"(pend_D[4]), [d5] "v"(pend_D[5]),
                         [o0] "i"(pend_off[0]), [o1] "i"(pend_off[1]), [o2] "i"(pend_off[2]), [o3] "i"(pend_off[3]),
                         [o4] "i"(pend_off[4]), [o5] "i"(pend_off[5])
                       : "memory");
        else
          asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[mk]\n\t"
                       "ds_add_f64 %[a], %[d0] offset:%[o0]\n\tds_add_f64 %[a], %[d1] offset:%[o1]\n\t"
                       "ds_add_f64 %[a], %[d2] offset:%[o2]\n\tds_add_f64 %[a], %[d3] offset:%[o3]\n\t"
                       "ds_add_f64 %[a], %[d4] offset:%[o4]\n\t"
                       "s_mov_b64 exec, %[sv]"
                       : [sv] "=&s"(saved)
                       : [a] "v"(addr), [mk] "s"(PINN_FOLD_STAGES == 2 ? 0x000f000f000f000full : PINN_FOLD_STAGES == 1 ? 0x00ff00ff00ff00ffull : 0xffffffffffffffffull), [d0] "v"(pend_D[0]), [d1] "v"(pend_D[1]),
                         [d2] "v"(pend_D[2]), [d3] "v"(pend_D[3]), [d4] "v"(pend_D[4]),
                         [o0] "i"(pend_off[0]), [o1] "i"(pend_off[1]), [o2] "i"(pend_off[2]), [o3] "i"(pend_off[3]),
                         [o4] "i"(pend_off[4])
                       : "memory");
      }
#endif
      pend_n = 0;
    };
    auto grad_store = [&](double D, const double old, const int blk) {
      if (ONE_TILE) { stage_w[(blk - phase_first) * 64 + sput] = D; return; }
#if PINN_GACC_ATOMIC && PINN_FOLD_STAGES < 2
      if (PINN_FOLD_STAGES == 1) D += dpp_mov<DPP_ROW_ROR8>(D);       // the LDS adder does the rest of the fold
#else
      D += dpp_mov<DPP_ROW_ROR8>(D);
      D += dpp_mov<DPP_ROW_ROR4>(D);
#endif
#if PINN_GACC_ATOMIC
      pend_D[pend_n] = D; pend_off[pend_n] = blk * 128; ++pend_n;      // added by gacc_flush below, one execution-mask region per group
      (void)old;
#else
      gacc[blk * 16 + ge] = old + D;
#endif
    };
    // entries of a hidden layer's 30 blocks, relative to the layer's first weight: the same for every layer, derived once.
    // Thread t owns entries t and t + 256 of a phase.
    int rel_hidden[2] = {-1, -1};
    if (ONE_TILE) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 256 * k, bl = e >> 4, i = (e >> 2) & 3, j = e & 3;
        if (bl < 25) { const int m = bl / 5, n = bl - 5 * m; rel_hidden[k] = (4 * m + i) * FW + 4 * n + j; }
        else if (bl < 30 && i == 0) rel_hidden[k] = FW * FW + 4 * (bl - 25) + j;       // the bias row follows the kernel
      }
    }
    double* __restrict__ const row1 = part + (size_t)blockIdx.x * R;
    // sum of one phase: all four waves have parked their NB blocks in buffer `buf`; entry e of the phase = 4 waves x 4
    // blocks in fixed order -> its place in the workgroup's gradient row.  to_index(e) = flat parameter index or -1.
    //   phase_issue   barrier (all four waves have parked the phase) + the 16 reads of this thread's two entries
    //   phase_finish  4 waves x 4 blocks added in fixed order, stored at the entry's place in the workgroup's gradient row
    // Threads without a second entry read a clamped address and store nothing (no divergent branch around the reads).
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 ps_lo[2][4], ps_hi[2][4];
    auto phase_issue = [&](const int n_entries, const int buf) {
      __syncthreads();
      const double* __restrict__ const sb = gacc_all + buf * FUSED20D_STAGE_BUF;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (256 * k >= n_entries) continue;
        const int e = tid + 256 * k < n_entries ? tid + 256 * k : n_entries - 1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          ps_lo[k][w] = *reinterpret_cast<const d2*>(sb + w * FUSED20D_STAGE_WAVE + 4 * e);
          ps_hi[k][w] = *reinterpret_cast<const d2*>(sb + w * FUSED20D_STAGE_WAVE + 4 * e + 2);
        }
      }
    };
    auto phase_finish = [&](const int n_entries, auto to_index) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (256 * k >= n_entries) continue;
        const int e = tid + 256 * k;
        const int idx = e < n_entries ? to_index(e, k) : -1;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const double t = (ps_lo[k][w].x + ps_lo[k][w].y) + (ps_hi[k][w].x + ps_hi[k][w].y);
          v = w == 0 ? t : v + t;
        }
        if (idx >= 0) row1[idx] = v;     // (as a nontemporal store -- the row streaming past the L2 -- the Adam step was 0.6 us
                                         //  LONGER, profiles/r06_ab_loopvariants.txt)
      }
    };
    auto idx_dense_h = [&](const int e, int) {
      const int m = e >> 4, i = (e >> 2) & 3, j = e & 3;
      return j != 0 ? -1 : m < 5 ? nd.off_w[H] + 4 * m + i : (i == 0 ? nd.off_b[H] : -1);
    };
    const int pt = tile * 64 + wave * 16 + q;
    const double hx = __builtin_fma(sx, x - lbx, -1.0), ht = __builtin_fma(st, t - lbt, -1.0);
    {
      const int nt = tile + gridDim.x;
      if (!ONE_TILE && nt < n_tiles) { x = xs[nt * 64 + wave * 16 + q]; t = ts[nt * 64 + wave * 16 + q]; }
    }

    // ------------------------------------------------------------------ forward
    double in[4][5];                         // [channel h,p,q,r][group]: outputs of the layer below, own (slot, point)
    double a0[5];                            // layer 0: tanh outputs (its z_x, z_t are weight constants, z_xx = 0)
    agd stash[H][5][4];                      // AGPR-resident, layers 1..H-2
    double top[5][4];                        // last hidden layer's stash entry, live across the seeds
#pragma unroll
    for (int n = 0; n < 5; ++n) {            // dense 0: p0 = (sx, 0), q0 = (0, st), r0 = 0
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
      // The 25 weight patterns of the layer are requested from LDS TWO steps ahead of the four matrix instructions that
      // consume them (sched_barrier pins the order).  Left to itself hipcc sinks every ds_read next to its consumer --
      // `ds_read2_b64; s_waitcnt lgkmcnt(0); v_mfma` 259 times per tile (round-4 ISA count) -- and a lone wave then
      // sits out the LDS latency in front of each group of matrix instructions.
      if constexpr (PA) {
        // (requested in PAIRS -- steps t + 2 and t + 3 at every even t -- so that two patterns travel in one ds_read2_b64:
        //  13 LDS instructions per GEMV instead of 25)
        auto fpat = [&](const int t) { return wd[80 * (t % 5) + 4 * (t / 5)]; };
        double Aq[4] = {fpat(0), fpat(1), 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 25; ++t) {
          const int n = t / 5, m = t - 5 * n;
          const double A = Aq[t & 3];
          if ((t & 1) == 0) {
            if (t + 2 < 25) Aq[(t + 2) & 3] = fpat(t + 2);
            if (t + 3 < 25) Aq[(t + 3) & 3] = fpat(t + 3);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c][n] = mfma444(A, in[c][m], acc[c][n]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
#pragma unroll
          for (int m = 0; m < 5; ++m) {
            const double A = wd[80 * m + 4 * n];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c][n] = mfma444(A, in[c][m], acc[c][n]);
          }
        }
      }
      STAMP2(d == 4, 29);
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const double a = tanh_d(acc[0][n]);
        channels_d(a, acc[1][n], acc[2][n], acc[3][n], in[0][n], in[1][n], in[2][n], in[3][n]);
        if (d < H - 1) {       // a is a VALU result; z_x, z_t, z_xx are raw matrix results (see agd_put_after)
          stash[d][n][0] = agd_put(a); stash[d][n][1] = agd_put_after(acc[1][n], in[1][n]);
          stash[d][n][2] = agd_put_after(acc[2][n], in[2][n]); stash[d][n][3] = agd_put_after(acc[3][n], in[3][n]);
        } else {
          top[n][0] = a; top[n][1] = acc[1][n]; top[n][2] = acc[2][n]; top[n][3] = acc[3][n];
        }
      }
      STAMP(1 + d);
    }
    // linear output layer: the pattern does not depend on the row, so all four slot lanes of a point get
    // o = (u, u_x, u_t, u_xx)
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
    double ob[4][5];                         // adjoint of the outputs of the layer below, own (slot, point)
    {  // dense H (linear, one output): z_bar = sb.  dW_H[k] = sum IN_c[k] sb_c, db_H = sum sb_h
      double sbT[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) sbT[c] = PINN_TO_POINTS(s == 0 ? sb[c] : 0.0);     // column 0 only
      {
        double D[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, old[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) old[m] = grad_fetch(BLK_H + m);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int m = 0; m < 5; ++m) D[m] = mfma444(PINN_TO_POINTS(in[c][m]), sbT[c], D[m]);
        }
        D[5] = mfma444(onesA, sbT[0], 0.0);
        phase_first = BLK_H;                                   // phase 0 of the reverse sweep: buffer 0
        stage_w = gacc_all + wave * FUSED20D_STAGE_WAVE;
#pragma unroll
        for (int m = 0; m < 6; ++m) grad_store(D[m], old[m], BLK_H + m);
        gacc_flush();
      }
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
      // pre-activation adjoints of layer d, and their point-major (rotated) copies for the weight gradient
      double zb[4][5], zbT[4][5];
      if constexpr (ONE_TILE && PA && PINN_ROT_IN_GEMV == 2)
        phase_issue(d == H - 1 ? 6 * 16 : 30 * 16, d == H - 1 ? 0 : (H - d - 1) & 1);
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        double a, zp, zq, zr;
        if (d == H - 1) { a = top[n][0]; zp = top[n][1]; zq = top[n][2]; zr = top[n][3]; }
        else { a = agd_get(stash[d][n][0]); zp = agd_get(stash[d][n][1]); zq = agd_get(stash[d][n][2]); zr = agd_get(stash[d][n][3]); }
        preact_adjoint_d(a, zp, zq, zr, ob[0][n], ob[1][n], ob[2][n], ob[3][n], zb[0][n], zb[1][n], zb[2][n], zb[3][n]);
        if constexpr (!(PA && PINN_ROT_IN_GEMV)) {
#pragma unroll
          for (int c = 0; c < 4; ++c) zbT[c][n] = PINN_TO_POINTS(zb[c][n]);
        }
      }
      STAMP2(d == 4, 20);
      // One tile: the phase before this one is summed HERE, not where its last block was parked -- the stash entries of
      // layer d, read a first time for that phase's rotated inputs, are still in registers for the adjoints above (a
      // barrier in between would end their basic block and cost a second v_accvgpr_read each).
      // The LDS pipe is the second bottleneck of this kernel (a ds_bpermute costs a wave ~31 cycles of it with four waves
      // on the CU, a ds_read_b128 ~53: profiles/r01_ubench_lds_rates.txt, r06_stamps2_new.txt), and it runs beside the
      // matrix pipe: the barrier and the 16 reads of the sum go in front of the reverse GEMV, this layer's 40 lane rotations
      // -- needed only by the weight-gradient blocks behind the GEMV -- are issued two per GEMV step, and the adds and
      // stores of the sum follow the GEMV.
      auto finish_prev = [&]() {
        if (d == H - 1) phase_finish(6 * 16, idx_dense_h);
        else phase_finish(30 * 16, [&](int, const int k) { return rel_hidden[k] < 0 ? -1 : nd.off_w[d + 1] + rel_hidden[k]; });
      };
      if constexpr (ONE_TILE) {
        if (!(PA && PINN_ROT_IN_GEMV == 2)) phase_issue(d == H - 1 ? 6 * 16 : 30 * 16, d == H - 1 ? 0 : (H - d - 1) & 1);
        if (!(PA && PINN_ROT_IN_GEMV == 1)) finish_prev();
      }
      STAMP2(d == 4, 21);
      // the first in-group's inputs of the weight-gradient blocks: formed here, rotated in the last steps of the GEMV
      constexpr bool ROT_FIRST = PA && PINN_ROT_IN_GEMV;
      double cur[4] = {0.0, 0.0, 0.0, 0.0}, first_nat[4] = {0.0, 0.0, 0.0, 0.0};
      if constexpr (ROT_FIRST) {
        double a_, zp_, zq_, zr_;
        if (d - 1 == 0) {
          a_ = a0[0]; zp_ = sx * wl[nd.off_w[0] + s]; zq_ = st * wl[nd.off_w[0] + FW + s]; zr_ = 0.0;
        } else {
          a_ = agd_get(stash[d - 1][0][0]); zp_ = agd_get(stash[d - 1][0][1]);
          zq_ = agd_get(stash[d - 1][0][2]); zr_ = agd_get(stash[d - 1][0][3]);
        }
        channels_d(a_, zp_, zq_, zr_, first_nat[0], first_nat[1], first_nat[2], first_nat[3]);
      }
      // adjoint of the layer-(d-1) outputs: in_bar[4m + i] = sum_j z_bar_j W_d[4m + i][j]
      const double* __restrict__ wd = wl + nd.off_w[d] + pr;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        ob[0][m] = ob[1][m] = ob[2][m] = ob[3][m] = 0.0;
      }
      if constexpr (PA) {
        auto rpat = [&](const int t) { return wd[80 * (t / 5) + 4 * (t % 5)]; };
        double Aq[4] = {rpat(0), rpat(1), 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 25; ++t) {
          const int m = t / 5, n = t - 5 * m;
          const double A = Aq[t & 3];
          if ((t & 1) == 0) {
            if (t + 2 < 25) Aq[(t + 2) & 3] = rpat(t + 2);
            if (t + 3 < 25) Aq[(t + 3) & 3] = rpat(t + 3);
          }
          if (PINN_ROT_IN_GEMV && t < 20) zbT[t / 5][t % 5] = PINN_TO_POINTS(zb[t / 5][t % 5]);   // one rotation (two ds_bpermute) per step
          if (PINN_ROT_IN_GEMV && t >= 20 && t < 24) cur[t - 20] = PINN_TO_POINTS(first_nat[t - 20]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 4; ++c) ob[c][m] = mfma444(A, zb[c][n], ob[c][m]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
#pragma unroll
          for (int n = 0; n < 5; ++n) {
            const double A = wd[80 * m + 4 * n];
#pragma unroll
            for (int c = 0; c < 4; ++c) ob[c][m] = mfma444(A, zb[c][n], ob[c][m]);
          }
        }
      }
      if constexpr (ONE_TILE && PA && PINN_ROT_IN_GEMV == 1) finish_prev();
      STAMP2(d == 4, 22);
      // dW_d[4m + i][4n + j]: the A operands are the layer-(d-1) output channels, rotated -- produced one in-group
      // ahead of the matrix instructions that consume them (20 values live instead of 40: the kernel sits at the
      // 256-VGPR limit, and with all of them live hipcc sank the accumulator fetches next to their uses)
      const int base = 5 + (d - 1) * 30;
      phase_first = base;                                      // phase H - d: buffers alternate
      stage_w = gacc_all + ((H - d) & 1) * FUSED20D_STAGE_BUF + wave * FUSED20D_STAGE_WAVE;
#define PINN_ROTATED_INPUTS(M, O4)                                                                          \
  do {                                                                                                      \
    double a_, zp_, zq_, zr_;                                                                               \
    if (d - 1 == 0) {                                                                                       \
      const int f_ = 4 * (M) + s;                                                                           \
      a_ = a0[M]; zp_ = sx * wl[nd.off_w[0] + f_]; zq_ = st * wl[nd.off_w[0] + FW + f_]; zr_ = 0.0;         \
    } else {                                                                                                \
      a_ = agd_get(stash[d - 1][M][0]); zp_ = agd_get(stash[d - 1][M][1]);                                  \
      zq_ = agd_get(stash[d - 1][M][2]); zr_ = agd_get(stash[d - 1][M][3]);                                 \
    }                                                                                                       \
    double h_, p_, q_, r_;                                                                                  \
    channels_d(a_, zp_, zq_, zr_, h_, p_, q_, r_);                                                          \
    O4[0] = PINN_TO_POINTS(h_); O4[1] = PINN_TO_POINTS(p_);                                             \
    O4[2] = PINN_TO_POINTS(q_); O4[3] = PINN_TO_POINTS(r_);                                             \
  } while (0)
      if constexpr (!ROT_FIRST) PINN_ROTATED_INPUTS(0, cur);
#pragma unroll
      for (int m = 0; m < 5; ++m) {          // five independent accumulator chains per in-group
        double D[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, old[5], nxt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int n = 0; n < 5; ++n) old[n] = grad_fetch(base + m * 5 + n);
        if (m + 1 < 5) PINN_ROTATED_INPUTS((m + 1 < 5 ? m + 1 : 4), nxt);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int n = 0; n < 5; ++n) D[n] = mfma444(cur[c], zbT[c][n], D[n]);
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) grad_store(D[n], old[n], base + m * 5 + n);
        gacc_flush();
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
        STAMP2(d == 4, 23 + m);
      }
#undef PINN_ROTATED_INPUTS
      {
        double D[5], old[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) old[n] = grad_fetch(base + 25 + n);
#pragma unroll
        for (int n = 0; n < 5; ++n) D[n] = mfma444(onesA, zbT[0][n], 0.0);
#pragma unroll
        for (int n = 0; n < 5; ++n) grad_store(D[n], old[n], base + 25 + n);
        gacc_flush();
      }
      STAMP(2 * H + 1 - d);
    }
    {  // dense 0: inputs (hx, ht, 1) in channel h, (sx, 0, 0) in channel p, (0, st, 0) in channel q
      const double hxT = PINN_TO_POINTS(hx), htT = PINN_TO_POINTS(ht);
      const double Ah = i4 == 0 ? hxT : i4 == 1 ? htT : i4 == 2 ? 1.0 : 0.0;
      const double Ap = i4 == 0 ? sx : 0.0, Aq = i4 == 1 ? st : 0.0;
      double bT[3][5];
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const int f = 4 * n + s;
        double bh, bp, bq, br;
        preact_adjoint_d(a0[n], sx * wl[nd.off_w[0] + f], st * wl[nd.off_w[0] + FW + f], 0.0, ob[0][n], ob[1][n],
                         ob[2][n], ob[3][n], bh, bp, bq, br);
        bT[0][n] = PINN_TO_POINTS(bh); bT[1][n] = PINN_TO_POINTS(bp); bT[2][n] = PINN_TO_POINTS(bq);
      }
      if constexpr (ONE_TILE) {                                // layer 1's phase
        phase_issue(30 * 16, (H - 1) & 1);
        phase_finish(30 * 16, [&](int, const int k) { return rel_hidden[k] < 0 ? -1 : nd.off_w[1] + rel_hidden[k]; });
      }
      double D[5], old[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) old[n] = grad_fetch(n);
#pragma unroll
      for (int n = 0; n < 5; ++n) D[n] = mfma444(Ah, bT[0][n], 0.0);
#pragma unroll
      for (int n = 0; n < 5; ++n) D[n] = mfma444(Ap, bT[1][n], D[n]);
#pragma unroll
      for (int n = 0; n < 5; ++n) D[n] = mfma444(Aq, bT[2][n], D[n]);
      phase_first = 0;                                         // phase H
      stage_w = gacc_all + (H & 1) * FUSED20D_STAGE_BUF + wave * FUSED20D_STAGE_WAVE;
#pragma unroll
      for (int n = 0; n < 5; ++n) grad_store(D[n], old[n], n);
      gacc_flush();
      if constexpr (ONE_TILE) {
        phase_issue(5 * 16, H & 1);
        phase_finish(5 * 16, [&](const int e, int) {
          const int f = 4 * (e >> 4) + (e & 3), i = (e >> 2) & 3;
          return i == 0 ? nd.off_w[0] + f : i == 1 ? nd.off_w[0] + FW + f : i == 2 ? nd.off_b[0] + f : -1;
        });
      }
    }
    if (ONE_TILE) break;
  }
  STAMP(2 * H + 1);
#undef PINN_LANE_INDICES
#undef PINN_TO_POINTS

  // -------------------------------------------------------------------- one gradient row per workgroup
  {
    // row_index[e]: flat parameter index of entry e of the block list (-1: padding), built once on the host;
    // fetched first so that its (cold) latency hides under the wave sums and the two barriers  (it does: prefetching
    // the table into LDS with the weights' DMA changed nothing -- Adam step 40.79 vs 40.81 us, same box)
    constexpr int NE = NBLK * 16, NIT = ONE_TILE ? 1 : (NE + 255) / 256;   // (one tile: the row was written phase by phase)
    int idx[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 256 * it;
      idx[it] = (!ONE_TILE && e < NE) ? row_index[e] : -1;
    }
    const double t0 = wave_sum(lacc[0]), t1 = wave_sum(lacc[64]);
    const double t2 = PDE == 1 ? wave_sum(lacc[128]) : 0.0, t3 = PDE == 1 ? wave_sum(lacc[192]) : 0.0;
    __syncthreads();                                   // every wave's accumulators are final
    double* const scal = wl;                           // the weight copy is dead: 4 x 4 loss / lambda partials
    if (lane == 0) { scal[wave * 4 + 0] = t0; scal[wave * 4 + 1] = t1; scal[wave * 4 + 2] = t2; scal[wave * 4 + 3] = t3; }
    __syncthreads();
    double* __restrict__ row = part + (size_t)blockIdx.x * R;
    if (!ONE_TILE) {
      double v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = tid + 256 * it;
        const int ee = e < NE ? e : 0;
        v[it] = ((gacc_all[ee] + gacc_all[NE + ee]) + gacc_all[2 * NE + ee]) + gacc_all[3 * NE + ee];
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (idx[it] >= 0) row[idx[it]] = v[it];
    } else if (blockIdx.x >= n_tiles) {                // (cannot happen: the launch plan gives every workgroup a tile)
      for (int i = tid; i < nd.n_theta; i += 256) row[i] = 0.0;
    }
    if (tid < 4) {
      const double v = ((scal[tid] + scal[4 + tid]) + scal[8 + tid]) + scal[12 + tid];
      if (tid == 0) { row[nd.n_theta + 0] = v; row[nd.n_theta + 2] = 0.0; }
      if (tid == 1) row[nd.n_theta + 1] = v;
      if (PDE == 1 && tid == 2) row[nd.n_net] = v;
      if (PDE == 1 && tid == 3) row[nd.n_net + 1] = v;
    }
  }
  STAMP(2 * H + 2);
}

// returns a hipError_t (0 = ok)
template <int PDE, int H>
inline int fused20d_launch(const NetDesc& nd, const SetDesc& sd, const double* th, const double* xs, const double* ts,
                           const double* tgt, double lbx, double lbt, double sx, double st, double nu, double* part,
                           int R, int n_wg, const int* row_index, hipStream_t stream, long long* stamps = nullptr,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  if (!w20_layout_ok(nd, H, PDE == 1)) return (int)hipErrorInvalidValue;
  const size_t lds = fused20d_lds_bytes(H, nd.n_theta);
  static unsigned long long attr_set = 0;
  if (first_call_on_device(attr_set)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_fused20d<PDE, H, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)k_fused20d<PDE, H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_tiles = sd.n_pad / 64;
  auto* const kern = n_wg >= n_tiles ? k_fused20d<PDE, H, true> : k_fused20d<PDE, H, false>;
  if (ev_start && ev_stop)
    hipExtLaunchKernelGGL(kern, dim3(n_wg), dim3(256), lds, stream, ev_start, ev_stop, 0, th, xs, ts, tgt, part,
                          row_index, R, n_tiles, lbx, lbt, sx, st, nu, sd, stamps, w20_desc(H, PDE == 1));
  else
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(256), lds, stream, th, xs, ts, tgt, part, row_index, R, n_tiles, lbx,
                       lbt, sx, st, nu, sd, stamps, w20_desc(H, PDE == 1));
  return (int)hipGetLastError();
}

}  // namespace pinn
