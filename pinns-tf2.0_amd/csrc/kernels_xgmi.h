// kernels_xgmi.h -- one-shot all-reduce of the [P+4] float64 gradient/loss vector through peer-mapped mailboxes.
//
// The exchange step of the data-parallel path is one 24 KB vector per evaluation: far below the
// size where a ring or tree pays off, and on the critical path of every optimiser step.  xGMI is point-to-point,
// so the latency-optimal schedule is the trivial one: every rank stores its vector straight into a slot of every
// peer's mailbox (7 concurrent streams, one per link), and every rank adds the slots of its own mailbox in rank
// order.  That is one kernel -- the same launch that reduces the per-workgroup partial rows, and for Adam also the
// one that applies the update -- instead of reduce -> ncclAllReduce -> update, and since every rank adds in the
// same order the replicas stay bit-identical.
//
// No fence, no counter, no separate flag: each float64 travels as one 16-byte line {lo32, seq32, hi32, seq32}
// written by a single global_store_dwordx4 (the "LL" idea of NCCL/RCCL's low-latency protocol): a line is valid
// when both sequence words equal the evaluation's sequence number, whatever order or granularity (>= 8 bytes) the
// fabric delivers the writes in.  The consumer thread of column c polls the n_ranks - 1 lines of column c in its
// own mailbox (its own contribution stays in a register) with cache-bypassing loads.
//
// Mailbox (hipExtMallocWithFlags(hipDeviceMallocUncached), mapped into every peer with hipIpc):
//   line[2][n_ranks][Rp]  16 bytes each; [parity = seq & 1][source rank][column].  A rank can be at most one
//   evaluation ahead of a peer (it needs the peer's lines of evaluation s to finish s), so two generations
//   never collide.  Waiting is bounded by wall_clock64: a lost peer becomes an error code, never a hang.
// RCCL stays the fallback (pinn_comm_set_mode) and the mailboxes are switched on only after a self-test passed
// on every rank.
#pragma once
#include "kernels_optim.h"

namespace pinn {

constexpr int XG_MAX_RANKS = 16;
constexpr long long XG_TIMEOUT_TICKS = 30ll * 100000000ll;      // 30 s of the 100 MHz wall clock (training)
constexpr long long XG_TEST_TIMEOUT_TICKS = 5ll * 100000000ll;  // 5 s (attach-time self-test)

typedef unsigned int xg_line_t __attribute__((ext_vector_type(4)));

struct XgPeers {
  xg_line_t* box[XG_MAX_RANKS];               // base of peer r's mailbox as mapped on this rank
  int n_ranks, rank, Rp;                      // Rp = slot pitch in lines
};

inline size_t xg_box_bytes(int n_ranks, int Rp) { return (size_t)2 * n_ranks * Rp * sizeof(xg_line_t); }

// system-coherent, cache-bypassing 16-byte accesses
__device__ __forceinline__ void xg_store(xg_line_t* dst, xg_line_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ xg_line_t xg_load(const xg_line_t* src) {
  xg_line_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
  return v;
}

template <typename real, bool ADAM>
__global__ __launch_bounds__(RED_THREADS) void k_reduce_xgmi(const real* __restrict__ part, int n_rows, int R,
                                                             double* __restrict__ gl, XgPeers px,
                                                             unsigned int seq, long long timeout_ticks,
                                                             int* __restrict__ err, int n, double* __restrict__ theta,
                                                             real* __restrict__ theta_r, double* __restrict__ m,
                                                             double* __restrict__ v, double alpha, double b1, double b2,
                                                             double eps, double* __restrict__ loss3, NetDesc nd,
                                                             float* __restrict__ img, TileScratch ts = TileScratch{}) {
  __shared__ double sh[RED_SLICES][RED_COLS];
  // the peers' mailbox pointers are indexed by a run-time rank: as a by-value kernel argument that indexing sent the whole
  // table through scratch (18 spilled registers, 76 B per lane until round 5).  Copied once into LDS with compile-time
  // indices, read from there with the run-time one: no scratch.
  __shared__ xg_line_t* sbox[XG_MAX_RANKS];
#pragma unroll
  for (int r = 0; r < XG_MAX_RANKS; ++r)
    if (threadIdx.x == (unsigned)r) sbox[r] = px.box[r];
  __syncthreads();
  const int q = threadIdx.x >> 6, n_cb = (R + RED_COLS - 1) / RED_COLS;
  const int par = (int)(seq & 1u), nr = px.n_ranks, me = px.rank;
  // One column block per workgroup normally (grid = n_cb).  When several ranks SHARE one device (single-GPU tests) the
  // host caps the grid (XgState::grid_cap) and a workgroup walks several column blocks: the polling wave of every
  // workgroup otherwise sits on every CU of the device while the peer it waits for cannot place a full-CU kernel.
  // exchange column c (this rank's sum g) with the peers and finish it (-> gl, Adam)
  auto exchange = [&](const int c, const double g) {
    {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(g);
      const xg_line_t line = {(unsigned int)bits, seq, (unsigned int)(bits >> 32), seq};
      for (int k = 1; k < nr; ++k) {            // start with the next rank: spreads the traffic over the links
        const int r = (me + k) % nr;
        xg_store(sbox[r] + (size_t)(par * nr + me) * px.Rp + c, line);
      }
    }
    double tot = 0;
    bool lost = false;                          // a peer's line never arrived: this column's sum is not valid
    const long long t0 = wall_clock64();
    // once a peer has been declared lost nobody waits again: one bounded stall, then the host sees the error code
    long long limit = timeout_ticks;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) limit = 0;
    for (int r = 0; r < nr; ++r) {
      if (r == me) { tot += g; continue; }
      const xg_line_t* src = sbox[me] + (size_t)(par * nr + r) * px.Rp + c;
      xg_line_t line = xg_load(src);
      while (line.y != seq || line.w != seq) {
        if (wall_clock64() - t0 > limit) { atomicExch(err, 1); lost = true; break; }
        __builtin_amdgcn_s_sleep(1);
        line = xg_load(src);
      }
      tot += __longlong_as_double((long long)(((unsigned long long)line.z << 32) | line.x));
    }
    if (lost) return;                           // this COLUMN keeps its old gl / weight / moments; columns whose lines did
                                                // arrive are updated, so after a lost peer the model state is a mix of two
                                                // iterates: the host sees err at its next synchronisation (xg_check), raises
                                                // PINN_ECOMM and marks the context's weights undefined (pinn_get_weights
                                                // refuses until pinn_set_weights)
    gl[c] = tot;
    if (ADAM) {
      if (c < n) {
        const double mi = m[c] + (1.0 - b1) * (tot - m[c]);
        const double vi = v[c] + (1.0 - b2) * (tot * tot - v[c]);
        m[c] = mi;
        v[c] = vi;
        const double t = theta[c] - alpha * mi / (sqrt(vi) + eps);
        theta[c] = t;
        theta_r[c] = (real)t;
        pack_store_any(nd, img, c, (float)t);
      } else if (loss3 && c < n + 3) {
        loss3[c - n] = tot;
      }
    }
  };
  const int n_blocks = n_cb + (ts.gscr ? SLOT_SPLIT * ts.n_slots : 0);   // column blocks, then the half slots of k_t16_fused's scratch
  for (int cb = blockIdx.x; cb < n_blocks; cb += gridDim.x) {
    if (cb != (int)blockIdx.x) __syncthreads();           // sh of the previous block has been consumed
    // the columns this thread finishes: four of a scratch slot or one of a column block -- ONE copy of the exchange below
    // (inlined once per call site it spilled 18-21 registers to scratch under the 128-register bound of 1024 threads)
    int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
    if (cb >= n_cb) {
      double tot4[4];
      int e, L;
      reduce_slot(ts, n_rows, cb - n_cb, sh, tot4, e, L);
      if (threadIdx.x < 64 / SLOT_SPLIT) {
        c0 = ts.column(e, L, 0); c1 = ts.column(e, L, 1); c2 = ts.column(e, L, 2); c3 = ts.column(e, L, 3);
        g0 = tot4[0]; g1 = tot4[1]; g2 = tot4[2]; g3 = tot4[3];
      }
    } else {
      const int c = cb * RED_COLS + (threadIdx.x & 63);
      const bool skip = ts.backed(c);
      const double g = reduce_column(part, n_rows, R, skip ? R : c, q, sh);
      if (q == 0 && c < R && !skip) { c0 = c; g0 = g; }
    }
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const int c = k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3;
      const double g = k == 0 ? g0 : k == 1 ? g1 : k == 2 ? g2 : g3;
      if (c >= 0) exchange(c, g);
    }
  }
}

}  // namespace pinn
