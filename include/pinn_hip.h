/*
 * pinn_hip.h -- C ABI of libpinn_hip.so, the MI355X (gfx950) PINN training engine.
 *
 * The reference (pierremtb/PINNs-TF2.0) has no FFI/plugin boundary: its hot path is
 * Python calling TensorFlow eager ops.  This ABI therefore sits *under* the reference's
 * Python surface; each entry point below replaces the TensorFlow work done by the cited
 * reference method, and `pinns-tf2.0_amd/utils/neuralnetwork.py` (same class/method
 * names as the reference) is its only caller.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PINN_E* code on failure;
 *     pinn_last_error() returns a thread-local message for the last failure.
 *   - host pointers are borrowed for the duration of the call only; host interchange
 *     dtype is always float64 (the reference's dtype, utils/neuralnetwork.py:24-26),
 *     whatever the kernel compute dtype.
 *   - the flat weight vector uses the reference layout (utils/neuralnetwork.py:68-89):
 *     per Dense layer W.ravel() (row-major [fan_in, fan_out]) then b; the identification
 *     problem appends lambda_1, lambda_2 (1d-burgers/ide_cont_burgers.py:98-107).
 *   - a pinn_ctx is bound to one device and one stream and is not thread-safe.
 *   - no torch / Python types anywhere in the signatures.
 */
#ifndef PINN_HIP_H
#define PINN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pinn_ctx pinn_ctx;

enum {
  PINN_PDE_BURGERS = 0,     /* 1d-burgers/inf_cont_burgers.py:65-90   f = u_t + u u_x - nu u_xx            */
  PINN_PDE_BURGERS_IDE = 1, /* 1d-burgers/ide_cont_burgers.py:56-85   f = u_t + l1 u u_x - exp(l2) u_xx    */
  PINN_PDE_SCHRODINGER = 2, /* 1dcomplex-schrodinger/inf_cont_schrodinger.py:79-129                        */
  /* discrete-time (implicit Runge-Kutta) models: 1 input (x), q or q+1 outputs, see pinn_disc_set_stage */
  PINN_PDE_BURGERS_DISC = 3,     /* 1d-burgers/inf_disc_burgers.py:57-95   N = U U_x - nu U_xx             */
  PINN_PDE_BURGERS_DISC_IDE = 4  /* 1d-burgers/ide_disc_burgers.py:81-115  N = l1 U U_x - exp(l2) U_xx     */
};
enum { PINN_F32 = 0, PINN_F64 = 1 };
enum {
  PINN_OK = 0, PINN_EINVAL = -1, PINN_EHIP = -2, PINN_ESTATE = -3, PINN_ECOMM = -4,
  PINN_EUNSUPPORTED = -5
};

/* diagnostics */
const char* pinn_last_error(void);
int pinn_abi_version(void);   /* 2: + discrete-time models, device LHS, mailbox all-reduce, kernel paths 3..6; 3: + pinn_residual_at;
                                 4: + pinn_error_l2, pinn_get_status; 5: + pinn_runtime_versions, pinn_debug_t16_deal;
                                 6: + pinn_adam_enqueue / _collect, pinn_lbfgs_enqueue / _collect,
                                    pinn_weights_snapshot / _restore */
int pinn_device_count(int* n);
/* HIP runtime / driver (hipRuntimeGetVersion, hipDriverGetVersion) and RCCL (ncclGetVersion) this process bound; any
 * pointer may be NULL.  No device is touched. */
int pinn_runtime_versions(int* hip_runtime, int* hip_driver, int* rccl);
/* name[0..cap) <- hipDeviceProp_t.gcnArchName etc. for Logger's banner (utils/logger.py:13-15) */
int pinn_device_info(int device, char* name, int cap, int* n_cu, int64_t* hbm_bytes);

/* NeuralNetwork.__init__ (utils/neuralnetwork.py:8-47): layers = hp["layers"], lb/ub = the
 * Lambda normalisation bounds.  dtype = kernel arithmetic type. */
int pinn_create(pinn_ctx** out, const int* layers, int n_layers, const double* lb,
                const double* ub, int pde_kind, int dtype, int device);
int pinn_destroy(pinn_ctx* c);
int pinn_num_params(pinn_ctx* c, int64_t* n);       /* incl. lambda_1, lambda_2 for IDE */

/* Point sets.  *_total are the GLOBAL set sizes used as the mean() denominators, so that a
 * rank holding a shard (n < n_total) produces partial sums that add up across ranks.
 *   collocation: self.x_f, self.t_f      (inf_cont_burgers.py:55-56, inf_cont_schrodinger.py:56-57)
 *   data:        fit(X_u, u) arguments   (utils/neuralnetwork.py:138-143); targets [n, n_out];
 *                for PINN_PDE_BURGERS_IDE these points also carry the residual
 *                (ide_cont_burgers.py:88-91) and no collocation set is used.
 *   boundary:    X_lb, X_ub              (inf_cont_schrodinger.py:50-53), Schrodinger only. */
int pinn_set_collocation(pinn_ctx* c, const double* X_f, int64_t n, int64_t n_total);
int pinn_set_data(pinn_ctx* c, const double* X_u, const double* u, int64_t n, int64_t n_total);
/* Collocation points drawn on the device instead of handed over: points [first, first + count) of an
 * n_design-point Latin hypercube over [lb, ub] -- the role of `lb + (ub - lb) * lhs(2, N_f)`
 * (1d-burgers/burgersutil.py:122), same kind of design, counter-based stream (csrc/kernels_sampling.h), so ranks
 * can build disjoint shards of one design and a re-draw with a new seed is a single launch (no reallocation when
 * count is unchanged).  The mean() denominator becomes n_design.  pinn_get_collocation reads the current set
 * back, [n][2] float64 (also valid after pinn_set_collocation). */
int pinn_lhs_collocation(pinn_ctx* c, int64_t n_design, int64_t first, int64_t count, uint64_t seed);
int pinn_get_collocation(pinn_ctx* c, double* X, int64_t n);
int pinn_set_boundary(pinn_ctx* c, const double* X_lb, const double* X_ub, int64_t n,
                      int64_t n_total);
/* Discrete-time models (pde_kind 3, 4; layers[0] == 1, lb/ub hold one value each).  A stage set contributes
 *     sum_{p,j} ( U[p][j] + sum_k N(U)[p][k] M[j][k] - target[p] )^2        (a SUM, inf_disc_burgers.py:92-95)
 * to the loss, N acting on the first q outputs.  M is [n_out][q] row-major and already carries the step size
 * (dt * IRK_weights for U_0_model, inf_disc_burgers.py:89; dt * IRK_alpha and -dt * (IRK_beta - IRK_alpha) for
 * ide_disc_burgers.py:92,108); M == NULL: no IRK term, the set penalises U itself (the walls x_1,
 * inf_disc_burgers.py:93-95).  target is [n] (broadcast over the outputs like the reference's [n,1] tensor).
 * set is 0 or 1; its loss is reported in terms[set].  n == 0 removes the set. */
int pinn_disc_set_stage(pinn_ctx* c, int set, const double* x, const double* target, int64_t n,
                        const double* M, int q);
/* U_0_model / U_1_model at arbitrary points with the table of `set` (ide_disc_burgers.py:188-193):
 * out [n][n_out] = U + N(U) M^T.  pinn_predict returns the plain network outputs U [n][n_out]. */
int pinn_disc_predict(pinn_ctx* c, int set, const double* x, int64_t n, double* out);

/* get_params (inf_cont_burgers.py:92): p[0] = nu for PINN_PDE_BURGERS and PINN_PDE_BURGERS_DISC */
int pinn_set_pde_params(pinn_ctx* c, const double* p, int n);

/* get_weights / set_weights (utils/neuralnetwork.py:68-89) */
int pinn_set_weights(pinn_ctx* c, const double* w, int64_t n);
int pinn_get_weights(pinn_ctx* c, double* w, int64_t n);

/* NeuralNetwork.grad / get_loss_and_flat_grad (utils/neuralnetwork.py:55-59, 91-103) at the
 * current weights: forward, residual, loss, flat gradient (+ all-reduce when a communicator is
 * attached).  grad may be NULL.  terms (may be NULL) <- {mse_f, mse_data, mse_boundary}. */
int pinn_loss_grad(pinn_ctx* c, double* loss, double* grad, double* terms);

/* tf.keras.optimizers.Adam (utils/neuralnetwork.py:19-22) + tf_optimization loop (:105-116).
 * pinn_adam_run does n_steps x {loss_grad; apply_gradients}; losses[i] (may be NULL) is the
 * loss *before* update i, as tf_optimization_step returns it.  With losses == NULL the call
 * returns without synchronising the stream. */
int pinn_adam_init(pinn_ctx* c, double lr, double beta1, double beta2, double eps);
int pinn_adam_run(pinn_ctx* c, int n_steps, double* losses);
/* the same, returning the three loss parts of every step, terms3[i] = (residual, data, boundary) before update i:
 * what the reference's Schrodinger loss prints on every evaluation (inf_cont_schrodinger.py:128) */
int pinn_adam_run_terms(pinn_ctx* c, int n_steps, double* terms3);

/* custom_lbfgs.lbfgs (utils/custom_lbfgs.py:39-236) as driven by nt_optimization_steps
 * (utils/neuralnetwork.py:131-136), device-resident.  pinn_lbfgs_begin does the initial
 * evaluation (:65-76); pinn_lbfgs_run advances up to n_iters iterations.
 *   iters[i], losses[i]: the (nIter, f) pairs custom_lbfgs would have passed to log_fn (:217-218)
 *   n_logged: how many pairs were written;  done: 0 running, 1 maxIter reached, >1 break reason
 * The last-iteration quirk is reproduced: the model weights end at the last *evaluated* x.
 * pinn_lbfgs_begin only enqueues (it returns before the initial evaluation has run, except with the
 * mailbox exchange attached); pinn_lbfgs_run synchronises once, at its end, to read state and log. */
int pinn_lbfgs_begin(pinn_ctx* c, int max_iter, double lr, int n_corr, double tol_fun,
                     double tol_x, double max_eval);
int pinn_lbfgs_run(pinn_ctx* c, int n_iters, int* iters, double* losses, int* n_logged,
                   int* done);
/* The same two loops with the host one chunk behind the GPU (ABI v6) -- how NeuralNetwork.fit logs every
 * log_frequency-th epoch (utils/neuralnetwork.py:105-109, utils/logger.py:45-51) without idling the device at a log line:
 * ..._enqueue puts a chunk of steps into the stream, with asynchronous copies of its losses / log entries / optimiser state
 * into pinned buffers behind an event, and returns a ticket at once; ..._collect waits for THAT chunk only and hands its
 * results out, while the chunk enqueued after it is already running.  Up to 4 chunks may be in flight; tickets are
 * collected in the order they were issued; pinn_lbfgs_begin drops whatever is still in flight (a restart).  The kernels
 * launched are those of pinn_adam_run / pinn_lbfgs_run on the same chunk sizes, so the results are bit-identical.
 * pinn_lbfgs_collect: iters / losses hold `cap` entries (enough: the iterations enqueued since the last collect + 1).
 * An L-BFGS chunk enqueued after the run has ended (done != 0 seen one chunk late) changes nothing on the device. */
int pinn_adam_enqueue(pinn_ctx* c, int n_steps, int* ticket);
int pinn_adam_enqueue_terms(pinn_ctx* c, int n_steps, int* ticket);   /* as pinn_adam_run_terms: collect returns 3 n values */
int pinn_adam_collect(pinn_ctx* c, int ticket, double* losses);
int pinn_lbfgs_enqueue(pinn_ctx* c, int n_iters, int* ticket);
int pinn_lbfgs_collect(pinn_ctx* c, int ticket, int cap, int* iters, double* losses, int* n_logged, int* done);
/* Device-side copies of the flat weight vector taken / put back in stream order (slots 0..3; ABI v6): what a caller that
 * judges a chunk one chunk late (the restart guard of NeuralNetwork.nt_optimization; the reference has neither) returns to
 * without a host round trip.  pinn_weights_restore also refreshes the compute-dtype mirror, like pinn_set_weights. */
int pinn_weights_snapshot(pinn_ctx* c, int slot);
int pinn_weights_restore(pinn_ctx* c, int slot);
/* 0: one-workgroup kernel performing the two-loop recursion in the reference's operation order;
 * 1 (default, history <= 61): compact form -- all dot products of an iteration in one parallel
 * kernel, recursion on the Gram matrices.  Same mathematics; rounding differs at 1e-16. */
int pinn_lbfgs_set_mode(pinn_ctx* c, int mode);
/* x as custom_lbfgs returns it (:236) -- one step past the model weights */
int pinn_lbfgs_get_x(pinn_ctx* c, double* x, int64_t n);

/* self.model(X_star) (utils/neuralnetwork.py:151-153): out[N, n_out] */
int pinn_predict(pinn_ctx* c, const double* X, int64_t n, double* out);
/* The scripts' error metric on the device: err = ||ref - pred||_2 / ||ref||_2 with pred = self.model(X) at the
 * n points X [n][2] (1d-burgers/inf_cont_burgers.py:114-116 via utils/logger.py:56-60) -- forward sweep, fixed-order
 * float64 reduction, 24 bytes back.  kind 0: ref is [n][n_out], compared element-wise; kind 1: ref is [n] and is
 * compared with the modulus sqrt(sum_o pred_o^2) (the |h| of 1dcomplex-schrodinger/inf_cont_schrodinger.py:155-158).
 * X and ref are kept on the device: a repeated call with the same grid (and pinn_predict / pinn_residual_at on it)
 * uploads nothing. */
int pinn_error_l2(pinn_ctx* c, const double* X, const double* ref, int64_t n, int kind, double* err);
/* f_model() at the stored collocation points (inf_cont_burgers.py:65-90): f[n_f, n_out]
 * (IDE: at the data points) */
int pinn_residual(pinn_ctx* c, double* f, int64_t n);
/* f_model(X) at n caller-supplied points X [n][2] -> f [n][n_out]: what the identification script's predict
 * evaluates on X_star (1d-burgers/ide_cont_burgers.py:169-172) */
int pinn_residual_at(pinn_ctx* c, const double* X, int64_t n, double* f);

/* Failure detection (the reference has none: a NaN loss just propagates, utils/custom_lbfgs.py:154).
 * n_evals = loss+gradient evaluations since pinn_create; first_nonfinite_eval = 1-based number of the first one whose
 * reduced loss was NaN/Inf, 0 if none.  Recording only: optimiser trajectories are unchanged.  When non-zero,
 * pinn_last_error() carries a message too.  Either pointer may be NULL. */
int pinn_get_status(pinn_ctx* c, int64_t* n_evals, int64_t* first_nonfinite_eval);

/* Data-parallel: one process per GPU, RCCL all-reduce(SUM) of [grad | loss terms].
 * Rank 0 calls pinn_comm_unique_id and ships the 128 bytes to the other ranks by any
 * out-of-band channel; every rank then calls pinn_comm_init. */
int pinn_comm_unique_id(char* id128);
int pinn_comm_init(pinn_ctx* c, const char* id128, int n_ranks, int rank);
/* Low-latency alternative to the RCCL call for the [P+4] vector: every rank maps every peer's mailbox (hipIpc) and
 * the reduction kernel itself stores, signals, waits and adds in rank order (csrc/kernels_xgmi.h).  Protocol:
 *   1. every rank: pinn_comm_xgmi_export(n_ranks, rank, handle64)        -> 64-byte hipIpcMemHandle_t
 *   2. ship all handles to all ranks (out of band, like the RCCL id), rank-major [n_ranks][64]
 *   3. every rank: pinn_comm_xgmi_attach(handles, n_ranks, &mapped)      -> mapped = 1 if every peer mailbox is mapped
 *   4. only if mapped on ALL ranks (agreed out of band): pinn_comm_xgmi_selftest(&ok) everywhere -- a few exchange
 *      rounds on integer-valued vectors checked against the closed-form sum, bounded waits (5 s)
 *   5. if ok on ALL ranks: pinn_comm_set_mode(2) everywhere; otherwise stay on / return to RCCL (mode 1).
 * pinn_comm_get_mode: 0 no communicator, 1 RCCL, 2 mailboxes.  Works without an RCCL communicator too. */
int pinn_comm_xgmi_export(pinn_ctx* c, int n_ranks, int rank, char* handle64);
int pinn_comm_xgmi_attach(pinn_ctx* c, const char* handles, int n_handles, int* mapped_ok);
int pinn_comm_xgmi_selftest(pinn_ctx* c, int* ok);
/* Wall time per exchange of the [P+4] vector with the given implementation (1 RCCL: k_reduce_rows + ncclAllReduce,
 * 2 mailboxes: k_reduce_xgmi), measured on this node: what init_engine_comm uses to pick the faster one.
 * Collective: every rank must call it with the same mode and iteration count. */
int pinn_comm_benchmark(pinn_ctx* c, int mode, int iters, double* us_per_iter);
int pinn_comm_set_mode(pinn_ctx* c, int mode);
int pinn_comm_get_mode(pinn_ctx* c, int* mode);

/* Measurement: bracket launches of the dominant kernel (the loss+grad kernels) with hipEvents
 * on the engine's stream.  An event record costs ~5 us on the GPU timeline, so only one
 * evaluation out of `every` is sampled (at most max_evals samples).  pinn_timing_read drains
 * them into avg_ms[5]: [0] forward sweep, [1] forward+reverse sweeps (the loss+grad kernel),
 * [2] whole evaluation, [3] what an EMPTY event bracket reads on this stream (calibrated at
 * enable time; subtract it from [1..2] to compare with rocprofv3 kernel durations),
 * [4] = 1 when [0] is the exact begin-to-end duration of the single loss+grad kernel (the events
 * were attached to the launch itself, hipExtLaunchKernelGGL: kernel paths 1, 2 and 7) and needs no correction;
 * n = evaluations sampled. */
int pinn_timing_enable(pinn_ctx* c, int max_evals, int every);
int pinn_timing_read(pinn_ctx* c, double* avg_ms, int* n);
int pinn_sync(pinn_ctx* c);
/* which kernel family serves the loss+grad evaluation: 0 generic (one lane per point), 1 fused width-20 (HBM stash),
 * 2 fused 8x20 float32 (MFMA GEMVs, register stash), 3 wide MFMA sweeps (width 100, two outputs, float32),
 * 4 shape-generic MFMA sweeps (any width <= 128 / 64 in float64, any depth), 5 / 6 = 4's forward / reverse half
 * paired with the generic other half (tests), 7 fused 8x20 float64 (v_mfma_f64_4x4x4 GEMVs, register stash, no
 * inter-wave exchange), 8 fused float64 MFMA sweep for hidden widths 65..128 and 2-4 hidden layers (forward + reverse of a
 * 16-point group in one kernel, stash in registers: the Schrodinger net in the reference's arithmetic).  The engine
 * picks the fastest eligible family at pinn_create. */
int pinn_set_kernel_path(pinn_ctx* c, int path);
int pinn_get_kernel_path(pinn_ctx* c, int* path);
/* Profiling build (-DPINN_STAMPS) only: one evaluation with a per-wave s_memtime timeline of the
 * fused kernel; out[wave][32] ticks, n_waves = 4 x workgroups.  PINN_EUNSUPPORTED otherwise. */
int pinn_debug_stamps(pinn_ctx* c, long long* out, int64_t cap, int64_t* n_waves);
/* Profiling build only: s_memtime stamps of the most recent k_lbc_coef launch (7 used of 16). */
int pinn_debug_coef_stamps(long long* out16);
/* Profiling build only: s_memtime stamps of workgroup 0's second group in the most recent k_t16_fused launch,
 * out[wave 0..7][64] (profiles/t16f_stamps.py names the phases). */
int pinn_debug_t16f_stamps(long long* out512);
/* Host-side launch plan of k_t16_fused for hidden width W (65..128; no device is touched): out[0..7] / out[8..15] = each
 * wave's range [lo, hi) in the list of full 16x16 gradient tiles, out[16..23] / out[24..31] = its range in the list of
 * 4-row / 4-column strips, out[32..39] = first feature row of the layer GEMMs it owns, out[40..47] = strips of four rows it
 * owns (4 = a 16-row tile), out[48] = 1 when the last tile per side runs as strips.  tests/test_host_api.py checks that
 * every tile and every row is dealt exactly once for every width. */
int pinn_debug_t16_deal(int W, int* out49);

#ifdef __cplusplus
}
#endif
#endif /* PINN_HIP_H */
