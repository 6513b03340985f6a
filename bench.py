#!/usr/bin/env python3
"""bench.py -- collocation-points/sec of the PINN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f64] [--nf-per-gpu M]

A *step* is one optimiser iteration = one full-batch pass of the hot path (Taylor-mode forward
over every collocation point, PDE residual, loss reduction, flat gradient, optimiser update)
over the rank's shard.  The workload is BASELINE.json configs[1]: 1D Burgers continuous
inference, 8x20 tanh MLP, N_u=100, N_f=10000 *per GPU* (weak scaling; at N=1 this is exactly the
reference configuration), Adam then L-BFGS in the reference's 1:2 proportion (100:200 default
epochs, 1d-burgers/inf_cont_burgers.py:35-41).  Inputs are resident in HBM before the timed
region.  W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
stream sync, MAX over ranks; rank 0 prints one JSON line.

Launched by the driver for N>1 as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
(one process per GPU; gloo is used only to exchange the RCCL unique id and the timings,
the gradient all-reduce itself is RCCL inside the engine).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pinns-tf2.0_amd")
for p in (ROOT, PKG, os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers")):
    if p not in sys.path:
        sys.path.insert(0, p)

LAYERS = [2, 20, 20, 20, 20, 20, 20, 20, 20, 1]
M_W = sum(a * b for a, b in zip(LAYERS[:-1], LAYERS[1:]))          # 2860 MACs per channel
NU = 0.01 / np.pi
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}                          # MI355X_MICROARCH.md (vector = matrix peak)


def shard(n, world, rank):
    from pinn_native.parallel import shard_bounds
    return shard_bounds(n, world, rank)


def canonical_weights():
    from scipy.stats import truncnorm
    rs = np.random.RandomState(1234)
    parts = []
    for fi, fo in zip(LAYERS[:-1], LAYERS[1:]):
        sigma = np.sqrt(2.0 / (fi + fo)) / 0.87962566103423978
        parts.append((truncnorm.rvs(-2, 2, size=(fi, fo), random_state=rs) * sigma).ravel())
        parts.append(np.zeros(fo))
    return np.concatenate(parts)


def make_engine(dtype, device, X_f, X_u, u, lb, ub, world, rank, n_f_total, n_u_total):
    import pinn_native
    eng = pinn_native.Engine(LAYERS, lb, ub, pde="burgers", dtype=dtype, device=device)
    from pinn_native.parallel import attach_shards
    attach_shards(eng, world, rank, X_f=X_f, X_u=X_u, u=u)
    eng.set_pde_params(NU)
    return eng


def cpu_baseline(X_f, X_u, u, lb, ub, w0, budget_s=12.0):
    """The oracle (numpy f64 port of the reference path) timed on this host: Adam iterations on
    the same N_f=10000 workload until ~budget_s of CPU time is spent."""
    from oracle import pde, optim
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    adam = optim.Adam(0.03, 0.9, 0.999, None)
    w = w0.copy()
    pde.burgers_loss_grad(w, LAYERS, lb, ub, X_f, X_u, u, NU)          # warm the BLAS threads
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < 400:
        _, g, _ = pde.burgers_loss_grad(w, LAYERS, lb, ub, X_f, X_u, u, NU)
        w = adam.step(w, g)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": X_f.shape[0] * n / dt, "unit": "collocation-points/s", "cores": int(threads),
            "kind": "port",
            "sample": "%d Adam iterations of oracle/ (numpy float64) on N_f=%d, N_u=%d, 8x20 "
                      "MLP in %.1f s" % (n, X_f.shape[0], X_u.shape[0], dt)}


def pmc_traffic(args, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE are collected in separate runs, so they cannot be read live here);
    valid for the default workload only, null otherwise."""
    if args.dtype != "f32" or args.nf_per_gpu != 10000 or args.kernel_path not in (-1, 2):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            return float(json.load(fh)["traffic_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--dtype", default=os.environ.get("PINN_BENCH_DTYPE", "f32"), choices=["f32", "f64"])
    ap.add_argument("--nf-per-gpu", type=int, default=10000)
    ap.add_argument("--kernel-path", type=int, default=-1, help="-1 engine default, 0 generic, 1 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-final-error", action="store_true")
    ap.add_argument("--no-f64-leg", action="store_true", help="skip the float64 (reference arithmetic) timing leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("PINN_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # override: tests on one GPU
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "(--nproc-per-node %d)" % (args.gpus, args.gpus))
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import burgersutil
    import pinn_native
    n_f_total = args.nf_per_gpu * world
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat"),
                              100, n_f_total, noise=0.0)
    X_star, u_star, X_u, u, X_f, ub, lb = r[5], r[6], r[7], r[8], r[9], r[10], r[11]
    w0 = canonical_weights()

    eng = make_engine(args.dtype, local_rank, X_f, X_u, u, lb, ub, world, rank, n_f_total, 100)
    if args.kernel_path >= 0:
        eng.set_kernel_path(args.kernel_path)
    comm_mode = "none"
    if world > 1:
        from pinn_native.parallel import init_engine_comm
        comm_mode = init_engine_comm(eng, dist, world, rank)     # "mailbox" if every rank's self-test passed, else "rccl"

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()
        eng.sync()

    k_adam = args.steps // 3
    k_lbfgs = args.steps - k_adam
    eng.set_weights(w0)
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    # ---- warm-up (untimed): W optimiser iterations in the same 1:2 Adam:L-BFGS mix, so that every
    # kernel of the timed region has been loaded and every device buffer allocated beforehand
    if args.warmup > 0:
        # a fresh box idles at 570 MHz: keep the GPU busy for ~0.3 s first so that the W warm-up steps and the timed
        # region run at the sustained clock (extra untimed work only; the timed region is still exactly K steps)
        # (a fixed number of steps, not a time limit: with a communicator every rank must run the same evaluations)
        eng.adam_run(6000, want_losses=False)
        eng.sync()
        eng.set_weights(w0)
        eng.adam_init(0.03, 0.9, 0.999, 1e-7)
        w_adam = max(args.warmup // 3, 1)
        w_lbfgs = max(args.warmup - w_adam, 2)
        eng.adam_run(w_adam, want_losses=False)
        eng.lbfgs_begin(max(k_lbfgs, w_lbfgs), 0.8, 50, float(np.finfo(float).eps))
        eng.lbfgs_run(w_lbfgs)
    eng.set_weights(w0)
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    eng.timing_enable(args.steps, every=16)          # sampled: an event record costs ~5 us of GPU time
    # ---- timed region: exactly K optimiser iterations = K loss+grad evaluations ----------------
    done = 0
    barrier()
    t0 = time.perf_counter()
    if k_adam:
        eng.adam_run(k_adam, want_losses=False)
    if k_lbfgs:
        eng.lbfgs_begin(k_lbfgs, 0.8, 50, float(np.finfo(float).eps))      # initial evaluation
        done = 0
        while not done:
            _, _, done = eng.lbfgs_run(k_lbfgs)
    barrier()
    elapsed = time.perf_counter() - t0
    tim = eng.timing_read()
    eng.timing_enable(0, 1)
    replicas_identical = None
    if dist is not None:
        import hashlib
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
        # every rank applied the same all-reduced gradients: the weight replicas must agree bit for bit
        digests = [None] * world
        dist.all_gather_object(digests, hashlib.sha256(eng.get_weights().tobytes()).hexdigest())
        replicas_identical = all(d == digests[0] for d in digests)

    # ---- accuracy leg (untimed): the reference's default schedule ON THE REFERENCE CONFIGURATION (N_f = 10000 in
    # total, seed 1234 -- sharded over the ranks when N > 1), final relative L2 error of u over the 25600-point grid.
    # The weak-scaling workload above has N x 10000 points, i.e. another training set with no reference value; and the
    # schedule (Adam lr 0.03, L-BFGS without line search) is roundoff-chaotic, so the number is only comparable on the
    # reference's own points.
    final_err = None
    if not args.no_final_error:
        if n_f_total != 10000:
            from pinn_native.parallel import attach_shards
            np.random.seed(1234)
            r2 = burgersutil.prep_data(os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
            attach_shards(eng, world, rank, X_f=r2[9], X_u=r2[7], u=r2[8])
        eng.set_weights(w0)
        eng.adam_init(0.03, 0.9, 0.999, 1e-7)
        eng.adam_run(100, want_losses=False)
        eng.lbfgs_begin(200, 0.8, 50, float(np.finfo(float).eps))
        d2 = 0
        while not d2:
            _, _, d2 = eng.lbfgs_run(200)
        u_pred = eng.predict(X_star)
        final_err = float(np.linalg.norm(u_star - u_pred, 2) / np.linalg.norm(u_star, 2))

    # ---- float64 leg (untimed for `value`; N=1 only): the same K iterations in the reference's arithmetic ----
    f64_leg = None
    if world == 1 and args.dtype == "f32" and not args.no_f64_leg:
        e64 = make_engine("f64", local_rank, X_f, X_u, u, lb, ub, world, rank, n_f_total, 100)
        e64.set_weights(w0)
        e64.adam_init(0.03, 0.9, 0.999, 1e-7)
        e64.adam_run(max(args.warmup // 3, 1), want_losses=False)
        e64.lbfgs_begin(max(k_lbfgs, 4), 0.8, 50, float(np.finfo(float).eps))
        e64.lbfgs_run(max(args.warmup - args.warmup // 3, 2))
        e64.set_weights(w0)
        e64.adam_init(0.03, 0.9, 0.999, 1e-7)
        e64.sync()
        t1 = time.perf_counter()
        if k_adam:
            e64.adam_run(k_adam, want_losses=False)
        if k_lbfgs:
            e64.lbfgs_begin(k_lbfgs, 0.8, 50, float(np.finfo(float).eps))
            d3 = 0
            while not d3:
                _, _, d3 = e64.lbfgs_run(k_lbfgs)
        e64.sync()
        el64 = time.perf_counter() - t1
        f64_leg = {"value": n_f_total * args.steps / el64, "unit": "collocation-points/s",
                   "ms_per_step": 1e3 * el64 / args.steps, "dtype": "f64", "kernel_path": e64.kernel_path()}
        e64.close()

    # every rank empties its C stdio buffer (RCCL's banner) before rank 0 prints: the JSON line stays the last line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        n_f_local = shard(n_f_total, world, 0)[1]
        n_u_local = shard(100, world, 0)[1]
        flops_per_eval = 24.0 * M_W * n_f_local + 6.0 * M_W * n_u_local    # SURVEY.md 8(d), per rank
        # HIP events bracket the kernel on the engine's stream; an empty bracket already reads a few
        # microseconds, so the kernel duration is the bracket minus that calibrated constant
        # (path 2: the events are attached to the kernel launch itself and read its begin/end timestamps -- exact)
        kernel_ms = tim["fwd_ms"] if tim["kernel_exact"] else max(tim["sweeps_ms"] - tim["empty_bracket_ms"], 0.0)
        sweeps_s = kernel_ms * 1e-3
        achieved = flops_per_eval / sweeps_s / 1e12 if sweeps_s > 0 else None
        peak = PEAK_TFLOPS[args.dtype]
        out = {
            "metric": "collocation-points/sec",
            "value": n_f_total * args.steps / elapsed,
            "unit": "collocation-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "1D Burgers continuous inference (BASELINE configs[1]): 8x20 tanh "
                                   "MLP, N_u=100, N_f=%d per GPU (LHS, seed 1234), %d Adam + %d L-BFGS "
                                   "iterations, canonical glorot init" % (args.nf_per_gpu, k_adam, k_lbfgs),
                       "n_f_total": n_f_total, "n_u": 100, "parallelism": "dp%d" % world, "allreduce": comm_mode,
                       "allreduce_probe_us": getattr(eng, "comm_probe_us", None),
                       "replicas_identical": replicas_identical,
                       "kernel_path": eng.kernel_path(), "lbfgs_done_code": int(done) if k_lbfgs else None},
            "float64_leg": f64_leg,
            "final_l2_error": final_err,
            "final_l2_error_schedule": "100 Adam (lr .03) + 200 L-BFGS (lr .8, m=50), reference defaults, on the "
                                       "reference set N_f=10000 (sharded over the ranks); reference run: 0.2656",
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": pmc_traffic(args, world),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json)",
                         "hbm_gbps": (pmc_traffic(args, world) / sweeps_s / 1e9) if (pmc_traffic(args, world) and sweeps_s > 0) else None,
                         "hbm_peak_gbps": 8000.0,
                         "kernel": {2: "pinn::k_fused20m", 1: "pinn::k_fused20", 0: "pinn::k_forward+k_backward"}[eng.kernel_path()],
                         "avg_launch_ms": kernel_ms, "avg_launch_method": "hipExtLaunchKernelGGL start/stop events" if tim["kernel_exact"] else "event bracket minus empty bracket", "event_bracket_ms": tim["sweeps_ms"],
                         "empty_event_bracket_ms": tim["empty_bracket_ms"], "evals_timed": tim["n"],
                         "algorithmic_flop_per_launch": flops_per_eval,
                         "eval_ms_incl_reduce_allreduce": tim["eval_ms"]},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(X_f[:args.nf_per_gpu], X_u, u, lb, ub, w0)
        else:
            out["cpu_baseline"] = None
        # RCCL writes its version banner through C stdio: flush that first, so the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
