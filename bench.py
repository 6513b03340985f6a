#!/usr/bin/env python3
"""bench.py -- collocation-points/sec of the PINN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f64] [--nf-total M | --nf-per-gpu M]

A *step* is one optimiser iteration = one full-batch pass of the hot path (Taylor-mode forward over every
collocation point, PDE residual, loss reduction, flat gradient, all-reduce when N > 1, optimiser update).
The headline workload is the configuration the metric is quoted on, BASELINE.json configs[1]: 1D Burgers continuous
inference, 8x20 tanh MLP, N_u = 100, N_f = 10000 IN TOTAL (at N > 1 the 10000 points are split over the ranks:
strong scaling, honest for a 40-microsecond step), Adam then L-BFGS in the reference's 1:2 proportion (100:200
default epochs, 1d-burgers/inf_cont_burgers.py:35-41), canonical glorot init, inputs resident in HBM.

Timing: W untimed warm-up steps; then blocks of EXACTLY K steps, each bracketed by barrier + stream sync on both sides
(barrier + sync before, sync + MAX-over-ranks reduction after) ; blocks are repeated (same initial state each time, reset outside the bracket) until
>= 3 s have been timed per leg (PINN_BENCH_MIN_TIMED_MS), and the MEDIAN block is reported -- a single 20-step block
is 1 ms, below the noise of a fresh box, and 50 ms per leg (rounds 1-2) was too short for the driver's 5-second
GPU-busy sampler to corroborate.  `value` = N_f_total x K / median block.  The kernel duration behind `roofline` is measured live in a
separate pass of the same steps with HIP events attached to the launches themselves (>= 32 samples).

One JSON line on rank 0.  The headline is float64 -- the reference's arithmetic (utils/neuralnetwork.py:24-26) and
the product's default (hp["dtype"]); --dtype f32 swaps the roles.  Beside it the line carries
  float32_leg   the same K steps on the float32 kernels (the FP32 mode north_star sanctions), with its own roofline
                (float64_leg under --dtype f32)
  cfg5_leg      BASELINE configs[4]: N_f = 10^6 in total, sharded over the N ranks (125k per GPU at N = 8); at N = 1
                this is the steady-state (many tiles per CU) figure of the same kernel
  cfg3_leg      BASELINE configs[2]: Burgers identification (1d-burgers/ide_cont_burgers.py:25-43), N_u = 10000 data points
                carrying the residual, lambda_1 / lambda_2 trainable, Adam (lr 1e-3) : L-BFGS in the reference's 100 : 500
  cfg4_leg      BASELINE configs[3]: Schrodinger (1dcomplex-schrodinger/inf_cont_schrodinger.py:19-41), 4x100 two-output
                net, N_f = 20000, N_0 = N_b = 50, Adam (lr .05, beta_1 .99, eps .1) only, as the reference
                (both in the headline's arithmetic, each with its own roofline, >= 1 s timed)
  final_l2_error{,_f64}   the reference's default schedule end to end in both arithmetics, beside the reference's own
                ulp-perturbation ensemble (tests/golden/burgers_band.json)
  script_leg    the step a user of the drop-in script sees: wall time of fit() of pinns-tf2.0_amd/1d-burgers/inf_cont_burgers.py's
                own class on the default schedule, progress lines every 10 epochs and the error metric included
  cpu_baseline  kind "port": oracle/fit.py (numpy restatement of NeuralNetwork.fit, logging included) timed on this host's
                cores on a bounded sample; the reference's own timing from the build container rides along as data

Launched by the driver for N > 1 as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
(one process per GPU; gloo carries the RCCL unique id and the timing reductions only, the gradient all-reduce itself
is RCCL inside the engine).  Plain `python bench.py --gpus N ...` is equivalent: with no RANK / WORLD_SIZE in the environment
the script starts its N ranks itself (self_launch) and forwards rank 0's line; `launch` in the line says which form ran,
`runtime` which HIP runtime / RCCL build the rank bound (pinn_native._bind_runtime).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pinns-tf2.0_amd")
for p in (ROOT, PKG, os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers")):
    if p not in sys.path:
        sys.path.insert(0, p)

LAYERS = [2, 20, 20, 20, 20, 20, 20, 20, 20, 1]
LAYERS_SCHRODINGER = [2, 100, 100, 100, 100, 2]


def macs(layers):
    return sum(a * b for a, b in zip(layers[:-1], layers[1:]))


M_W = macs(LAYERS)                                                 # 2860 MACs per Taylor channel (Schrodinger: 30400)
NU = 0.01 / np.pi
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}                          # MI355X_MICROARCH.md: vector = matrix FP32/FP64 peak
HBM_PEAK_GBPS = 8000.0
MIN_TIMED_MS = float(os.environ.get("PINN_BENCH_MIN_TIMED_MS", "3000"))   # per leg: long enough for the driver's
MAX_BLOCKS = 8000                                                         # 5-second GPU-busy sampler to see the legs
KERNEL_NAMES = {2: "pinn::k_fused20m", 1: "pinn::k_fused20", 7: "pinn::k_fused20d", 0: "pinn::k_forward+k_backward",
                3: "pinn::k_wide_fwd+k_wide_bwd", 8: "pinn::k_t16_fused"}
BURGERS_ADAM = (0.03, 0.9, 0.999, 1e-7)                            # 1d-burgers/inf_cont_burgers.py:35-37 (eps None = Keras 1e-7)
BURGERS_LBFGS = (0.8, 50)                                          # :39-41
LAUNCH_FLOOR_US = 4.5                                              # a trivial launch on this stream


def canonical_weights(layers=None, extra=()):
    from scipy.stats import truncnorm
    layers = layers or LAYERS
    rs = np.random.RandomState(1234)
    parts = []
    for fi, fo in zip(layers[:-1], layers[1:]):
        sigma = np.sqrt(2.0 / (fi + fo)) / 0.87962566103423978
        parts.append((truncnorm.rvs(-2, 2, size=(fi, fo), random_state=rs) * sigma).ravel())
        parts.append(np.zeros(fo))
    parts.append(np.asarray(extra, dtype=np.float64))
    return np.concatenate(parts)


def burgers_workload(data):
    """BASELINE configs[1] / [4]: the sets of prep_data, nu, the reference's optimiser settings"""
    X_f, X_u, u, lb, ub = data
    return {"layers": LAYERS, "pde": "burgers", "lb": lb, "ub": ub, "sets": {"X_f": X_f, "X_u": X_u, "u": u},
            "pde_params": (NU,), "adam": BURGERS_ADAM, "lbfgs": BURGERS_LBFGS, "points": len(X_f),
            # 24 M_w per collocation point (4 Taylor channels x (forward + 2 x reverse) x 2 FLOP), 6 M_w per data point
            "flops": lambda n: 24.0 * M_W * n["f"] + 6.0 * M_W * n["u"]}


def make_engine(dtype, device, wl, world, rank):
    import pinn_native
    from pinn_native.parallel import attach_shards
    eng = pinn_native.Engine(wl["layers"], wl["lb"], wl["ub"], pde=wl["pde"], dtype=dtype, device=device)
    attach_shards(eng, world, rank, **wl["sets"])
    if wl.get("pde_params"):
        eng.set_pde_params(*wl["pde_params"])
    return eng


class World(object):
    """the three collective things the timing needs; trivial at world size 1"""

    def __init__(self, dist, world, rank):
        self.dist, self.world, self.rank = dist, world, rank

    def barrier(self, eng):
        eng.sync()
        if self.dist is not None:
            self.dist.barrier()
        eng.sync()

    def max(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])


def reset(eng, w0, adam=BURGERS_ADAM):
    eng.set_weights(w0)
    eng.adam_init(*adam)


def run_steps(eng, k_adam, k_lbfgs, lbfgs=BURGERS_LBFGS):
    """exactly k_adam + k_lbfgs optimiser iterations = as many loss+grad evaluations; -> L-BFGS done code"""
    done = 1
    if k_adam:
        eng.adam_run(k_adam, want_losses=False)
    if k_lbfgs:
        eng.lbfgs_begin(k_lbfgs, lbfgs[0], lbfgs[1], float(np.finfo(float).eps))      # the initial evaluation
        done = 0
        while not done:
            _, _, done = eng.lbfgs_run(k_lbfgs)
    return int(done)


def time_blocks(eng, wd, w0, k_adam, k_lbfgs, min_ms=MIN_TIMED_MS, max_blocks=MAX_BLOCKS, adam=BURGERS_ADAM,
                lbfgs=BURGERS_LBFGS):
    """blocks of exactly K steps, barrier + sync on both sides, MAX over ranks; repeated until >= min_ms are timed.
    Every rank sees the same (max-reduced) block times, so every rank runs the same number of blocks."""
    times, done = [], 1
    while (sum(times) * 1e3 < min_ms and len(times) < max_blocks) or not times:
        reset(eng, w0, adam)
        wd.barrier(eng)                                    # everybody starts together ...
        t0 = time.perf_counter()
        done = run_steps(eng, k_adam, k_lbfgs, lbfgs)
        eng.sync()                                         # ... this rank's K steps are complete on its GPU ...
        dt = time.perf_counter() - t0
        times.append(wd.max(dt))                           # ... and the block lasts as long as the slowest rank
        # (the MAX all-reduce is also the closing barrier; it sits outside every rank's own interval, so a host-side
        #  gloo round trip of a few hundred microseconds does not inflate a 1-ms block)
    return times, done


def kernel_samples(eng, w0, k_adam, k_lbfgs, want=32, adam=BURGERS_ADAM, lbfgs=BURGERS_LBFGS):
    """the loss+grad kernel's own duration: HIP events attached to the launches (separate, untimed pass)"""
    evals = k_adam + k_lbfgs
    n_blocks = max(1, -(-want // max(evals, 1)))
    eng.timing_enable(n_blocks * (evals + 1), every=1)
    for _ in range(n_blocks):
        reset(eng, w0, adam)
        run_steps(eng, k_adam, k_lbfgs, lbfgs)
    tim = eng.timing_read()
    eng.timing_enable(0, 1)
    return tim


def roofline(eng, tim, dtype, n_f_local, n_u_local, traffic=None, flops=None, n_b_local=0):
    if flops is None:
        flops = 24.0 * M_W * n_f_local + 6.0 * M_W * n_u_local       # algorithmic FLOP per launch
    kernel_ms = tim["fwd_ms"] if tim["kernel_exact"] else max(tim["sweeps_ms"] - tim["empty_bracket_ms"], 0.0)
    achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else None
    peak = PEAK_TFLOPS[dtype]
    info = {}
    try:
        import pinn_native
        info = pinn_native.device_info(0)
    except Exception:
        pass
    n_cu = info.get("compute_units", 256)
    tiles = (n_f_local + n_u_local + 2 * n_b_local + 63) // 64
    path = eng.kernel_path()
    single_kernel = path in (1, 2, 7)
    wgs = min(tiles, n_cu) if path in (2, 7) else tiles
    if path in (3, 8):                       # persistent workgroups over 16-point groups, one per CU
        wgs = min(4 * tiles, n_cu)
    return {
        "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
        "frac": (achieved / peak) if achieved else None,
        "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, profiles/)",
        "hbm_gbps": (traffic / (kernel_ms * 1e-3) / 1e9) if (traffic and kernel_ms > 0) else None,
        "hbm_peak_gbps": HBM_PEAK_GBPS,
        "algorithmic_flop_per_launch": flops,
        "algorithmic_hbm_bytes_per_launch": (4 if dtype == "f32" else 8) * 2 * (n_f_local + n_u_local + 2 * n_b_local),
        "kernel": KERNEL_NAMES.get(path, "pinn::k_t16_fwd+k_t16_bwd"),
        "avg_launch_ms": kernel_ms, "launches_sampled": tim["n"],
        "avg_launch_method": "hipExtLaunchKernelGGL start/stop events on the engine's stream" if tim["kernel_exact"]
                             else "event bracket minus empty bracket",
        "eval_ms_incl_reduce_allreduce": tim["eval_ms"],
        # which regime the launch is in: with fewer 64-point tiles than CUs the launch is ONE tile deep and its duration
        # is the latency of a single workgroup (neither roof is reachable); many tiles per CU = throughput regime
        "workgroups": wgs, "compute_units": n_cu, "tiles_per_workgroup": tiles / max(wgs, 1),
        "regime": ("latency: %d workgroups on %d CUs, one tile deep" % (wgs, n_cu)) if (single_kernel and tiles <= n_cu)
                  else "throughput: %.1f tiles (64 points) per workgroup" % (tiles / max(wgs, 1)),
        "launch_floor_us": LAUNCH_FLOOR_US,
    }


def leg(name, dtype, device, data, w0, wd, k_adam, k_lbfgs, warmup, kernel_path=-1, traffic=None, spin=True,
        init_comm=None, min_ms=None):
    """one timed leg on one workload (a dict as burgers_workload makes it, or the Burgers data tuple); -> (result dict, engine)"""
    wl = data if isinstance(data, dict) else burgers_workload(data)
    sets, adam, lbfgs = wl["sets"], wl["adam"], wl["lbfgs"] or BURGERS_LBFGS
    if not wl["lbfgs"]:                                    # an Adam-only schedule (the reference's Schrodinger script)
        k_adam, k_lbfgs = k_adam + k_lbfgs, 0
    eng = make_engine(dtype, device, wl, wd.world, wd.rank)
    if kernel_path >= 0:
        eng.set_kernel_path(kernel_path)
    comm_mode = init_comm(eng) if init_comm else "none"
    reset(eng, w0, adam)
    n_pts = wl["points"]
    if warmup > 0:
        if spin:
            # a fresh box idles at a low clock: keep the GPU busy for a few tenths of a second first (untimed; a
            # fixed number of steps, not a time limit: with a communicator every rank must run the same evaluations)
            eng.adam_run(max(200, int(6000 * 10000 / max(n_pts, 1))), want_losses=False)
            eng.sync()
            reset(eng, w0, adam)
        w_adam = max(warmup // 3, 1) if wl["lbfgs"] else warmup
        eng.adam_run(w_adam, want_losses=False)
        if wl["lbfgs"]:
            w_lbfgs = max(warmup - w_adam, 2)
            eng.lbfgs_begin(max(k_lbfgs, w_lbfgs), lbfgs[0], lbfgs[1], float(np.finfo(float).eps))
            eng.lbfgs_run(w_lbfgs)
    times, done = time_blocks(eng, wd, w0, k_adam, k_lbfgs, min_ms=MIN_TIMED_MS if min_ms is None else min_ms, adam=adam,
                              lbfgs=lbfgs)
    tim = kernel_samples(eng, w0, k_adam, k_lbfgs, adam=adam, lbfgs=lbfgs)
    from pinn_native.parallel import shard_bounds

    def local(key):
        lo, hi = shard_bounds(len(sets[key]), wd.world, 0) if key in sets else (0, 0)
        return hi - lo
    n_local = {"f": local("X_f"), "u": local("X_u"), "b": local("X_lb")}
    K = k_adam + k_lbfgs
    med = float(np.median(times))
    out = {
        "name": name, "dtype": dtype, "n_f_total": int(n_pts), "n_f_per_gpu": n_local["f"] or n_local["u"],
        "value": n_pts * K / med if done == 1 else None, "unit": "collocation-points/s",
        "ms_per_step": 1e3 * med / K, "steps_per_block": K, "adam_steps_per_block": k_adam, "lbfgs_steps_per_block": k_lbfgs,
        "blocks_timed": len(times),
        "timed_ms_total": 1e3 * float(np.sum(times)), "ms_per_step_first_block": 1e3 * times[0] / K,
        "ms_per_step_min_block": 1e3 * float(np.min(times)) / K,
        "kernel_path": eng.kernel_path(), "lbfgs_done_code": done, "valid": done == 1,
        "allreduce": comm_mode, "allreduce_probe_us": getattr(eng, "comm_probe_us", None),
        "allreduce_fallback": getattr(eng, "comm_fallback", None),       # "rccl failed: ..." when the mailboxes took over
        "roofline": roofline(eng, tim, dtype, n_local["f"], n_local["u"], traffic, flops=wl["flops"](n_local),
                             n_b_local=n_local["b"]),
    }
    return out, eng


def final_error(eng, w0, X_star, u_star):
    """the reference's default schedule (100 Adam lr .03 + 200 L-BFGS lr .8 m 50), then its error metric
    (inf_cont_burgers.py:114-116)"""
    reset(eng, w0)
    eng.adam_run(100, want_losses=False)
    eng.lbfgs_begin(200, 0.8, 50, float(np.finfo(float).eps))
    d = 0
    while not d:
        _, _, d = eng.lbfgs_run(200)
    return float(eng.error_l2(X_star, u_star))         # device-side reduction (pinn_error_l2)


def reference_ensemble(dtype="f64"):
    """the reference's own final errors under perturbations of the initial weights (make_band.py): (1 + k 2^-52) for the
    float64 engine, (1 + k 2^-23) -- the size of a float32 rounding -- for the float32 engine, as the tests judge them"""
    name = "burgers_band.json" if dtype == "f64" else "burgers_band_eps32.json"
    try:
        with open(os.path.join(ROOT, "tests", "golden", name)) as fh:
            b = json.load(fh)
        errs = sorted(v["final_error"] for v in b["runs"].values())
        med = float(np.median(errs))
        return {"reference": b["reference_final_error"], "ensemble_min": errs[0], "ensemble_max": errs[-1],
                "ensemble_median": med, "ensemble_radius": float(max(abs(e - med) for e in errs)),
                "members": len(errs), "source": "tests/golden/%s (reference over the shim, init x (1 + k 2^%d))" % (
                    name, -52 if dtype == "f64" else -23)}
    except Exception:
        return None


def cpu_baseline_port(X_f, X_u, u, lb, ub, w0, X_star, u_star, budget_s=20.0):
    """cpu_baseline, kind "port": oracle/fit.py -- the numpy float64 restatement of the reference's NeuralNetwork.fit
    (utils/neuralnetwork.py:138-149: Adam loop, custom_lbfgs, a progress line every 10 epochs, the error metric once at
    the end) -- timed on THIS host's cores on a bounded sample of the headline workload: the reference's schedule in
    its 1:2 Adam : L-BFGS proportion, shortened so that the run takes about budget_s seconds (a probe of three
    evaluations sizes it).  The reference itself is Python and does not travel to this box; its own timing, taken in
    the build container, is carried beside this as `reference_in_build_container` (oracle/ref_baseline.py)."""
    from oracle import pde, fit
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    pde.burgers_loss_grad(w0, LAYERS, lb, ub, X_f, X_u, u, NU)          # warm the BLAS threads
    t0 = time.perf_counter()
    for _ in range(3):
        pde.burgers_loss_grad(w0, LAYERS, lb, ub, X_f, X_u, u, NU)
    per_eval = (time.perf_counter() - t0) / 3
    tf_ep = int(min(100, max(10, round(budget_s / per_eval / 3 / 10) * 10)))
    res = fit.burgers_fit(w0, LAYERS, lb, ub, X_f, X_u, u, NU, X_star, u_star, tf_epochs=tf_ep, nt_epochs=2 * tf_ep)
    out = {"value": X_f.shape[0] * res["evals"] / res["fit_seconds"], "unit": "collocation-points/s", "cores": int(threads),
           "kind": "port", "host_cores": os.cpu_count(), "ms_per_step": 1e3 * res["fit_seconds"] / res["evals"],
           "sample": "oracle/fit.py (numpy float64 port of NeuralNetwork.fit, progress line every 10 epochs, error metric "
                     "at the end): %d Adam + %d L-BFGS iterations = %d loss+grad evaluations on N_f=%d, N_u=%d, 8x20 MLP "
                     "in %.1f s, %d BLAS threads" % (tf_ep, 2 * tf_ep, res["evals"], X_f.shape[0], X_u.shape[0],
                                                     res["fit_seconds"], threads),
           "progress_lines": len([l for l in res["lines"] if l.startswith(("tf_epoch", "nt_epoch"))])}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "cpu_reference_timing.json")) as fh:
            r = json.load(fh)
        out["reference_in_build_container"] = {
            "value": r["value"], "unit": r["unit"], "cores": r["threads"], "fit_seconds": r["fit_seconds"],
            "evals": r["evals"], "host": r["host"], "final_l2_error": r["final_l2_error"],
            "note": "the reference's own script over the torch-CPU tensorflow stand-in, timed where /root/reference exists "
                    "(tests/golden/cpu_reference_timing.json <- oracle/ref_baseline.py --write-fixture); a different host "
                    "from this box: shown for scale, not used in any ratio"}
    except Exception:
        pass
    return out


def script_leg(dtype, device, reps=5):
    """the step as a USER of the drop-in sees it -- wall time of NeuralNetwork.fit of the unmodified drop-in
    script's own class (pinns-tf2.0_amd/1d-burgers/inf_cont_burgers.py: BurgersInformedNN, Logger, prep_data) on the default
    schedule (100 Adam + 200 L-BFGS), log_frequency = 10 with every progress line formatted and written, the error
    metric evaluated by log_train_end -- exactly what cpu_baseline times on the CPU (utils/neuralnetwork.py:138-149,
    utils/logger.py:45-60).  A fresh model per repetition (construction and data preparation outside the timed call,
    as in cpu_baseline); the median over `reps` fits is reported, the first (cold) one beside it."""
    import contextlib
    import importlib.util
    import io
    spec = importlib.util.spec_from_file_location("pinn_dropin_inf_cont_burgers",
                                                  os.path.join(PKG, "1d-burgers", "inf_cont_burgers.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, sys.argv[:1]                   # the script reads an hp file from argv[1]
    try:
        spec.loader.exec_module(mod)                          # defines hp, BurgersInformedNN, run(); runs nothing
    finally:
        sys.argv = argv
    hp = dict(mod.hp, dtype=dtype, device=device)
    np.random.seed(1234)
    r = mod.prep_data(os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat"), hp["N_u"], hp["N_f"], noise=0.0)
    X_star, u_star, X_u, u, X_f, ub, lb = r[5], r[6], r[7], r[8], r[9], r[10], r[11]
    steps = hp["tf_epochs"] + hp["nt_epochs"]
    times, parts, lines, err = [], [], 0, None
    for _ in range(reps):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mod.set_seed(1234)
            logger = mod.Logger(hp)
            pinn = mod.BurgersInformedNN(hp, logger, X_f, ub, lb, nu=NU)
            logger.set_error_fn(lambda: pinn.error_l2(X_star, u_star))
            split = {}
            for name in ("tf_optimization", "nt_optimization"):
                def timed(*a, _f=getattr(pinn, name), _n=name, **k):
                    t = time.perf_counter()
                    out = _f(*a, **k)
                    split[_n] = time.perf_counter() - t
                    return out
                setattr(pinn, name, timed)
            pinn._engine.sync()
            t0 = time.perf_counter()
            pinn.fit(X_u, u)
            times.append(time.perf_counter() - t0)
        split["log_train_end"] = times[-1] - split["tf_optimization"] - split["nt_optimization"]
        parts.append(split)
        text = buf.getvalue()
        lines = len([l for l in text.splitlines() if l.startswith(("tf_epoch", "nt_epoch"))])
        err = float(text.split("error = ")[1].split()[0]) if "error = " in text else None
        pinn._engine.close()
    med = float(np.median(times))
    k = int(np.argsort(times)[len(times) // 2])
    return {"name": "script", "dtype": dtype, "what": "wall time of BurgersInformedNN.fit (the drop-in 1d-burgers/inf_cont_burgers.py, "
            "default hp: %d Adam + %d L-BFGS, log_frequency %d, error metric at the end), fresh model per repetition"
            % (hp["tf_epochs"], hp["nt_epochs"], hp["log_frequency"]),
            "steps": steps, "reps": reps, "fit_ms": 1e3 * med, "fit_ms_first": 1e3 * times[0], "fit_ms_min": 1e3 * min(times),
            "ms_per_step": 1e3 * med / steps, "value": hp["N_f"] * steps / med, "unit": "collocation-points/s",
            "progress_lines_written": lines, "final_l2_error_printed": err,
            "split_ms": {n: 1e3 * v for n, v in parts[k].items()}}


def pmc_traffic(name, dtype, n_f_total, world, path):
    """HBM bytes per launch of a leg's loss+gradient kernel(s) from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs under the profiler, profiles/pmc_eval.py, so they cannot be read live
    here) -> (bytes or None, where the number comes from / why there is none).  An entry of profiles/pmc_traffic.json
    applies only to the workload, arithmetic, kernel path and point count it was collected on, on one GPU."""
    table = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world != 1:
        return None, "the counter passes were collected on one GPU; this run shards the points over %d" % world, None
    try:
        with open(table) as fh:
            entries = json.load(fh)["entries"]
    except Exception as e:
        return None, "profiles/pmc_traffic.json unreadable: %s" % e, None
    for e in entries:
        if (e["leg"], e["dtype"], e["kernel_path"], e["points"]) == (name, dtype, path, n_f_total):
            return (float(e["traffic_bytes_per_launch"]), "profiles/pmc_traffic.json <- %s" % e["source"],
                    e.get("sources_sha256"))
    return None, ("no counter pass was collected for leg %s, %s, kernel path %d, %d points (profiles/pmc_traffic.json)"
                  % (name, dtype, path, n_f_total)), None


def with_traffic(leg_dict, world, name=None):
    """attach the PMC traffic (and the HBM rate it implies) to a leg's roofline"""
    rf = leg_dict["roofline"]
    key = name or {"float32": "headline", "float64": "headline"}.get(leg_dict["name"], leg_dict["name"])
    rf["traffic"], rf["traffic_source"], collected_on = pmc_traffic(key, leg_dict["dtype"], leg_dict["n_f_total"], world,
                                                                    leg_dict["kernel_path"])
    # the counters are NOT read in this run (rocprofv3 --pmc needs its own passes, MI355X_MICROARCH.md): the figure is the
    # committed pass of the same launch (profiles/collect_pmc.py), looked up in profiles/pmc_traffic.json.  Every entry
    # names the digest of the kernel sources of the library it was collected on: library_matches says whether that is
    # the library this run executes.
    running = None
    try:
        import pinn_native
        running = pinn_native.library_digest() or pinn_native._source_digest()
    except Exception:
        pass
    rf["traffic_provenance"] = {"measured_in_run": False, "file": "profiles/pmc_traffic.json" if rf["traffic"] else None,
                                "collected_on_sources_sha256": collected_on, "running_sources_sha256": running,
                                "library_matches": bool(collected_on and running and collected_on == running)}
    if rf["traffic"] and rf["avg_launch_ms"]:
        rf["hbm_gbps"] = rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9
    return leg_dict


def identification_workload():
    """BASELINE configs[2] (1d-burgers/ide_cont_burgers.py:25-43): prep_data's identification branch with
    N_u = 10000 samples of the full field (they carry data misfit AND residual), lambda = (0, -6) appended to the
    canonical weights, Adam lr 1e-3, L-BFGS lr .8 / 50 pairs; the reference runs 100 : 500 iterations"""
    import burgersutil
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat"), 10000, noise=0.0)
    X_u, u, ub, lb = r[7], r[8], r[9], r[10]
    return {"layers": LAYERS, "pde": "burgers_ide", "lb": lb, "ub": ub, "sets": {"X_u": X_u, "u": u}, "pde_params": None,
            "adam": (0.001, 0.9, 0.999, 1e-7), "lbfgs": (0.8, 50), "points": len(X_u), "adam_share": 1.0 / 6.0,
            "w0": canonical_weights(LAYERS, extra=(0.0, -6.0)),
            # every data point goes through the full Taylor forward and reverse sweep: 24 M_w, like a collocation point
            "flops": lambda n: 24.0 * M_W * n["u"]}


def schrodinger_workload():
    """BASELINE configs[3] (1dcomplex-schrodinger/inf_cont_schrodinger.py:19-41): 2-100-100-100-100-2, N_0 = N_b = 50,
    N_f = 20000, Adam lr .05 / beta_1 .99 / eps .1, no L-BFGS; X0 = (x0, 0) as the evident intent of the driver"""
    sys.path.insert(0, os.path.join(PKG, "1dcomplex-schrodinger"))
    import schrodingerutil
    np.random.seed(1234)
    r = schrodingerutil.prep_data(os.path.join(PKG, "1dcomplex-schrodinger", "data", "NLS.mat"), 50, 50, 20000, noise=0.0)
    X_f, ub, lb, tb, x0, u0, v0, X0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17], r[18]
    m_w = macs(LAYERS_SCHRODINGER)
    return {"layers": LAYERS_SCHRODINGER, "pde": "schrodinger", "lb": lb, "ub": ub,
            "sets": {"X_f": X_f, "X_u": X0, "u": np.concatenate([u0, v0], 1),
                     "X_lb": np.concatenate((0 * tb + lb[0], tb), 1), "X_ub": np.concatenate((0 * tb + ub[0], tb), 1)},
            "pde_params": None, "adam": (0.05, 0.99, 0.999, 0.1), "lbfgs": None, "points": len(X_f),
            "w0": canonical_weights(LAYERS_SCHRODINGER),
            # collocation 24 M_w; initial-data points value channel only (6 M_w); the 2 x N_b boundary points carry
            # value and x-derivative (two of the four channels: 12 M_w)
            "flops": lambda n: m_w * (24.0 * n["f"] + 6.0 * n["u"] + 12.0 * 2 * n["b"])}


def runtime_record():
    try:
        import pinn_native
        return pinn_native.runtime_info()
    except Exception as e:                                     # never lose the line over a diagnostic
        return {"error": str(e)[:200]}


def self_launch(n, argv, child=None, device_count=None, timeout_s=None, out=None, err=None):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves -- one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT exactly as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py ...` would set them -- forward rank 0's stdout (the one JSON line),
    and return the exit code: 0, or the first failing rank's code with the tail of its stderr, or 124 on a time-out
    (PINN_BENCH_LAUNCH_TIMEOUT_S, default 3600).  Nothing is left running on any exit path.
    child / device_count are seams for tests/test_data_parallel_gloo.py (a scripted engine needs no GPU)."""
    import socket
    import tempfile
    out, err = out or sys.stdout, err or sys.stderr
    if n < 1:
        err.write("bench.py: --gpus must be >= 1 (got %d)\n" % n)
        return 2
    if "PINN_BENCH_DEVICE" not in os.environ:                 # (set = every rank on that one device: single-GPU tests)
        if device_count is None:
            import pinn_native
            device_count = pinn_native.device_count
        have = device_count()
        if have < n:
            err.write("bench.py: --gpus %d asked, but this node exposes %d GPU(s) to this process "
                      "(hipGetDeviceCount; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES respected)\n" % (n, have))
            return 2
    timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("PINN_BENCH_LAUNCH_TIMEOUT_S", "3600"))
    with socket.socket() as sk:                               # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child = child or [sys.executable, os.path.abspath(__file__)]
    procs, logs = [], []
    with tempfile.TemporaryDirectory(prefix="pinn_bench_") as tmp:
        try:
            for r in range(n):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
                fo = open(os.path.join(tmp, "rank%d.out" % r), "w+")
                fe = open(os.path.join(tmp, "rank%d.err" % r), "w+")
                logs.append((fo, fe))
                procs.append(subprocess.Popen(child + list(argv), env=env, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL))
            deadline = time.time() + timeout_s
            failed = None
            while failed is None and any(p.poll() is None for p in procs):
                for r, p in enumerate(procs):
                    if p.poll() not in (None, 0):
                        failed = r
                        break
                if time.time() > deadline:
                    failed = -1
                    break
                time.sleep(0.05)
            if failed is None:
                failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)

            def tail(f, k=25):
                f.flush()
                f.seek(0)
                return "".join(f.readlines()[-k:])
            if failed is not None:
                for p in procs:                               # exactly the processes started above, nothing by pattern
                    if p.poll() is None:
                        p.kill()
                for p in procs:
                    p.wait()
                if failed < 0:
                    err.write("bench.py: the %d-rank launch did not finish within %.0f s; ranks killed.  rank 0 stderr tail:\n%s"
                              % (n, timeout_s, tail(logs[0][1])))
                    return 124
                rc = procs[failed].returncode
                err.write("bench.py: rank %d of %d exited with code %d; the other ranks were stopped.  Its stderr tail:\n%s"
                          % (failed, n, rc, tail(logs[failed][1])))
                return rc if 0 < rc < 256 else 1
            fo = logs[0][0]
            fo.flush()
            fo.seek(0)
            for line in fo:                                   # the contract line to stdout; library chatter on rank 0's
                (out if line.startswith("{") else err).write(line)   # stdout ("[Gloo] Rank 0 is connected ...") to stderr
            out.flush()
            return 0
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
                    p.wait()
            for fo, fe in logs:
                fo.close()
                fe.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--dtype", default=os.environ.get("PINN_BENCH_DTYPE", "f64"), choices=["f32", "f64"],
                    help="arithmetic of the headline leg; f64 = the reference's (utils/neuralnetwork.py:24-26) and the "
                         "product's default; the other arithmetic is carried as float32_leg / float64_leg")
    ap.add_argument("--nf-total", type=int, default=10000,
                    help="collocation points in total, split over the ranks (strong scaling; the metric's N_f = 10000)")
    ap.add_argument("--nf-per-gpu", type=int, default=0,
                    help="weak scaling instead: this many collocation points per GPU (overrides --nf-total)")
    ap.add_argument("--kernel-path", type=int, default=-1, help="-1 engine default; see pinn_set_kernel_path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-final-error", action="store_true")
    ap.add_argument("--no-script-leg", action="store_true", help="skip the script-level leg (fit() of the drop-in script, logging on)")
    ap.add_argument("--no-f64-leg", "--no-other-leg", dest="no_f64_leg", action="store_true",
                    help="skip the leg in the other arithmetic (float32_leg under --dtype f64, float64_leg under f32)")
    ap.add_argument("--no-cfg5-leg", action="store_true", help="skip the N_f = 10^6 leg (BASELINE configs[4])")
    ap.add_argument("--no-cfg34-legs", action="store_true",
                    help="skip the identification (BASELINE configs[2]) and Schrodinger (configs[3]) legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = int(os.environ.get("PINN_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # override: tests on one GPU
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` typed as it stands: become the launcher (one rank per GPU, same environment as
            # the torch.distributed.run form; both forms are equivalent)
            sys.exit(self_launch(args.gpus, sys.argv[1:]))
        args.gpus = world
    dist = None
    if world > 1:
        # a collective that never returns (a communicator that cannot form, a lost rank) must not hang the whole launch: after
        # PINN_BENCH_RANK_TIMEOUT_S (default 1800) the rank says where it stands and exits 124 -- the launcher stops the rest
        import threading
        limit = float(os.environ.get("PINN_BENCH_RANK_TIMEOUT_S", "1800"))

        def give_up():
            sys.stderr.write("bench.py: rank %d of %d did not finish within %.0f s; giving up (exit code 124)\n" % (rank, world, limit))
            sys.stderr.flush()
            os._exit(124)
        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()
        import pinn_native
        pinn_native.load()                    # binds the HIP runtime (torch's set in a rank: pinn_native._bind_runtime) first
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    wd = World(dist, world, rank)

    import burgersutil
    mat = os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat")

    def dataset(n_f):
        np.random.seed(1234)
        r = burgersutil.prep_data(mat, 100, n_f, noise=0.0)
        return r[5], r[6], (r[9], r[7], r[8], r[11], r[10])          # X_star, u_star, (X_f, X_u, u, lb, ub)

    weak = args.nf_per_gpu > 0
    n_f_total = args.nf_per_gpu * world if weak else args.nf_total
    X_star, u_star, data = dataset(n_f_total)
    w0 = canonical_weights()
    k_adam = args.steps // 3
    k_lbfgs = args.steps - k_adam

    def init_comm(eng):
        if world == 1:
            return "none"
        from pinn_native.parallel import init_engine_comm
        return init_engine_comm(eng, dist, world, rank, probe=True)   # "rccl" unless PINN_COMM=auto and the mailboxes win

    # ---- headline leg ---------------------------------------------------------------------------------------
    main_leg, eng = leg("headline", args.dtype, device, data, w0, wd, k_adam, k_lbfgs, args.warmup, args.kernel_path,
                        init_comm=init_comm)
    with_traffic(main_leg, world)
    replicas_identical = None
    if dist is not None:
        import hashlib
        digests = [None] * world
        dist.all_gather_object(digests, hashlib.sha256(eng.get_weights().tobytes()).hexdigest())
        replicas_identical = all(d == digests[0] for d in digests)

    # ---- accuracy legs (untimed): the reference's default schedule on the reference's own N_f = 10000 set ------
    errs = {}
    ref_data = data if n_f_total == 10000 else None
    if not args.no_final_error:
        if ref_data is None:
            _, _, ref_data = dataset(10000)
            from pinn_native.parallel import attach_shards
            attach_shards(eng, world, rank, X_f=ref_data[0], X_u=ref_data[1], u=ref_data[2])
        errs[args.dtype] = final_error(eng, w0, X_star, u_star)
    eng.close()

    # ---- the same steps in the other arithmetic (float32 = the throughput mode north_star sanctions, when the
    # headline is the reference's float64; float64 when the headline was asked for in float32) -----------------------
    other = "f32" if args.dtype == "f64" else "f64"
    other_leg = None
    if not args.no_f64_leg:
        other_leg, e2 = leg("float32" if other == "f32" else "float64", other, device, data, w0, wd, k_adam, k_lbfgs,
                            min(args.warmup, 9), spin=False, init_comm=init_comm)
        with_traffic(other_leg, world)
        if not args.no_final_error:
            if n_f_total != 10000:
                from pinn_native.parallel import attach_shards
                attach_shards(e2, world, rank, X_f=ref_data[0], X_u=ref_data[1], u=ref_data[2])
            errs[other] = final_error(e2, w0, X_star, u_star)
        e2.close()

    # ---- cfg 5: N_f = 10^6 in total, sharded over the ranks (BASELINE configs[4]) ---------------------------------
    cfg5 = None
    if not args.no_cfg5_leg and not weak:
        _, _, data5 = dataset(1000000)
        cfg5, e5 = leg("cfg5", args.dtype, device, data5, w0, wd, k_adam, k_lbfgs, min(args.warmup, 6),
                       args.kernel_path, spin=False, init_comm=init_comm)
        with_traffic(cfg5, world)
        # parity at the size this leg times: the loss at the canonical weights beside the reference's own value for this
        # set (tests/golden/burgers_eval_1e6.npz, generated from the reference by tests/golden/make_golden.py; the full
        # loss+gradient comparison is tests/test_gpu_cfg5_parity.py)
        try:
            e5.set_weights(w0)
            l5 = float(e5.loss_grad(want_grad=False)[0])
            gold = float(np.load(os.path.join(ROOT, "tests", "golden", "burgers_eval_1e6.npz"))["loss_w0"])
            cfg5["loss_at_w0"], cfg5["loss_at_w0_reference"] = l5, gold
            cfg5["loss_at_w0_rel_dev"] = abs(l5 - gold) / gold
        except Exception as e:                                    # never lose the line over a diagnostic
            cfg5["loss_at_w0_error"] = str(e)[:200]
        cfg5["note"] = ("BASELINE configs[4]: the throughput regime, and the leg multi-GPU scaling is to be judged on "
                        "(the N_f = 10000 headline is one tile per workgroup: its step is a latency chain that "
                        "sharding cannot shorten).  scaling_vs_n1_estimate inputs: ms_per_step here at N ranks vs the "
                        "N = 1 line, allreduce_probe_us = one [P+4] float64 exchange on this node")
        e5.close()

    # ---- cfg 3 / cfg 4: the other two single-GPU configurations of BASELINE.json, in the headline's arithmetic -------------
    cfg3 = cfg4 = None
    if not args.no_cfg34_legs and not weak:
        for tag, make in (("cfg3", identification_workload), ("cfg4", schrodinger_workload)):
            wl = make()
            ka = max(1, int(round(args.steps * wl.get("adam_share", 1.0 / 3.0)))) if wl["lbfgs"] else args.steps
            lg, e = leg(tag, args.dtype, device, wl, wl["w0"], wd, ka, args.steps - ka, min(args.warmup, 6), spin=False,
                        init_comm=init_comm, min_ms=MIN_TIMED_MS / 3.0)
            with_traffic(lg, world)
            lg["schedule"] = ("%d Adam + %d L-BFGS iterations per block" % (lg["adam_steps_per_block"], lg["lbfgs_steps_per_block"]))
            e.close()
            if tag == "cfg3":
                cfg3 = lg
                lg["note"] = ("BASELINE configs[2]: identification, N_u = 10000 data points carrying misfit and residual, "
                              "lambda_1 / lambda_2 trainable (P = 3023); value = data points per second")
            else:
                cfg4 = lg
                lg["note"] = ("BASELINE configs[3]: Schrodinger 2-100x4-2 (P = 30802), N_f = 20000, N_0 = N_b = 50, Adam only; "
                              "roofline on the forward + reverse sweep pair (729600 FLOP per collocation point)")

    # every rank empties its C stdio buffer (RCCL's banner) before rank 0 prints: the JSON line stays the last line
    def flush_c():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    flush_c()
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        ens = reference_ensemble("f64")
        ens_of = {"f64": ens, "f32": reference_ensemble("f32")}      # each arithmetic against the ensemble the tests use for it

        def delta(v):
            return abs(v - ens["reference"]) if (v is not None and ens) else None
        out = {
            "metric": "collocation-points/sec",
            "value": main_leg["value"], "unit": "collocation-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_leg["ms_per_step"],
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": args.dtype,
            "data": "reference fixture 1d-burgers/data/burgers_shock.mat (N_u=100 boundary/initial samples, the "
                    "25600-point error grid) + Latin-hypercube collocation points, numpy seed 1234 (the reference's "
                    "prep_data stream); canonical glorot initial weights, seed 1234",
            "config": {"workload": "1D Burgers continuous inference (BASELINE configs[1]): 8x20 tanh MLP, N_u=100, "
                                   "N_f=%d in total over %d GPU(s) (LHS, seed 1234), %d Adam + %d L-BFGS iterations "
                                   "per block, canonical glorot init" % (n_f_total, world, k_adam, k_lbfgs),
                       "n_f_total": n_f_total, "n_f_per_gpu": main_leg["n_f_per_gpu"], "n_u": 100,
                       "parallelism": "dp%d" % world, "allreduce": main_leg["allreduce"],
                       "allreduce_probe_us": main_leg["allreduce_probe_us"], "replicas_identical": replicas_identical,
                       "kernel_path": main_leg["kernel_path"], "lbfgs_done_code": main_leg["lbfgs_done_code"],
                       "timing": "median of %d blocks of exactly %d steps (barrier+sync around each, MAX over ranks), "
                                 "%.1f ms timed in total" % (main_leg["blocks_timed"], args.steps, main_leg["timed_ms_total"]),
                       "ms_per_step_first_block": main_leg["ms_per_step_first_block"],
                       "ms_per_step_min_block": main_leg["ms_per_step_min_block"]},
            "valid": main_leg["valid"],
            "roofline": main_leg["roofline"],
            # which HIP runtime / RCCL build this process bound (pinn_native._bind_runtime: /opt/rocm in a plain process,
            # torch's bundled set in a rank that needs torch.distributed) and how the ranks were started
            "runtime": runtime_record(),
            # a scaling series mixes runtimes under the default policy: a rank of N > 1 binds the ROCm 7.0 set
            # bundled with torch (it needs torch.distributed), a plain N = 1 process binds /opt/rocm (7.2).  Measured
            # difference at N = 1: 0.14 % (profiles/r05_bench_a_*_runtime.json); PINN_HIP_RUNTIME=torch puts N = 1 on the
            # ranks' runtime, and this key says which policy this line ran under
            "runtime_policy": {"PINN_HIP_RUNTIME": os.environ.get("PINN_HIP_RUNTIME", "auto"),
                               "same_runtime_as_multi_rank_runs": (runtime_record().get("bound") == "torch")},
            "launch": ("single process" if world == 1 else "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                       else "bench.py --gpus %d (self-launched ranks)" % world),
            ("float32_leg" if other == "f32" else "float64_leg"): other_leg,
            "cfg5_leg": cfg5, "cfg3_leg": cfg3, "cfg4_leg": cfg4,
            "final_l2_error": errs.get(args.dtype), "final_l2_error_f64": errs.get("f64"),
            "final_l2_error_f32": errs.get("f32"),
            "final_l2_error_reference": ens, "final_l2_error_reference_f32": ens_of["f32"],
            # north_star's literal criterion, per arithmetic: |final error - the reference's k = 0 run| (<= 1e-3 asked;
            # the reference's own runs differ by more than that between two hosts)
            "final_l2_error_abs_delta": delta(errs.get(args.dtype)), "final_l2_error_abs_delta_f64": delta(errs.get("f64")),
            "final_l2_error_abs_delta_f32": delta(errs.get("f32")),
            "final_l2_error_within_1e-3": {k: (delta(v) is not None and delta(v) <= 1e-3) for k, v in errs.items()},
            "final_l2_error_inside_reference_ensemble": {k: (ens_of[k] is not None and
                                                             ens_of[k]["ensemble_min"] <= v <= ens_of[k]["ensemble_max"])
                                                         for k, v in errs.items()},
            "final_l2_error_schedule": "100 Adam (lr .03) + 200 L-BFGS (lr .8, m=50), reference defaults, on the "
                                       "reference set N_f=10000 (sharded over the ranks)",
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_port(ref_data[0] if ref_data else data[0][:10000], data[1], data[2],
                                                    data[3], data[4], w0, X_star, u_star)
        else:
            out["cpu_baseline"] = None
        if not args.no_script_leg and world == 1:
            try:
                sl = script_leg(args.dtype, device)
                sl["ratio_to_engine_level_step"] = sl["ms_per_step"] / main_leg["ms_per_step"]
                if out["cpu_baseline"] and out["cpu_baseline"].get("value"):
                    # like for like: fit() with its logging on both sides
                    sl["gpu_over_cpu_baseline"] = sl["value"] / out["cpu_baseline"]["value"]
                out["script_leg"] = sl
            except Exception as e:                                  # never lose the line over a secondary leg
                out["script_leg"] = {"error": str(e)[-300:]}
        flush_c()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
