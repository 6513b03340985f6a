"""CPU, world_size 2, gloo: the data-parallel contract of the engine's multi-GPU path.

Each rank takes its contiguous shard of every point set (pinn_native.parallel.shard_bounds,
the same helper bench.py and the engine launcher use), evaluates loss/gradient partial sums
normalised by the GLOBAL set sizes, and an all-reduce(SUM) of [grad | loss] must reproduce the
single-process result.  The per-shard evaluator here is the CPU oracle standing in for the
HIP engine (no GPU in this test); on the GPU box the same check runs with the real engine and
RCCL at world_size 1 (tests/test_gpu_comm.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG, BURGERS_MAT


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(PKG, "1d-burgers"), os.path.join(PKG, "utils")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import burgersutil
    from oracle import init, pde
    from pinn_native.parallel import shard_bounds
    np.random.seed(1234)
    r = burgersutil.prep_data(BURGERS_MAT, 63, 1001, noise=0.0)     # ragged on purpose
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    layers = [2] + [20] * 8 + [1]
    w = init.glorot_flat(layers)
    nu = 0.01 / np.pi
    f0, f1 = shard_bounds(len(X_f), world, rank)
    u0, u1 = shard_bounds(len(X_u), world, rank)
    # collocation partial (global 1/N_f) + data partial (global 1/N_u) for this rank's shards
    lo_f, g_f, _ = pde.burgers_loss_grad(w, layers, lb, ub, X_f[f0:f1], X_u, u, nu,
                                         n_f_total=len(X_f), with_data=False)
    full_u = pde.burgers_loss_grad(w, layers, lb, ub, X_f[:0], X_u[u0:u1], u[u0:u1], nu,
                                   n_f_total=1)
    scale = (u1 - u0) / len(X_u)            # mean over the shard -> contribution to the global mean
    buf = torch.from_numpy(np.concatenate([g_f + scale * full_u[1], [lo_f + scale * full_u[0]]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process(tmp_path, burgers_sets):
    from oracle import init, pde
    import burgersutil
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    red = np.load(tmp_path / "reduced.npy")
    np.random.seed(1234)
    r = burgersutil.prep_data(BURGERS_MAT, 63, 1001, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    layers = [2] + [20] * 8 + [1]
    lo, g, _ = pde.burgers_loss_grad(init.glorot_flat(layers), layers, lb, ub, X_f, X_u, u,
                                     0.01 / np.pi)
    assert abs(red[-1] - lo) < 1e-14
    assert np.max(np.abs(red[:-1] - g)) / np.max(np.abs(g)) < 1e-13


def test_shard_bounds_cover_and_balance():
    from pinn_native.parallel import shard_bounds
    for n in (0, 1, 7, 100, 10000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- communicator set-up protocol (parallel.init_engine_comm) with a scripted engine, world_size 2 over gloo ----------
class _ScriptedEngine(object):
    """Stands in for pinn_native.Engine: records the calls, fails where the scenario says so."""

    def __init__(self, rank, scenario):
        self.rank, self.s, self.calls, self.mode = rank, scenario, [], "none"

    def _fail(self, what):
        import pinn_native
        if self.s.get(what) == self.rank:
            raise pinn_native.PinnNativeError("scripted failure of %s on rank %d" % (what, self.rank))

    def comm_init(self, uid, world, rank):
        self.calls.append("comm_init")
        import time
        if self.s.get("late") == self.rank:                    # a rank that reaches the collective seconds after the others
            time.sleep(self.s.get("late_s", 2.0))
        if self.s.get("rccl_hangs_unless_failing") is not None and self.s.get("rccl_fails") != self.rank:
            time.sleep(3600)                                   # its peer died before the call: ncclCommInitRank never returns
        if self.s.get("rccl_fails") in ("all", self.rank):
            import pinn_native
            raise pinn_native.PinnNativeError("ncclCommInitRank failed: scripted (rank %d)" % self.rank)
        self.mode = "rccl"

    def comm_xgmi_export(self, world, rank):
        self.calls.append("export"); self._fail("export"); return b"h%d" % rank + bytes(62)

    def comm_xgmi_attach(self, handles):
        self.calls.append("attach"); self._fail("attach"); return self.s.get("unmapped") != self.rank

    def comm_xgmi_selftest(self):
        self.calls.append("selftest"); return self.s.get("bad_selftest") != self.rank

    def comm_benchmark(self, mode, iters=200):
        self.calls.append("bench_" + mode)
        return {"rccl": 20.0, "mailbox": self.s.get("mailbox_us", 5.0)}[mode]

    def comm_set_mode(self, mode):
        self.calls.append("set_" + mode); self.mode = mode


def _comm_worker(rank, world, port, out_dir, scenario):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if scenario.get("init_timeout_s"):
        os.environ["PINN_COMM_INIT_TIMEOUT_S"] = str(scenario["init_timeout_s"])
    if scenario.get("policy"):
        os.environ["PINN_COMM"] = scenario["policy"]
    else:
        os.environ.pop("PINN_COMM", None)                  # default policy = RCCL (north_star), mailboxes opt-in
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pinn_native
    from pinn_native import parallel
    pinn_native.Engine.comm_unique_id = staticmethod(lambda: b"u" * 128)
    eng = _ScriptedEngine(rank, scenario)
    try:
        mode = parallel.init_engine_comm(eng, dist, world, rank)
    except RuntimeError as e:
        mode = "error:" + str(e)[:120]
        eng.mode = mode
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%s|%s|%s|%s" % (mode, eng.mode, ",".join(eng.calls), getattr(eng, "comm_fallback", None)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,expect", [
    ({}, "rccl"),                                      # default policy: RCCL, the mailboxes are never touched
    ({"policy": "auto"}, "mailbox"),                   # opt-in; everything works, mailboxes faster -> mailboxes everywhere
    ({"policy": "auto", "mailbox_us": 50.0}, "rccl"),  # ... but slower on this node -> RCCL everywhere
    ({"policy": "auto", "export": 1}, "rccl"),         # one rank cannot export -> nobody attaches
    ({"policy": "auto", "unmapped": 0}, "rccl"),       # one rank cannot map a peer -> nobody runs the self-test
    ({"policy": "auto", "bad_selftest": 1}, "rccl"),   # one rank's self-test fails -> RCCL everywhere
    ({"rccl_fails": "all"}, "mailbox"),                # round 5: ncclCommInitRank fails on EVERY rank -> the self-tested mailboxes take over
])
def test_comm_setup_is_unanimous(tmp_path, scenario, expect):
    port = 29600 + (os.getpid() + len(str(scenario))) % 300
    mp.spawn(_comm_worker, args=(2, port, str(tmp_path), scenario), nprocs=2, join=True)
    outs = [open(tmp_path / ("rank%d.txt" % r)).read().split("|") for r in range(2)]
    assert outs[0][0] == outs[1][0] == expect and outs[0][1] == outs[1][1] == expect, outs
    if "rccl_fails" in scenario:
        assert all(o[3].startswith("rccl failed: ncclCommInitRank failed") and "selftest" in o[2] for o in outs), outs
    elif "policy" not in scenario:
        assert all("export" not in o[2] and "attach" not in o[2] for o in outs)
    if "unmapped" in scenario:
        assert all("selftest" not in o[2] for o in outs)
    if "export" in scenario:
        assert all("attach" not in o[2] for o in outs)


# ---- bench.py's timing contract (world_size 2 over gloo, scripted engine) ------------------------------------------
class _SleepyEngine(object):
    """stands in for pinn_native.Engine in bench.time_blocks: a 'step' costs a rank-dependent sleep"""

    def __init__(self, rank):
        self.dt = 0.004 if rank == 0 else 0.010
        self.steps = 0

    def set_weights(self, w): pass
    def adam_init(self, *a): pass
    def sync(self): pass

    def adam_run(self, n, want_losses=True):
        import time
        time.sleep(self.dt * n)
        self.steps += n

    def lbfgs_begin(self, n, *a): self.left = n

    def lbfgs_run(self, n):
        import time
        time.sleep(self.dt * self.left)
        self.steps += self.left
        return [], [], 1


def _bench_worker(rank, world, port, out_dir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    eng = _SleepyEngine(rank)
    times, done = bench.time_blocks(eng, bench.World(dist, world, rank), None, 1, 2, min_ms=100.0)
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%d|%d|%s" % (done, eng.steps, ",".join("%.6f" % t for t in times)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_blocks_are_max_over_ranks_and_identical_everywhere(tmp_path):
    """every block is exactly K steps, its time is the MAX over ranks (the slow rank sleeps 10 ms per step), every rank
    sees the same list and therefore runs the same number of blocks, and blocks repeat until >= min_ms are timed"""
    port = 29700 + os.getpid() % 200
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [open(tmp_path / ("rank%d.txt" % r)).read().split("|") for r in range(2)]
    assert outs[0][2] == outs[1][2] and outs[0][0] == outs[1][0] == "1"
    times = [float(t) for t in outs[0][2].split(",")]
    assert all(t >= 3 * 0.010 for t in times)                    # 3 steps of the slow rank
    assert sum(times) >= 0.100 and sum(times[:-1]) < 0.100       # stops once >= 100 ms are timed
    assert int(outs[0][1]) == int(outs[1][1]) == 3 * len(times)  # exactly K steps per block on every rank


# ---- bench.py end to end at world size 2 (gloo) with a scripted engine: control flow, legs, JSON contract -----------------
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
from bench_stub_child import StubEngine as _StubEngine            # noqa: E402  (shared with the self-launch child)


def _bench_main_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    os.environ.pop("PINN_COMM", None)
    os.environ["PINN_BENCH_MIN_TIMED_MS"] = "20"          # the product default (3 s per leg) is for real GPUs
    import contextlib
    import io
    import pinn_native
    import bench
    pinn_native.Engine = _StubEngine
    pinn_native.device_info = lambda d=0: {"name": "stub", "compute_units": 256, "hbm_bytes": 0}
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "6", "--warmup", "3"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(os.path.join(out_dir, "rank%d.out" % rank), "w") as f:
        f.write(buf.getvalue())
    with open(os.path.join(out_dir, "rank%d.sets" % rank), "w") as f:
        f.write(";".join("%s:%d:%d:%d:%s" % (e.dtype, e.n_f, e.n_u, e.n_b, e.comm[1:] if e.comm else None) for e in _StubEngine.made))


def test_bench_main_world2_emits_one_contract_line(tmp_path):
    """python -m torch.distributed.run ... bench.py --gpus 2, with the engine scripted: rank 0 prints exactly one JSON
    line carrying the contract keys; the metric's N_f = 10000 and cfg 5's N_f = 10^6 are SPLIT over the ranks (strong
    scaling), every leg's engine joined a 2-rank communicator, nothing is printed by rank 1"""
    import json
    port = 29400 + os.getpid() % 100
    mp.spawn(_bench_main_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out0 = open(tmp_path / "rank0.out").read().strip().splitlines()
    assert open(tmp_path / "rank1.out").read().strip() == ""
    assert len(out0) == 1
    j = json.loads(out0[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 3 and j["scaling"] == "strong"
    assert j["vs_baseline"] is None and j["cpu_baseline"] is None and j["higher_is_better"] is True
    assert j["config"]["n_f_total"] == 10000 and j["config"]["n_f_per_gpu"] == 5000 and j["config"]["parallelism"] == "dp2"
    assert j["config"]["allreduce"] == "rccl" and j["config"]["replicas_identical"] is True
    assert j["config"]["allreduce_probe_us"] == {"rccl": 21.0}       # probed at every N > 1, RCCL default included
    assert j["dtype"] == "f64" and j["roofline"]["peak"] == 78.6     # headline = the reference's arithmetic
    assert "burgers_shock.mat" in j["data"] and j["final_l2_error_abs_delta"] is not None
    assert abs(j["value"] - 10000 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"]
    assert j["cfg5_leg"]["n_f_total"] == 1000000 and j["cfg5_leg"]["n_f_per_gpu"] == 500000
    assert j["float32_leg"]["dtype"] == "f32" and j["float32_leg"]["kernel_path"] == 2
    # BASELINE configs[2] and [3] as driver-timed legs, in the headline's arithmetic, each with its own roofline
    c3, c4 = j["cfg3_leg"], j["cfg4_leg"]
    assert c3["n_f_total"] == 10000 and c3["dtype"] == "f64" and c3["lbfgs_steps_per_block"] == 5 and c3["adam_steps_per_block"] == 1
    assert abs(c3["roofline"]["algorithmic_flop_per_launch"] - 24 * 2860 * 5000) < 1
    assert c4["n_f_total"] == 20000 and c4["adam_steps_per_block"] == 6 and c4["lbfgs_steps_per_block"] == 0 and c4["kernel_path"] == 4
    assert abs(c4["roofline"]["algorithmic_flop_per_launch"] - 30400 * (24 * 10000 + 6 * 25 + 12 * 2 * 25)) < 1
    assert c4["roofline"]["kernel"] == "pinn::k_t16_fwd+k_t16_bwd" and c4["roofline"]["traffic"] is None
    assert "one GPU" in c4["roofline"]["traffic_source"]            # a null traffic says why
    for r in range(2):
        sets = open(tmp_path / ("rank%d.sets" % r)).read().split(";")
        assert sets == ["f64:5000:50:0:(2, %d)" % r, "f32:5000:50:0:(2, %d)" % r, "f64:500000:50:0:(2, %d)" % r,
                        "f64:0:5000:0:(2, %d)" % r, "f64:10000:25:25:(2, %d)" % r], sets


# ---- `python bench.py --gpus N` with no launcher around it: bench.self_launch starts the ranks itself ------------------
_STUB_CHILD = [sys.executable, os.path.join(ROOT, "tests", "helpers", "bench_stub_child.py")]


def _self_launch(n, argv, env=None, **kw):
    import io
    import bench
    out, err = io.StringIO(), io.StringIO()
    old = dict(os.environ)
    os.environ.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PINN_BENCH_DEVICE"):
        os.environ.pop(k, None)
    try:
        rc = bench.self_launch(n, argv, child=_STUB_CHILD, out=out, err=err, **kw)
    finally:
        os.environ.clear()
        os.environ.update(old)
    return rc, out.getvalue(), err.getvalue()


def test_bench_self_launch_two_ranks_prints_one_contract_line():
    """the driver's N-GPU command may be plain `python3 bench.py --gpus N`; the launcher spawns the
    ranks with the environment torch.distributed.run would give them and forwards rank 0's single JSON line"""
    import json
    rc, out, err = _self_launch(2, ["--gpus", "2", "--steps", "6", "--warmup", "3", "--no-cfg34-legs"],
                                device_count=lambda: 2)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and "exited with code" not in err
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 3 and j["scaling"] == "strong"
    assert j["config"]["parallelism"] == "dp2" and j["config"]["n_f_per_gpu"] == 5000 and j["config"]["replicas_identical"] is True
    assert j["launch"] == "bench.py --gpus 2 (self-launched ranks)"
    # a rank of a data-parallel launch binds ONE runtime: torch's bundled set (it needs torch.distributed)
    assert j["runtime"]["bound"] == "torch" and j["runtime"]["single_runtime"] is True
    assert j["runtime"]["hip_runtime"] > 0 and j["runtime"]["rccl_version"].count(".") == 2
    assert j["roofline"]["traffic_provenance"]["measured_in_run"] is False


def test_bench_self_launch_refuses_more_ranks_than_devices():
    rc, out, err = _self_launch(9, ["--gpus", "9"], device_count=lambda: 8)
    assert rc == 2 and out == ""
    assert err.count("\n") == 1 and "--gpus 9" in err and "8 GPU(s)" in err


def test_bench_self_launch_reports_the_failing_rank_and_stops_the_others():
    import time
    t0 = time.time()
    rc, out, err = _self_launch(2, ["--gpus", "2", "--steps", "6", "--warmup", "3"], env={"BENCH_STUB_FAIL_RANK": "1"},
                                device_count=lambda: 2)
    assert rc == 7 and out == ""
    assert "rank 1 of 2 exited with code 7" in err and "simulated failure" in err
    assert time.time() - t0 < 120                     # rank 0 was waiting in the rendezvous: stopped, not waited for


def test_bench_self_launch_times_out_instead_of_hanging():
    rc, out, err = _self_launch(2, ["--gpus", "2"], env={"BENCH_STUB_HANG": "1"}, device_count=lambda: 2, timeout_s=3)
    assert rc == 124 and out == "" and "did not finish within 3 s" in err


def test_bench_rank_watchdog_ends_a_rank_that_never_returns():
    """a collective that never completes must not hang the launch: the stalled rank gives up after
    PINN_BENCH_RANK_TIMEOUT_S with exit code 124 and the launcher stops the other one"""
    rc, out, err = _self_launch(2, ["--gpus", "2", "--steps", "6", "--warmup", "3", "--no-cfg34-legs"],
                                env={"BENCH_STUB_STALL_RANK": "1", "PINN_BENCH_RANK_TIMEOUT_S": "8"}, device_count=lambda: 2)
    assert rc == 124 and out == ""
    # (rank 0 waits for rank 1 in a barrier: both watchdogs expire, whichever rank exits first is the one reported)
    assert "of 2 exited with code 124" in err and "did not finish within 8 s" in err


def test_bench_main_becomes_the_launcher_when_no_rank_environment_is_set(monkeypatch):
    import bench
    seen = {}
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: seen.update(n=n, argv=list(argv)) or 0)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen == {"n": 4, "argv": ["--gpus", "4", "--steps", "5", "--warmup", "2"]}


# ---- the drop-in classes launched data-parallel (world_size 2 over gloo, scripted engine) ---------------------------------
class _SurfaceEngine(object):
    """what utils/neuralnetwork.py and the scripts' classes touch of pinn_native.Engine; records the point sets it is
    given; 'training' moves the weights by a function of the (all-reduced, hence rank-independent) step count only --
    unless the scenario makes one rank drift"""
    made = []
    drift_rank = None

    def __init__(self, layers, lb, ub, pde="burgers", dtype="f64", device=0):
        self.layers, self.pde, self.dtype, self.device = layers, pde, dtype, device
        self.n_params = sum(a * b + b for a, b in zip(layers[:-1], layers[1:])) + (2 if pde == "burgers_ide" else 0)
        self.w = np.zeros(self.n_params)
        self.n_f = self.n_u = self.n_b = 0
        self.sets, self.comm, self.steps = {}, None, 0
        _SurfaceEngine.made.append(self)

    def set_collocation(self, X, n_total=None): self.n_f = len(X); self.sets["f"] = (np.array(X), n_total)
    def set_data(self, X, u, n_total=None): self.n_u = len(X); self.sets["u"] = (np.array(X), np.array(u), n_total)
    def set_boundary(self, A, B, n_total=None): self.n_b = len(A); self.sets["b"] = (np.array(A), np.array(B), n_total)
    def set_pde_params(self, *p): pass
    def set_weights(self, w): self.w = np.array(w, dtype=np.float64)
    def get_weights(self): return self.w.copy()
    def adam_init(self, *a): pass

    def lhs_collocation(self, n_design, seed, first=0, count=None):        # "design point i" = (i, seed)
        self.n_f = count
        self.sets["f"] = (np.stack([np.arange(first, first + count, dtype=np.float64), np.full(count, float(seed))], 1), n_design)

    def get_collocation(self): return self.sets["f"][0].copy()

    def _walk(self, n):
        self.steps += n
        self.w = self.w + 1e-3 * n + (1e-9 if _SurfaceEngine.drift_rank == int(os.environ["RANK"]) else 0.0)

    def adam_run(self, n, want_losses=True): self._walk(n); return 1.0 / (self.steps - np.arange(n)[::-1] + 1.0)
    def adam_run_terms(self, n): self._walk(n); return np.full((n, 3), 0.1)
    def lbfgs_begin(self, n, *a): self.left, self.it = n, 0

    def lbfgs_run(self, n):
        k = min(n, self.left)
        self.left -= k
        its = np.arange(self.it + 1, self.it + k + 1 - (1 if self.left == 0 else 0), dtype=np.int32)
        self.it += k
        self._walk(k)
        return its, 0.5 / (its + 1.0), int(self.left == 0)

    def loss_grad(self, want_grad=True): return 0.3, (np.zeros(self.n_params) if want_grad else None), np.array([0.1, 0.1, 0.1])
    def predict(self, X): return np.zeros((len(X), self.layers[-1]))
    def residual(self): return np.zeros((self.n_f, self.layers[-1]))
    def residual_at(self, X): return np.zeros((len(X), self.layers[-1]))
    def error_l2(self, X, ref, modulus=False): return 0.25
    def status(self): return self.steps, 0
    def comm_init(self, uid, world, rank): self.comm = (world, rank)
    def comm_mode(self): return "rccl" if self.comm else "none"
    def comm_benchmark(self, mode, iters=200): raise AssertionError("the drop-in surface must not run the timing probe")
    def close(self): pass

    @staticmethod
    def comm_unique_id(): return b"u" * 128


def _surface_worker(rank, world, port, out_dir, which, drift):
    for p in (ROOT, PKG, os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers"), os.path.join(PKG, "1dcomplex-schrodinger")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), PINN_NO_PLOT="1")
    os.environ.pop("PINN_COMM", None)
    os.environ.pop("PINN_DEVICE", None)
    import contextlib
    import importlib
    import io
    import pinn_native
    pinn_native.Engine = _SurfaceEngine
    pinn_native.device_info = lambda d=0: {"name": "stub", "compute_units": 256, "hbm_bytes": 0}
    pinn_native.device_count = lambda: 2
    _SurfaceEngine.drift_rank = drift
    sys.argv = ["script"]
    buf, err = io.StringIO(), ""
    with contextlib.redirect_stdout(buf):
        import neuralnetwork
        neuralnetwork.Engine = _SurfaceEngine
        try:
            if which == "burgers":
                mod = importlib.import_module("inf_cont_burgers")
                extra = {"resample_every": 5} if os.environ.get("SURFACE_RESAMPLE") else {}
                pinn = mod.run(dict(mod.hp, N_u=63, N_f=1001, tf_epochs=12, nt_epochs=7, log_frequency=5, **extra))
                if extra:
                    # after a device-side redraw f_model() still returns the residual at the FULL design on every
                    # rank (the blocks are gathered in rank order), not this rank's shard
                    seen = []
                    pinn._engine.residual_at = lambda X: (seen.append(np.array(X)), np.zeros((len(X), 1)))[1]
                    assert pinn.f_model().shape == (1001, 1)
                    assert np.array_equal(seen[-1][:, 0], np.arange(1001.0)) and np.all(seen[-1][:, 1] == 1234 + 10)
                X_f_full = None
            else:
                import runpy
                g = runpy.run_path(os.path.join(PKG, "1dcomplex-schrodinger", "inf_cont_schrodinger.py"), run_name="schro")
                hp = dict(g["hp"], N_0=51, N_b=37, N_f=2003, tf_epochs=6, log_frequency=3)
                import schrodingerutil
                from logger import Logger
                np.random.seed(1234)
                r = schrodingerutil.prep_data(os.path.join(PKG, "1dcomplex-schrodinger", "data", "NLS.mat"), 51, 37, 2003, noise=0.0)
                X_f, ub, lb, tb, x0, u0, v0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17]
                pinn = g["SchrodingerInformedNN"](hp, Logger(hp), X_f, tb, ub, lb)
                pinn.logger.set_error_fn(lambda: 0.5)
                pinn.fit(x0, np.concatenate([u0, v0], 1))
                assert pinn.f_model()[0].shape == (2003, 1)          # full arrays on every rank
        except RuntimeError as e:
            err = str(e)
    eng = _SurfaceEngine.made[-1]
    info = {"stdout": buf.getvalue(), "err": err, "device": eng.device, "comm": eng.comm, "n_f": eng.n_f, "n_u": eng.n_u,
            "n_b": eng.n_b, "n_total": {k: v[-1] for k, v in eng.sets.items()},
            "first_f": eng.sets["f"][0][0].tolist(), "first_u": eng.sets["u"][0][0].tolist(), "steps": eng.steps}
    import json
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(info, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["burgers", "schrodinger"])
def test_drop_in_scripts_shard_their_point_sets_over_the_ranks(tmp_path, which):
    """`python -m torch.distributed.run --nproc-per-node 2 1d-burgers/inf_cont_burgers.py` (and the Schrodinger class):
    device = LOCAL_RANK, each rank gets its contiguous block of the collocation / data / boundary sets with the GLOBAL
    counts as denominators, the engine joins a 2-rank communicator without the bench's timing probe, rank 0 alone
    prints the log, and fit() ends by checking that the replicas hold identical weights"""
    import json
    from pinn_native.parallel import shard_bounds
    port = 29800 + (os.getpid() + len(which)) % 90
    mp.spawn(_surface_worker, args=(2, port, str(tmp_path), which, None), nprocs=2, join=True)
    out = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    n_f, n_u, n_b = (1001, 63, 0) if which == "burgers" else (2003, 51, 37)
    for r, o in enumerate(out):
        assert o["err"] == "" and o["device"] == r and o["comm"] == [2, r]
        lo, hi = shard_bounds(n_f, 2, r)
        assert o["n_f"] == hi - lo and o["n_total"]["f"] == n_f
        lo, hi = shard_bounds(n_u, 2, r)
        assert o["n_u"] == hi - lo and o["n_total"]["u"] == n_u
        if n_b:
            lo, hi = shard_bounds(n_b, 2, r)
            assert o["n_b"] == hi - lo and o["n_total"]["b"] == n_b
    assert out[0]["first_f"] != out[1]["first_f"] and out[0]["first_u"] != out[1]["first_u"]      # different blocks
    assert out[0]["steps"] == out[1]["steps"] > 0
    assert out[1]["stdout"] == ""
    assert "Training started" in out[0]["stdout"] and "Training finished" in out[0]["stdout"]
    assert ("nt_epoch =      5" in out[0]["stdout"]) == (which == "burgers")
    if which == "schrodinger":
        assert out[0]["stdout"].count("mse_0") == 6


def test_residual_after_a_device_side_redraw_is_the_full_design_on_every_rank(tmp_path, monkeypatch):
    import json
    monkeypatch.setenv("SURFACE_RESAMPLE", "1")
    port = 29700 + os.getpid() % 90
    mp.spawn(_surface_worker, args=(2, port, str(tmp_path), "burgers", None), nprocs=2, join=True)
    out = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert all(o["err"] == "" for o in out)
    assert [o["n_f"] for o in out] == [501, 500] and all(o["n_total"]["f"] == 1001 for o in out)


def test_rccl_failure_on_some_ranks_only_is_an_error_everywhere(tmp_path):
    port = 29650 + os.getpid() % 40
    mp.spawn(_comm_worker, args=(2, port, str(tmp_path), {"rccl_fails": 1}), nprocs=2, join=True)
    outs = [open(tmp_path / ("rank%d.txt" % r)).read().split("|") for r in range(2)]
    assert all(o[0].startswith("error:RCCL communicator: ranks disagree") for o in outs), outs


# ---- the driver's world size: 8 ranks (gloo, scripted engines) -- first-contact insurance for `bench.py --gpus 8` ----------------
@pytest.mark.parametrize("scenario,expect", [
    ({}, "rccl"),
    ({"late": 5, "late_s": 3.0}, "rccl"),                                  # one rank reaches ncclCommInitRank 3 s after the rest
    ({"rccl_fails": "all"}, "mailbox"),                                    # RCCL fails everywhere -> self-tested mailboxes everywhere
    ({"rccl_fails": "all", "bad_selftest": 6}, "error:no gradient exchange available: rccl failed"),   # ... and they fail too
    ({"rccl_fails": 3}, "error:RCCL communicator: ranks disagree"),        # one failing rank: an error on all eight
    # one rank fails BEFORE the collective, the other seven would sit in it for ever: the deadline turns that into a
    # failure on every rank (unanimous), and the mailboxes take over
    ({"rccl_fails": 3, "rccl_hangs_unless_failing": True, "init_timeout_s": 3}, "mailbox"),
])
def test_comm_setup_is_unanimous_at_world_size_8(tmp_path, scenario, expect):
    port = 29300 + (os.getpid() + 7 * len(str(scenario))) % 90
    mp.spawn(_comm_worker, args=(8, port, str(tmp_path), scenario), nprocs=8, join=True)
    outs = [open(tmp_path / ("rank%d.txt" % r)).read().split("|") for r in range(8)]
    assert all(o[0].startswith(expect) for o in outs), outs
    assert len(set(o[1] for o in outs)) == 1 or expect.startswith("error"), outs      # one mode on all eight
    if scenario.get("init_timeout_s"):
        assert sum("did not return within 3 s" in o[3] for o in outs) >= 1 or "did not return" in outs[0][3], outs


def test_bench_self_launch_eight_ranks_prints_one_contract_line():
    """the driver's 8-GPU command, with the engine scripted: one JSON line, the metric's 10000 points and cfg 5's 10^6 split
    eight ways, every leg on an 8-rank communicator"""
    import json
    rc, out, err = _self_launch(8, ["--gpus", "8", "--steps", "6", "--warmup", "3"], device_count=lambda: 8)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and "exited with code" not in err
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["parallelism"] == "dp8"
    assert j["config"]["n_f_per_gpu"] == 1250 and j["config"]["replicas_identical"] is True
    assert j["cfg5_leg"]["n_f_per_gpu"] == 125000 and j["cfg5_leg"]["allreduce"] == "rccl"
    assert j["cfg3_leg"]["n_f_per_gpu"] == 1250 and j["cfg4_leg"]["n_f_per_gpu"] == 2500
    assert j["launch"] == "bench.py --gpus 8 (self-launched ranks)" and j["cpu_baseline"] is None
    assert abs(j["value"] - 10000 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]


def test_bench_self_launch_eight_ranks_one_failing_rank():
    rc, out, err = _self_launch(8, ["--gpus", "8", "--steps", "6", "--warmup", "3"], env={"BENCH_STUB_FAIL_RANK": "6"},
                                device_count=lambda: 8)
    assert rc == 7 and out == "" and "rank 6 of 8 exited with code 7" in err


def test_bench_rank_watchdog_at_eight_ranks():
    rc, out, err = _self_launch(8, ["--gpus", "8", "--steps", "6", "--warmup", "3", "--no-cfg34-legs", "--no-cfg5-leg"],
                                env={"BENCH_STUB_STALL_RANK": "5", "PINN_BENCH_RANK_TIMEOUT_S": "15"}, device_count=lambda: 8)
    assert rc == 124 and out == ""
    assert "of 8 exited with code 124" in err and "did not finish within 15 s" in err


def test_fit_refuses_replicas_that_drifted_apart(tmp_path):
    import json
    port = 29890 + os.getpid() % 9
    mp.spawn(_surface_worker, args=(2, port, str(tmp_path), "burgers", 1), nprocs=2, join=True)
    out = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert all("replicas hold different weights" in o["err"] for o in out), out
