"""CPU: host-side logic of the drop-in surface -- data preparation against the reference's own
prep_data (golden hashes), the portable custom_lbfgs driver against the reference's lbfgs
(known-answer fixture), Logger output bytes, the C ABI (library loads, exports every symbol
of include/pinn_hip.h), and argument validation that needs no GPU."""
import contextlib
import ctypes
import hashlib
import io
import json
import os
import re
import sys

import numpy as np
import pytest

from conftest import golden, BURGERS_MAT, NLS_MAT, ROOT


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_burgers_prep_data_matches_reference(burgers_sets):
    g = json.load(open(golden("burgers_data.json")))
    r = burgers_sets(100, 10000)
    assert len(r) == 12
    x, t, X, T, Exact_u, X_star, u_star, X_u, u, X_f, ub, lb = r
    assert lb.tolist() == g["inf"]["lb"] == [-1.0, 0.0] and ub.tolist() == g["inf"]["ub"]
    assert X_star.shape == (25600, 2) and X_f.shape == (10000, 2) and X_u.shape == (100, 2)
    for name, arr in (("X_f", X_f), ("X_u", X_u), ("u", u), ("X_star", X_star), ("u_star", u_star)):
        assert sha16(arr) == g["inf"]["sha"][name], name
    assert X_f[0].tolist() == [0.6853359174287899, 0.3944403153496481]
    r = burgers_sets(64, 2048)
    assert sha16(r[9]) == g["inf_small"]["sha"]["X_f"] and sha16(r[7]) == g["inf_small"]["sha"]["X_u"]


def test_burgers_identification_branch():
    import burgersutil
    g = json.load(open(golden("burgers_data.json")))
    np.random.seed(1234)
    r = burgersutil.prep_data(BURGERS_MAT, 10000, noise=0.0)
    assert len(r) == g["ide"]["n_returns"] == 11
    assert sha16(r[7]) == g["ide"]["sha"]["X_u"] and sha16(r[8]) == g["ide"]["sha"]["u"]
    assert r[10].tolist() == g["ide"]["lb"] and r[9].tolist() == g["ide"]["ub"]


def test_schrodinger_prep_data_matches_reference(schrodinger_sets):
    g = json.load(open(golden("schrodinger_data.json")))
    r = schrodinger_sets(50, 50, 20000)
    assert len(r) == 20
    names = dict(X_f=11, x0=15, tb=14, u0=16, v0=17, X_star=7, h_star=10, u_star=8, v_star=9)
    for k, i in names.items():
        assert sha16(r[i]) == g["sha"][k], k
    assert r[13].tolist() == [-5.0, 0.0] and abs(r[12][1] - np.pi / 2) < 1e-16
    assert np.all(r[18][:, 1] == 0) and np.array_equal(r[18][:, 0:1], r[15])      # X0 = (x0, 0)


def test_lhs_is_a_latin_hypercube():
    from sampling import lhs
    np.random.seed(7)
    H = lhs(2, 50)
    assert H.shape == (50, 2)
    for j in range(2):
        assert sorted(np.floor(H[:, j] * 50).astype(int).tolist()) == list(range(50))


def test_custom_lbfgs_known_answer():
    from custom_lbfgs import lbfgs, Struct
    import custom_lbfgs
    k = json.load(open(golden("lbfgs_kat.json")))
    A = np.diag(np.arange(1.0, 7.0)) + 0.1 * np.ones((6, 6))
    b = np.arange(1.0, 7.0)
    args = []

    def opfunc(x):
        args.append(np.array(x, copy=True))
        return 0.5 * x @ A @ x - b @ x + 0.25 * np.sum(x ** 4), A @ x - b + x ** 3

    cfg = Struct()
    cfg.learningRate, cfg.maxIter, cfg.nCorrection = 0.8, 8, 3
    cfg.tolFun = 1.0 * np.finfo(float).eps
    logs = []
    state = Struct()
    x, f_hist, n_eval = lbfgs(opfunc, np.zeros(6), cfg, state, True,
                              lambda it, f, is_iter: logs.append((it, f, is_iter)))
    assert np.allclose(f_hist, k["f_hist"], rtol=0, atol=1e-13)
    assert np.allclose(x, k["x_returned"], rtol=0, atol=1e-13)
    assert np.allclose(args[-1], k["last_opfunc_arg"], rtol=0, atol=1e-13)     # model = x_{maxIter-1}
    assert n_eval == k["n_eval"] == len(args) == 8
    assert [l[0] for l in logs] == [l[0] for l in k["logs"]] == list(range(1, 8))
    assert all(l[2] is True for l in logs)
    assert abs(custom_lbfgs.final_loss - k["final_loss_global"]) < 1e-13
    assert state.nIter == 8 and state.funcEval == 8 and len(state.old_dirs) == 3
    cfg0 = Struct()
    cfg0.maxIter = 0
    assert lbfgs(opfunc, np.zeros(6), cfg0, Struct(), True, None) is None
    assert Struct().anything == 0 and k["struct_default"] == 0
    # initial-point optimality: returns (x, f_hist) only
    out = lbfgs(lambda z: (0.0, np.zeros(3)), np.ones(3), cfg, Struct(), False, None)
    assert len(out) == 2


def test_logger_bytes_match_reference():
    from logger import Logger
    ref = json.load(open(golden("logger_bytes.json")))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        lg = Logger(ref["hp"])
        lg.set_error_fn(lambda: 0.0123456)
        lg.start_time = lg.prev_time = 0.0
        lg.log_train_start(None)
        lg.log_train_opt("Adam")
        lg.log_train_epoch(0, 0.26786867)
        lg.log_train_epoch(7, 0.5)
        lg.log_train_epoch(20, 0.058455, "l1 = 0.5", True)
        lg.log_train_end(300)
    mine, theirs = buf.getvalue(), ref["stdout"]

    def norm(s):
        s = re.sub(r"elapsed = \d\d:\d\d \(\+\d\d\.\d\)", "elapsed = MM:SS (+SS.S)", s)
        return re.sub(r"duration = \d\d:\d\d", "duration = MM:SS", s)
    # everything from "Training started" on is byte-identical (clock fields masked); the three
    # TensorFlow banner lines are replaced by engine/device lines
    cut = "\nTraining started"
    assert norm(mine[mine.index(cut):]) == norm(theirs[theirs.index(cut):])
    assert mine.startswith("Hyperparameters:\n" + json.dumps(ref["hp"], indent=2) + "\n\n")
    assert "tf_epoch =      0  elapsed = " in mine and "loss = 2.6787e-01  \n" in mine
    assert "nt_epoch =     20  " in mine and "loss = 5.8455e-02  l1 = 0.5\n" in mine
    assert "tf_epoch =      7" not in mine
    assert "GPU-accerelated: " in mine


def test_c_abi_exports_every_declared_symbol():
    import pinn_native
    lib = pinn_native.load()
    header = open(os.path.join(ROOT, "include", "pinn_hip.h")).read()
    declared = set(re.findall(r"\b(pinn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pinn_native.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pinn_abi_version() == 6
    # plain C types only in the header
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)          # strip comments
    assert "torch" not in code and "std::" not in code and "#include <stdint.h>" in code
    # ... and the header is valid ISO C on its own (what a cgo / ctypes / JNI binding generator would be fed)
    import shutil
    import subprocess
    if shutil.which("gcc"):
        res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                              os.path.join(ROOT, "include", "pinn_hip.h")], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr


def test_library_staleness_is_decided_by_source_contents_not_file_times(monkeypatch):
    """pinn_native._stale(): a library whose flags file records these flags and the digest of these sources is current
    whatever the file times say (a snapshot copied to a GPU box keeps no usable time order); any change of a source, or
    of the compile flags, makes it stale; a library named by PINN_HIP_LIB is never rebuilt"""
    import pinn_native
    pinn_native.load()                                          # builds if needed, writes libpinn_hip.so.flags
    tag = pinn_native.LIB_PATH + ".flags"
    assert os.path.exists(tag) and "sources-sha256 " in open(tag).read()
    assert not pinn_native._stale()
    src = os.path.join(os.path.dirname(os.path.dirname(pinn_native.LIB_PATH)), "csrc", "wave.h")
    st = os.stat(src)
    try:
        os.utime(src)                                           # newer than the library, same content
        assert not pinn_native._stale()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    monkeypatch.setattr(pinn_native, "_source_digest", lambda: "0" * 64)
    assert pinn_native._stale()
    monkeypatch.undo()
    monkeypatch.setattr(pinn_native, "COMMON_FLAGS", pinn_native.COMMON_FLAGS + ["-DSOMETHING"])
    assert pinn_native._stale()


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    import pinn_native
    lib = pinn_native.load()
    h = ctypes.c_void_p()
    lb = (ctypes.c_double * 2)(-1.0, 0.0)
    ub = (ctypes.c_double * 2)(1.0, 1.0)
    bad = (ctypes.c_int * 3)(3, 20, 1)                    # input dim must be 2
    rc = lib.pinn_create(ctypes.byref(h), bad, 3, lb, ub, 0, 0, 0)
    assert rc == -1 and b"input dimension" in lib.pinn_last_error()
    ragged = (ctypes.c_int * 4)(2, 20, 10, 1)             # unequal hidden widths
    rc = lib.pinn_create(ctypes.byref(h), ragged, 4, lb, ub, 0, 0, 0)
    assert rc == -1 and b"hidden widths" in lib.pinn_last_error()
    ok = (ctypes.c_int * 3)(2, 20, 1)
    assert lib.pinn_create(ctypes.byref(h), ok, 3, lb, ub, 2, 0, 0) == -1    # Schrodinger needs 2 outputs
    assert lib.pinn_create(ctypes.byref(h), ok, 3, lb, ub, 0, 7, 0) == -1    # dtype
    assert lib.pinn_destroy(None) == 0


def test_no_cpu_fallback_in_the_product_path():
    """The product package must not import the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "pinns-tf2.0_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_engine_construction_fails_loudly_without_gpu():
    import pinn_native
    if pinn_native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pinn_native.PinnNativeError):
        pinn_native.Engine([2, 20, 20, 1], [-1, 0], [1, 1])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
def test_ide_cont_repair_is_whitespace_only_and_parses():
    """tests/golden/repair_ide_cont.py holds an indent table, no reference text; applied to the reference's
    ide_cont_burgers.py it changes leading blanks only (what `diff -w` checks) and the result compiles"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import repair_ide_cont
    src = repair_ide_cont.check()
    assert "class BurgersInformedNN" in src
    with pytest.raises(SyntaxError):
        compile(open(repair_ide_cont.REF_FILE, encoding="utf-8").read(), "ide_cont_burgers.py", "exec")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        dst = repair_ide_cont.write(os.path.join(d, "1d-burgers", "ide_cont_burgers.py"))
        res = subprocess.run(["diff", "-w", repair_ide_cont.REF_FILE, dst], capture_output=True, text=True)
        assert res.returncode == 0 and res.stdout == ""


def test_no_reference_source_travels_with_the_tree():
    """the reference is Python: it is imported in the build container only (tests/golden/make_*.py, oracle/ref_baseline.py
    read /root/reference in place) and must not sit in the tree in any form -- source, bytecode or archive.  No file of the
    repository carries the name and the content of one of the reference's Python sources, and no archive is staged."""
    import hashlib
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref"))
    skip = {".git", "gpurun_out", "__pycache__", ".pytest_cache"}
    ref = "/root/reference"
    ref_hashes = set()
    if os.path.isdir(ref):
        for dp, dn, fn in os.walk(ref):
            dn[:] = [d for d in dn if d != ".git"]
            for f in fn:
                if f.endswith(".py"):
                    ref_hashes.add(hashlib.sha256(open(os.path.join(dp, f), "rb").read()).hexdigest())
    for dp, dn, fn in os.walk(ROOT):
        dn[:] = [d for d in dn if d not in skip]
        for f in fn:
            path = os.path.join(dp, f)
            assert not f.endswith((".tar", ".tar.gz", ".tgz", ".zip", ".pyc")), path
            if ref_hashes and f.endswith(".py"):
                assert hashlib.sha256(open(path, "rb").read()).hexdigest() not in ref_hashes, path


def test_port_fit_log_matches_the_reference_run():
    """oracle/fit.py (what bench.py's cpu_baseline times) against the reference's own printed log of the default schedule's
    Adam phase (tests/golden/burgers_default_run.json): same epochs logged, same losses to the printed digits"""
    import json
    import burgersutil
    from oracle import fit, init
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "burgers_default_run.json")))
    hp = ref["hp"]
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(ROOT, "pinns-tf2.0_amd", "1d-burgers", "data", "burgers_shock.mat"),
                              hp["N_u"], hp["N_f"], noise=0.0)
    X_star, u_star, X_u, u, X_f, ub, lb = r[5], r[6], r[7], r[8], r[9], r[10], r[11]
    res = fit.burgers_fit(init.glorot_flat(hp["layers"]), hp["layers"], lb, ub, X_f, X_u, u, 0.01 / np.pi, X_star[::50], u_star[::50],
                          tf_epochs=30, nt_epochs=20, log_frequency=hp["log_frequency"])
    mine = [l for l in res["lines"] if l.startswith(("tf_epoch", "nt_epoch"))]
    want = [l for l in ref["lines"] if l.startswith("tf_epoch")][:3]
    assert len(mine) == 3 + 1 and res["evals"] == 50      # L-BFGS iterations 1..19 are logged at multiples of 10; the 20th breaks first (custom_lbfgs.py:192)
    for a, b in zip(mine, want):
        assert a.split("=")[1].split()[0] == b.split("=")[1].split()[0]            # epoch number
        assert a.split("loss = ")[1].strip() == b.split("loss = ")[1].strip()      # the printed loss


# ---- log lines one chunk behind the GPU (NeuralNetwork._pipelined), scripted engine ------------------------------------------
class _TicketEngine(object):
    """the enqueue / collect surface: records the ORDER of calls; a chunk's losses are a function of the epoch alone, so the
    pipelined and the synchronous loops must print the same lines"""

    def __init__(self, layers, lb, ub, pde="burgers", dtype="f64", device=0):
        self.n_params, self.w, self.calls = 5, np.zeros(5), []
        self.n_f = self.n_u = self.n_b = 0
        self.t, self.epoch, self.pending = 0, 0, {}
        self.lb_total = self.lb_issued = self.lb_logged = 0

    def set_weights(self, w): self.w = np.array(w, dtype=np.float64)
    def get_weights(self): return self.w.copy()
    def adam_init(self, *a): pass
    def set_data(self, X, u, n_total=None): pass
    def status(self): return 0, 0
    def _adam(self, n): e = np.arange(self.epoch, self.epoch + n); self.epoch += n; return 1.0 / (1.0 + e)
    def adam_run(self, n, want_losses=True): self.calls.append(("run", n)); return self._adam(n)

    def adam_enqueue(self, n):
        self.t += 1; self.calls.append(("enq", self.t, n)); self.pending[self.t] = self._adam(n); return self.t

    def adam_collect(self, t): self.calls.append(("col", t)); return self.pending.pop(t)
    def lbfgs_begin(self, n, *a): self.lb_total, self.lb_issued, self.lb_logged = n, 0, 0

    def _lb(self, n):                       # iterations 1 .. total-1 are logged (the last one breaks first, custom_lbfgs.py:192)
        k = min(n, self.lb_total - self.lb_issued)
        self.lb_issued += k
        upto = min(self.lb_issued, self.lb_total - 1)
        its = np.arange(self.lb_logged + 1, upto + 1, dtype=np.int32)
        self.lb_logged = upto
        return its, 0.5 / (1.0 + its), int(self.lb_issued >= self.lb_total)

    def lbfgs_run(self, n): self.calls.append(("lrun", n)); return self._lb(n)

    def lbfgs_enqueue(self, n):
        self.t += 1; self.calls.append(("lenq", self.t, n)); self.pending[self.t] = self._lb(n); return self.t

    def lbfgs_collect(self, t): self.calls.append(("lcol", t)); return self.pending.pop(t)


def _ticket_model(monkeypatch, **hp_extra):
    p = os.path.join(ROOT, "pinns-tf2.0_amd", "utils")
    if p not in sys.path:
        sys.path.insert(0, p)
    import neuralnetwork
    from logger import Logger
    monkeypatch.setattr(neuralnetwork, "Engine", _TicketEngine)
    hp = dict({"layers": [2, 1], "tf_epochs": 25, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 34, "nt_lr": 0.8,
               "nt_ncorr": 50, "log_frequency": 10}, **hp_extra)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        nn = neuralnetwork.NeuralNetwork(hp, Logger(hp), [1.0, 1.0], [-1.0, 0.0])
        nn.logger.set_error_fn(lambda: 0.5)
        nn.fit(np.zeros((4, 2)), np.zeros((4, 1)))
    strip = re.compile(r"elapsed = \S+ \(\+\S+\)")
    lines = [strip.sub("", l) for l in buf.getvalue().splitlines() if l.startswith(("tf_epoch", "nt_epoch", "Training", "--"))]
    return nn, lines


def test_pipelined_logging_prints_the_synchronous_lines_one_chunk_behind(monkeypatch):
    sync, want = _ticket_model(monkeypatch, async_log=False)
    assert [c[0] for c in sync._engine.calls if c[0] in ("enq", "lenq")] == []
    assert [c for c in sync._engine.calls if c[0] == "run"] == [("run", 1), ("run", 10), ("run", 10), ("run", 4)]
    nn, got = _ticket_model(monkeypatch)
    assert got == want and len([l for l in got if l.startswith("tf_epoch")]) == 3 and len([l for l in got if l.startswith("nt_epoch")]) == 3
    calls = nn._engine.calls
    # Adam: the same four chunks; chunk k + 1 is in the stream BEFORE chunk k is collected, never more than two in flight
    assert [c for c in calls if c[0] in ("enq", "col")] == [("enq", 1, 1), ("enq", 2, 10), ("col", 1), ("enq", 3, 10), ("col", 2),
                                                            ("enq", 4, 4), ("col", 3), ("col", 4)]
    # L-BFGS: chunks of log_frequency iterations, one ahead; `done` is seen one chunk late and the chunk that ran ahead is collected too
    lb = [c for c in calls if c[0] in ("lenq", "lcol")]
    assert lb[:3] == [("lenq", 5, 10), ("lenq", 6, 10), ("lcol", 5)] and lb[-1][0] == "lcol"
    assert sorted(c[1] for c in lb if c[0] == "lenq") == sorted(c[1] for c in lb if c[0] == "lcol")     # nothing left in flight
    assert not nn._engine.pending


def test_pipelined_logging_steps_aside_for_lines_that_need_the_device(monkeypatch):
    """the restart guard judges a chunk before the next one may start; a subclass that appends device state to a line
    (_log_custom) or prints per evaluation (_adam_chunk) needs the weights of exactly that epoch"""
    nn, _ = _ticket_model(monkeypatch, nt_guard=1e3)
    assert [c[0] for c in nn._engine.calls if c[0] in ("lenq", "lrun")] == ["lrun"] * 4       # guard on: synchronous L-BFGS
    assert any(c[0] == "enq" for c in nn._engine.calls)                                        # (Adam is still pipelined)
    import neuralnetwork

    class Custom(neuralnetwork.NeuralNetwork):
        def _log_custom(self):
            return "l1 = 1"
    from logger import Logger
    hp = {"layers": [2, 1], "tf_epochs": 12, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 12, "nt_lr": 0.8,
          "nt_ncorr": 50, "log_frequency": 10}
    with contextlib.redirect_stdout(io.StringIO()):
        nn = Custom(hp, Logger(hp), [1.0, 1.0], [-1.0, 0.0])
        nn.logger.set_error_fn(lambda: 0.5)
        nn.fit(np.zeros((4, 2)), np.zeros((4, 1)))
    assert not any(c[0] in ("enq", "lenq") for c in nn._engine.calls)


class _StatefulEngine(object):
    """a deterministic model of the engine for the restart guard: the loss of an iteration is a function of (how many times
    L-BFGS was begun, global iteration number) -- it explodes where EXPLODE says -- and the 'weights' are the global iteration
    reached.  With snapshot = True it offers the ticketed loops and device-side snapshots, otherwise only the synchronous calls."""
    EXPLODE = {(1, 13): 3e5, (1, 14): 1e9, (2, 27): float("nan")}       # begin #1 explodes at iteration 13, begin #2 at 27

    def __init__(self, layers, lb, ub, pde="burgers", dtype="f32", device=0):
        self.n_params, self.pos, self.calls = 5, 0, []
        self.n_f = self.n_u = self.n_b = 0
        self.begins, self.base, self.left, self.t, self.pending, self.snaps = 0, 0, 0, 0, {}, {}

    def set_weights(self, w): self.pos = int(round(np.asarray(w, dtype=np.float64).ravel()[0])); self.calls.append(("set", self.pos))
    def get_weights(self): self.calls.append(("get", self.pos)); return np.full(5, float(self.pos))
    def adam_init(self, *a): pass
    def set_data(self, X, u, n_total=None): pass
    def status(self): return 0, 0
    def lbfgs_begin(self, n, *a): self.begins += 1; self.base, self.left, self.pending = self.pos, n, {}; self.calls.append(("begin", n))

    def _chunk(self, n):
        k = min(n, self.left)
        self.left -= k
        last = self.pos + k - (1 if self.left == 0 else 0)               # the last iteration breaks before it logs
        its = np.arange(self.pos + 1, last + 1)
        losses = np.array([self.EXPLODE.get((self.begins, int(i)), 1.0 / (1.0 + i)) for i in its])
        self.pos += k
        return (its - self.base).astype(np.int32), losses, int(self.left == 0)

    def lbfgs_run(self, n): self.calls.append(("run", n)); return self._chunk(n)


class _StatefulTicketEngine(_StatefulEngine):
    N_SNAPSHOTS = 4
    def lbfgs_enqueue(self, n): self.t += 1; self.pending[self.t] = self._chunk(n); self.calls.append(("enq", self.t)); return self.t
    def lbfgs_collect(self, t): self.calls.append(("col", t)); return self.pending.pop(t)
    def adam_enqueue(self, n): raise AssertionError("no Adam in this test")
    def adam_collect(self, t): raise AssertionError("no Adam in this test")
    def weights_snapshot(self, slot): self.snaps[slot] = self.pos; self.calls.append(("snap", slot, self.pos))
    def weights_restore(self, slot): self.pos = self.snaps[slot]; self.calls.append(("restore", slot, self.pos))


def test_restart_guard_one_chunk_behind_takes_the_synchronous_decisions(monkeypatch):
    """round 6: with the guard on, nt_optimization stays one chunk behind the GPU as well -- its way back is a device-side
    snapshot behind every chunk instead of a host copy.  Same explosions -> same discarded chunks, same restart points, same
    printed lines as the synchronous loop; the chunk that ran ahead of a bad one is dropped with it."""
    p = os.path.join(ROOT, "pinns-tf2.0_amd", "utils")
    if p not in sys.path:
        sys.path.insert(0, p)
    import neuralnetwork
    from logger import Logger
    out = {}
    for name, cls in (("sync", _StatefulEngine), ("pipelined", _StatefulTicketEngine)):
        monkeypatch.setattr(neuralnetwork, "Engine", cls)
        hp = {"layers": [2, 1], "tf_epochs": 0, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 60, "nt_lr": 0.8,
              "nt_ncorr": 50, "log_frequency": 10, "dtype": "f32"}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            nn = neuralnetwork.NeuralNetwork(hp, Logger(hp), [1.0, 1.0], [-1.0, 0.0])
            nn._engine.calls.clear()
            nn._engine.pos = 0                        # (the constructor handed the engine glorot weights: start the model at 0)
            nn.nt_optimization(np.zeros((4, 2)), np.zeros((4, 1)))
        strip = re.compile(r"elapsed = \S+ \(\+\S+\)")
        out[name] = (nn.nt_restarts, [strip.sub("", l) for l in buf.getvalue().splitlines() if l.startswith("nt_epoch")],
                     nn._engine.pos, nn._engine.calls)
    assert out["sync"][0] == out["pipelined"][0] == [(13, 10, 1.0), (27, 20, 1.0)]
    assert out["sync"][1] == out["pipelined"][1] and [l.split()[2] for l in out["sync"][1]] == ["10", "20", "30", "40", "50"]
    assert out["sync"][2] == out["pipelined"][2] == 60                        # both end at the same iterate
    calls = out["pipelined"][3]
    assert not any(c[0] in ("get", "set", "run") for c in calls)              # nothing crosses the bus, nothing synchronises
    assert [c for c in calls if c[0] == "restore"] == [("restore", 1, 10), ("restore", 2, 20)] or \
        [c[2] for c in calls if c[0] == "restore"] == [10, 20]
    assert not out["pipelined"][3][-1][0] == "enq" and not nn._engine.pending   # nothing left in flight
    assert [c for c in out["sync"][3] if c[0] == "set"] == [("set", 10), ("set", 20)]


# ---- the L-BFGS restart guard of NeuralNetwork.nt_optimization (hp["nt_guard"]), scripted engine ---------------------------
class _GuardEngine(object):
    """lbfgs_run follows a script of per-chunk loss lists; records the calls the guard makes"""

    def __init__(self, layers, lb, ub, pde="burgers", dtype="f32", device=0):
        self.n_params, self.w, self.calls = 5, np.zeros(5), []
        self.n_f = self.n_u = self.n_b = 0
        self.script, self.it, self.left = [], 0, 0

    def set_weights(self, w): self.w = np.array(w, dtype=np.float64); self.calls.append(("set_weights", self.w.copy()))
    def get_weights(self): return self.w.copy()
    def adam_init(self, *a): pass
    def set_data(self, X, u, n_total=None): pass
    def status(self): return 0, 0
    def lbfgs_begin(self, n, *a): self.calls.append(("begin", n)); self.lrs = getattr(self, "lrs", []) + [a[0]]; self.it, self.left = 0, n

    def lbfgs_run(self, n):
        losses = np.array(self.script.pop(0), dtype=np.float64)
        its = np.arange(self.it + 1, self.it + 1 + len(losses), dtype=np.int32)
        self.it += len(losses); self.left -= len(losses)
        self.w = self.w + len(losses)                   # "weights" = iterations applied since the last set_weights
        return its, losses, int(not self.script)


def _guarded_model(monkeypatch, hp_extra, script):
    for p in (os.path.join(ROOT, "pinns-tf2.0_amd", "utils"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    import neuralnetwork
    from logger import Logger
    monkeypatch.setattr(neuralnetwork, "Engine", _GuardEngine)
    hp = dict({"layers": [2, 1], "tf_epochs": 0, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 40, "nt_lr": 0.8,
               "nt_ncorr": 50, "log_frequency": 10}, **hp_extra)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        nn = neuralnetwork.NeuralNetwork(hp, Logger(hp), [1.0, 1.0], [-1.0, 0.0])
        nn._engine.script = script
        nn._engine.calls.clear()
        nn.w_start = nn._engine.w.copy()
        nn.nt_optimization(np.zeros((4, 2)), np.zeros((4, 1)))
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("nt_epoch")]
    return nn, lines


def test_nt_guard_discards_an_exploding_chunk_and_restarts_from_the_last_accepted_one(monkeypatch, capsys):
    ok1 = list(np.linspace(1.0, 0.5, 10))
    boom = [0.4, 0.3, 3.0e5, 1e9, 1e8, 1e7, 1e6, 1e5, 1e4, 1e3]          # iteration 13 jumps by 6 decades
    ok2 = list(np.linspace(0.5, 0.3, 10))
    ok3 = list(np.linspace(0.3, 0.2, 10))
    ok4 = list(np.linspace(0.2, 0.1, 9))
    nn, lines = _guarded_model(monkeypatch, {"dtype": "f32"}, [ok1, boom, ok2, ok3, ok4])
    calls = nn._engine.calls
    assert [c for c in calls if c[0] == "begin"] == [("begin", 40), ("begin", 30)]      # 30 iterations were left
    restored = [c[1] for c in calls if c[0] == "set_weights"]
    assert len(restored) == 1 and np.array_equal(restored[0], nn.w_start + 10.0)        # the weights after chunk 1
    assert nn.nt_restarts == [(13, 10, 1.0)]
    assert [int(l.split()[2]) for l in lines] == [10, 20, 30]          # numbering continues; the discarded chunk is not logged
    assert "nt_guard: loss 3.000e+05 at L-BFGS iteration 13" in capsys.readouterr().err


def test_nt_guard_is_off_in_the_reference_arithmetic_and_bounded_when_on(monkeypatch):
    boom = [0.4, 0.3, 3.0e5, float("nan"), 1e8, 1e7, 1e6, 1e5, 1e4, 1e3]
    nn, lines = _guarded_model(monkeypatch, {"dtype": "f64"}, [list(np.ones(10)), boom, list(np.ones(10)), list(np.ones(9))])
    assert nn._nt_guard == 0.0 and nn.nt_restarts == [] and [c for c in nn._engine.calls if c[0] == "begin"] == [("begin", 40)]
    assert len(lines) == 3                                             # float64 = the reference: nothing is discarded
    # on, and every retry explodes again: MAX_RESTARTS discards, then the run is left alone
    script = [list(np.ones(10))] + [boom] * 5 + [boom, list(np.ones(10)), list(np.ones(9))]
    nn, lines = _guarded_model(monkeypatch, {"dtype": "f32"}, script)
    assert len(nn.nt_restarts) == nn.MAX_RESTARTS == 5 and len([c for c in nn._engine.calls if c[0] == "begin"]) == 6
    # a restart from the same boundary would replay the explosion bit for bit -- every repeat halves the step;
    # and once the restarts are spent the exploded chunk is logged (the run is left alone) but never becomes a restart point
    assert np.allclose(nn._engine.lrs, [0.8, 0.8, 0.4, 0.2, 0.1, 0.05])
    assert all(r[1] == 10 for r in nn.nt_restarts) and [r[2] for r in nn.nt_restarts] == [1.0, 0.5, 0.25, 0.125, 0.0625]   # the factor is on record
    nn, _ = _guarded_model(monkeypatch, {"dtype": "f64", "nt_guard": 100.0}, [list(np.ones(10)), [200.0] * 10, list(np.ones(10)), list(np.ones(10)), list(np.ones(9))])
    assert nn.nt_restarts == [(11, 10, 1.0)]                                # explicit hp key wins over the dtype default


def test_t16_fused_launch_plan_deals_every_tile_and_every_row_exactly_once():
    """csrc/kernels_tile16f.h t16_deal (host side of k_t16_fused, round 5): for every hidden width the fused float64 sweep
    takes (65..128) the eight waves' ranges partition the full gradient tiles and the strips, the rows of the layer GEMMs
    are contiguous and cover [0, 4 ceil(W / 4)), nobody owns more than four strips, and the matrix time of a reversed
    layer -- in units of one 16x16x4 instruction -- is spread within 12 % over the four SIMDs (waves w and w + 4 share one)"""
    import ctypes
    import pinn_native
    lib = pinn_native.load()
    for W in range(65, 129):
        out = (ctypes.c_int * 49)()
        assert lib.pinn_debug_t16_deal(W, out) == 0
        o = list(out)
        f_lo, f_hi, e_lo, e_hi, row0, ns, edge = o[0:8], o[8:16], o[16:24], o[24:32], o[32:40], o[40:48], o[48]
        ntl, ksteps = (W + 15) // 16, (W + 3) // 4
        assert edge == (1 if 1 <= W % 16 <= 4 else 0), W
        nfs = ntl - 1 if edge else ntl
        n_full, n_edge = nfs * nfs, (2 * ntl - 1 if edge else 0)
        for lo, hi, n in ((f_lo, f_hi, n_full), (e_lo, e_hi, n_edge)):
            assert lo[0] == 0 and hi[7] == n and all(lo[w] <= hi[w] for w in range(8)), (W, lo, hi, n)
            assert all(hi[w] == lo[w + 1] for w in range(7)), (W, lo, hi)
        assert row0[0] == 0 and all(0 <= n <= 4 for n in ns), (W, ns)
        assert all(row0[w + 1] == row0[w] + 4 * ns[w] for w in range(7)), (W, row0, ns)
        assert row0[7] + 4 * ns[7] == 4 * ksteps and sum(ns) == ksteps, (W, row0, ns)
        cost = [ns[w] * ksteps + 16 * (f_hi[w] - f_lo[w]) + 4 * (e_hi[w] - e_lo[w]) for w in range(8)]
        simd = [cost[w] + cost[w + 4] for w in range(4)]
        assert max(simd) <= 1.12 * (sum(simd) / 4.0), (W, simd)
    out = (ctypes.c_int * 49)()
    lib.pinn_debug_t16_deal(100, out)
    assert list(out)[40:48] == [4, 4, 4, 4, 2, 2, 2, 3] and list(out)[32:40] == [0, 16, 32, 48, 64, 72, 80, 88]


def test_runtime_binding_policies_give_one_hip_runtime_per_process():
    """pinn_native._bind_runtime (round 5): the image holds /opt/rocm and torch's bundled ROCm under the same
    sonames.  auto: a plain process binds /opt/rocm, a rank of WORLD_SIZE > 1 binds torch's set (torch imported first) and
    still maps ONE runtime after torch.distributed is in use; torch: the same at world size 1; rocm: refuses once torch is in
    the process.  No device is touched (pinn_runtime_versions asks the libraries, not a GPU)."""
    import subprocess
    code = ("import sys, os, json; sys.path.insert(0, %r); import pinn_native\n"
            "if os.environ.get('PRE_TORCH'): import torch\n"
            "try:\n"
            "    pinn_native.load(); info = pinn_native.runtime_info()\n"
            "    if os.environ.get('POST_TORCH'):\n"
            "        import torch.distributed; info = pinn_native.runtime_info()\n"
            "    print(json.dumps(info))\n"
            "except pinn_native.PinnNativeError as e:\n"
            "    print(json.dumps({'error': str(e)}))\n") % os.path.join(ROOT, "pinns-tf2.0_amd")

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PINN_HIP_RUNTIME", "PINN_HIP_LIB")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    plain = run()
    assert plain["bound"] == "rocm" and plain["single_runtime"] and "/opt/rocm" in plain["libamdhip64_path"]
    assert plain["hip_runtime"] > 0 and plain["rccl_version"].count(".") == 2
    rank = run(WORLD_SIZE="2", RANK="0", POST_TORCH="1")
    assert rank["bound"] == "torch" and rank["single_runtime"] and "/torch/lib/" in rank["librccl_path"]
    forced = run(PINN_HIP_RUNTIME="torch", POST_TORCH="1")
    assert forced["bound"] == "torch" and forced["single_runtime"]
    assert forced["hip_runtime"] != plain["hip_runtime"] or forced["libamdhip64_path"] != plain["libamdhip64_path"]
    refused = run(PINN_HIP_RUNTIME="rocm", PRE_TORCH="1")
    assert "two HIP runtimes" in refused.get("error", "")
    assert "must be auto, torch or rocm" in run(PINN_HIP_RUNTIME="bogus").get("error", "")
