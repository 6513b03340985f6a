"""End-to-end drop-in tests (pytest -m gpu): the scripts a user of the reference runs
(`python 1d-burgers/inf_cont_burgers.py [hp.json]` etc.) executed as subprocesses on the GPU, their
stdout compared with the reference's own printed log (tests/golden/burgers_default_run.json =
the unmodified reference script over the test shims, see make_golden.py).

f64 kernels: printed losses of the default schedule (100 Adam + 200 L-BFGS) agree with the
reference's to 4 significant digits for the whole Adam phase and the first 100 L-BFGS iterations
(after that two float64 implementations of the same mathematics already drift); the
final error must land inside the reference's own 1-ulp sensitivity band [0.24, 0.31].
f32 kernels: same schedule, first Adam losses to 3 digits, final error inside the same band."""
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import PKG, golden

pytestmark = pytest.mark.gpu

LINE = re.compile(r"^(tf_epoch|nt_epoch) =\s+(\d+)\s+elapsed = \d\d:\d\d \(\+\d\d\.\d\)  loss = (\S+)  ")
END = re.compile(r"^Training finished \(epoch (\d+)\): duration = \d\d:\d\d  error = (\S+)  ")


def run_script(rel, hp, tmp_path, extra_env=None):
    hp_file = tmp_path / "hp.json"
    hp_file.write_text(json.dumps(hp))
    env = dict(os.environ, PINN_NO_PLOT="1", MPLBACKEND="Agg")
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, os.path.join(PKG, rel), str(hp_file)], cwd=PKG, env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return res.stdout


def parse(stdout):
    rows, end = [], None
    for line in stdout.splitlines():
        m = LINE.match(line)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
        m = END.match(line)
        if m:
            end = (int(m.group(1)), float(m.group(2)))
    return rows, end


# relative deviation of the printed losses allowed per phase: (Adam epochs 0..90, L-BFGS its <= 50, L-BFGS its <= 100)
# (the log prints 5 significant digits: 5e-5 is its resolution).  Measured on MI355X: float64 identical in every printed
# digit through L-BFGS iteration 100 (5 % at the end of the 200); float32 1.1e-5 over the 100 Adam epochs, 8e-3 /
# 3.8e-2 after 50 / 100 L-BFGS iterations (13 % at the end) -- the schedule amplifies roundoff, see test_gpu_end_to_end.py
LOG_TOL = {"f64": (1.5e-4, 1.5e-4, 1.5e-4), "f32": (1.5e-4, 3e-2, 1.2e-1)}


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_inf_cont_burgers_default_run_matches_reference_log(tmp_path, record, dtype):
    g = json.load(open(golden("burgers_default_run.json")))
    b = json.load(open(golden("burgers_band.json")))
    hp = dict(g["hp"], dtype=dtype)
    out = run_script(os.path.join("1d-burgers", "inf_cont_burgers.py"), hp, tmp_path)
    assert "Training started" in out and "-- Starting Adam optimization --" in out
    rows, end = parse(out)
    ref, _ = parse("\n".join(g["lines"]))
    assert [(r[0], r[1]) for r in rows] == [(r[0], r[1]) for r in ref]      # same lines, same epochs
    dev = {"adam": 0.0, "lbfgs50": 0.0, "lbfgs100": 0.0, "lbfgs_all": 0.0}
    for (kind, ep, loss), (_, _, loss_ref) in zip(rows, ref):
        d = abs(loss - loss_ref) / loss_ref
        if kind == "tf_epoch":
            dev["adam"] = max(dev["adam"], d)
        else:
            if ep <= 50:
                dev["lbfgs50"] = max(dev["lbfgs50"], d)
            if ep <= 100:
                dev["lbfgs100"] = max(dev["lbfgs100"], d)
            dev["lbfgs_all"] = max(dev["lbfgs_all"], d)
    record(dtype=dtype, final_error=end[1], **dev)
    ta, t50, t100 = LOG_TOL[dtype]
    assert dev["adam"] <= ta and dev["lbfgs50"] <= t50 and dev["lbfgs100"] <= t100, dev
    assert end is not None and end[0] == 300
    from conftest import ensemble_accepts
    ok, med, radius = ensemble_accepts([v["final_error"] for v in b["runs"].values()], end[1])
    assert ok, (end, med, radius)                                           # reference: 2.6564e-01


def test_ide_cont_burgers_run_matches_reference_log(tmp_path):
    """1d-burgers/ide_cont_burgers.py, 100 Adam + 100 L-BFGS, both models (clean + the "noise" rerun), float64,
    against the printed log and lambdas of the reference's own script (whitespace-repaired, make_golden.py):
    every Adam line to 1.5e-4, the L-BFGS lines to 2 %, the identified lambdas to 2 %."""
    g = json.load(open(golden("burgers_ide_run.json")))
    hp = dict(g["hp"], dtype="f64")
    out = run_script(os.path.join("1d-burgers", "ide_cont_burgers.py"), hp, tmp_path)
    mine = [l for l in out.splitlines() if l.startswith(("tf_epoch", "nt_epoch"))]
    ref = [l for l in g["lines"] if l.startswith(("tf_epoch", "nt_epoch"))]
    assert len(mine) == len(ref) == 2 * (10 + 9)
    for a, b in zip(mine, ref):
        ma, mb = LINE.match(a), LINE.match(b)
        assert (ma.group(1), ma.group(2)) == (mb.group(1), mb.group(2))
        la, lb_ = float(ma.group(3)), float(mb.group(3))
        assert abs(la - lb_) <= (2e-2 if ma.group(1) == "nt_epoch" else 1.5e-4) * lb_, (a, b)
    vals = dict(re.findall(r"^(l1|l2|l1_noise|l2_noise):\s+(\S+)$", out, flags=re.M))
    for key, ref_key in (("l1", "lambda_1"), ("l2", "lambda_2"), ("l1_noise", "lambda_1_noise"), ("l2_noise", "lambda_2_noise")):
        assert abs(float(vals[key]) - g[ref_key]) <= 2e-2 * abs(g[ref_key]), (key, vals[key], g[ref_key])
    ends = [float(m.group(2)) for m in map(END.match, out.splitlines()) if m]
    ref_ends = [float(m.group(2)) for m in map(END.match, g["lines"]) if m]
    assert len(ends) == len(ref_ends) == 2
    for a, b in zip(ends, ref_ends):
        assert abs(a - b) <= 2e-2 * b, (ends, ref_ends)


def test_inf_cont_schrodinger_log_and_loss_parts_match_reference(tmp_path):
    """1dcomplex-schrodinger/inf_cont_schrodinger.py, 10 Adam epochs, float64, defaults (= the reference's
    fit(x0 [N0,1], ...) call): the progress lines, the per-evaluation `mse_0 / mse_b / mse_f` lines of loss() (:128)
    and the final error on |h| against the reference script's own stdout"""
    g = json.load(open(golden("schrodinger_run.json")))
    out = run_script(os.path.join("1dcomplex-schrodinger", "inf_cont_schrodinger.py"), dict(g["hp"], dtype="f64"), tmp_path)
    rows, end = parse(out)
    ref, ref_end = parse("\n".join(g["lines"]))
    assert [(r[0], r[1]) for r in rows] == [(r[0], r[1]) for r in ref] and len(ref) == 10
    for (_, ep, loss), (_, _, loss_ref) in zip(rows, ref):
        assert abs(loss - loss_ref) <= 1.5e-4 * loss_ref, (ep, loss, loss_ref)
    mse = [[float(t[1]), float(t[3]), float(t[5])] for t in (l.split() for l in out.splitlines() if l.startswith("mse_0"))]
    assert len(mse) == len(g["mse_0_b_f"]) == 10
    for a, b in zip(mse, g["mse_0_b_f"]):
        for x, y in zip(a, b):
            assert abs(x - y) <= 1e-8 * max(abs(y), 1e-12), (a, b)
    assert end is not None and abs(end[1] - g["final_error"]) <= 6e-5 * g["final_error"], (end, g["final_error"])   # 5 printed digits


def test_inf_cont_schrodinger_default_schedule_matches_reference(tmp_path, record):
    """BASELINE configs[3]: the script's own defaults (200 Adam epochs at lr .05 / beta_1 .99 / eps .1 on N_f = 20000,
    4x100; inf_cont_schrodinger.py:23-41,164), float64.  This optimiser regime is benign, so -- unlike the Burgers
    schedules -- the END of the run is a reproducible quantity: every printed loss to 4 digits and the final
    relative L2 error of |h| within north_star's 1e-3 of the reference's own run."""
    g = json.load(open(golden("schrodinger_default_run.json")))
    out = run_script(os.path.join("1dcomplex-schrodinger", "inf_cont_schrodinger.py"),
                     dict(g["hp"], dtype="f64", quiet_loss_parts=True), tmp_path)
    rows, end = parse(out)
    ref, ref_end = parse("\n".join(g["lines"]))
    assert [(r[0], r[1]) for r in rows] == [(r[0], r[1]) for r in ref] and len(ref) == 20
    dev = max(abs(a[2] - b[2]) / b[2] for a, b in zip(rows, ref))
    record(max_loss_dev=dev, err_gpu=end[1], err_ref=g["final_error"])
    assert dev <= 1.5e-4, dev
    # measured: every printed loss identical to its 5 digits, final error 0.79319 vs 0.7931950 (north_star asks 1e-3)
    assert end is not None and end[0] == 200 and abs(end[1] - g["final_error"]) <= 1e-4, (end, g["final_error"])


def test_ide_cont_burgers_cfg3_default_schedule(tmp_path, record):
    """BASELINE configs[2]: identification with N_u = 10000, 100 Adam + 500 L-BFGS, both models, float64, against the
    reference's own run (whitespace-repaired ide_cont_burgers.py over the shims).  Adam lines to 1.5e-4; L-BFGS without
    line search drifts apart between any two float64 implementations after ~100 iterations, so the tail is held to the
    identified physics: lambda_1 and lambda_2 = exp(l2) of both models against the reference's."""
    g = json.load(open(golden("burgers_ide_cfg3_run.json")))
    out = run_script(os.path.join("1d-burgers", "ide_cont_burgers.py"), dict(g["hp"], dtype="f64"), tmp_path)
    mine = [LINE.match(l) for l in out.splitlines() if l.startswith(("tf_epoch", "nt_epoch"))]
    ref = [LINE.match(l) for l in g["lines"] if l.startswith(("tf_epoch", "nt_epoch"))]
    assert len(mine) == len(ref) and all(mine) and all(ref)
    adam_dev, lb_dev = 0.0, 0.0
    for a, b in zip(mine, ref):
        assert (a.group(1), a.group(2)) == (b.group(1), b.group(2))
        d = abs(float(a.group(3)) - float(b.group(3))) / float(b.group(3))
        if a.group(1) == "tf_epoch":
            adam_dev = max(adam_dev, d)
        elif int(a.group(2)) <= 100:
            lb_dev = max(lb_dev, d)
    vals = dict(re.findall(r"^(l1|l2|l1_noise|l2_noise):\s+(\S+)$", out, flags=re.M))
    devs = {k: abs(float(vals[k]) - g[r]) / abs(g[r]) for k, r in (("l1", "lambda_1"), ("l2", "lambda_2"),
                                                                   ("l1_noise", "lambda_1_noise"), ("l2_noise", "lambda_2_noise"))}
    record(adam_dev=adam_dev, lbfgs_dev_first_100=lb_dev, **{"dev_" + k: v for k, v in devs.items()},
           l1=float(vals["l1"]), l2=float(vals["l2"]), l1_ref=g["lambda_1"], l2_ref=g["lambda_2"])
    assert adam_dev <= 1.5e-4 and lb_dev <= 2e-2, (adam_dev, lb_dev)
    for k, v in devs.items():          # measured 0.1 ... 0.4 % after the 500 L-BFGS iterations
        assert v <= 0.02, (k, v, vals)


def test_default_arithmetic_is_the_references_float64(monkeypatch):
    """without an hp["dtype"] key the drop-in computes in float64 like the reference (utils/neuralnetwork.py:24-26), on
    the float64 register-stash kernel; "f32" is the opt-in throughput mode"""
    import importlib
    import numpy as np
    for p in (os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers")):
        if p not in sys.path:
            sys.path.insert(0, p)
    monkeypatch.setattr(sys, "argv", ["inf_cont_burgers.py"])
    inf = importlib.import_module("inf_cont_burgers")
    from logger import Logger
    hp = dict(json.load(open(golden("burgers_default_run.json")))["hp"], tf_epochs=0, nt_epochs=0)
    assert "dtype" not in hp
    X_f = np.random.RandomState(0).uniform([-1, 0], [1, 0.99], (512, 2))
    a = inf.BurgersInformedNN(hp, Logger(hp), X_f, np.array([1.0, 0.99]), np.array([-1.0, 0.0]), nu=0.01 / np.pi)
    assert a.compute_dtype == "f64" and a._engine.dtype == "f64" and a._engine.kernel_path() == 7
    b = inf.BurgersInformedNN(dict(hp, dtype="f32"), Logger(hp), X_f, np.array([1.0, 0.99]), np.array([-1.0, 0.0]), nu=0.01 / np.pi)
    assert b._engine.dtype == "f32" and b._engine.kernel_path() == 2


def test_plotting_and_result_directory(tmp_path):
    """the scripts end like the reference's: utils/plotting.py:8-16 saveResultDir + burgersutil.py:133-206 write
    results/<stamp>-<script>/{graph.pdf, graph.png, hp.json} (+ weights.npy: the flat vector)"""
    import glob
    import shutil
    hp = {"N_u": 64, "N_f": 2048, "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1], "tf_epochs": 20, "tf_lr": 0.03,
          "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 20, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10, "dtype": "f32"}
    res_dir = os.path.join(PKG, "1d-burgers", "results")
    before = set(glob.glob(os.path.join(res_dir, "*")))
    hp_file = tmp_path / "hp.json"
    hp_file.write_text(json.dumps(hp))
    env = dict(os.environ, MPLBACKEND="Agg")
    env.pop("PINN_NO_PLOT", None)
    res = subprocess.run([sys.executable, os.path.join(PKG, "1d-burgers", "inf_cont_burgers.py"), str(hp_file)],
                         cwd=PKG, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    new = sorted(set(glob.glob(os.path.join(res_dir, "*"))) - before)
    try:
        assert len(new) == 1 and new[0].endswith("-inf_cont_burgers"), new
        files = sorted(os.listdir(new[0]))
        for need in ("graph.pdf", "graph.png", "hp.json"):
            assert need in files and os.path.getsize(os.path.join(new[0], need)) > 0, files
        assert json.load(open(os.path.join(new[0], "hp.json"))) == hp
        import numpy as np
        w = np.load(os.path.join(new[0], "weights.npy"))
        assert w.shape == (3021,) and w.dtype == np.float64 and np.all(np.isfinite(w))
    finally:
        for d in new:
            shutil.rmtree(d, ignore_errors=True)


def test_inf_cont_schrodinger_runs(tmp_path):
    hp = {"N_0": 50, "N_b": 50, "N_f": 20000, "layers": [2, 100, 100, 100, 100, 2], "tf_epochs": 60,
          "tf_lr": 0.05, "tf_b1": 0.99, "tf_eps": 0.1, "nt_epochs": 0, "nt_lr": 1.2, "nt_ncorr": 50,
          "log_frequency": 10, "dtype": "f32"}
    out = run_script(os.path.join("1dcomplex-schrodinger", "inf_cont_schrodinger.py"), hp, tmp_path)
    rows, end = parse(out)
    assert len(rows) == 6 and end is not None and end[0] == 60
    assert all(r[2] == r[2] and r[2] < 10 for r in rows)                    # finite, sane losses
    assert rows[-1][2] < rows[0][2]


def test_checkpoint_round_trip_is_bit_exact(tmp_path, monkeypatch):
    """save_weights / load_weights / hp["init_weights"]: the float64 flat vector survives exactly and a
    resumed model evaluates to the same loss"""
    import numpy as np
    sys.path.insert(0, os.path.join(PKG, "utils"))
    sys.path.insert(0, os.path.join(PKG, "1d-burgers"))
    import importlib
    monkeypatch.setattr(sys, "argv", ["inf_cont_burgers.py"])     # the script reads argv[1] as hp.json
    burgersutil = importlib.import_module("burgersutil")
    inf = importlib.import_module("inf_cont_burgers")
    from logger import Logger
    g = json.load(open(golden("burgers_default_run.json")))
    hp = dict(g["hp"], N_f=2048, N_u=64, tf_epochs=5, nt_epochs=5, dtype="f64", log_frequency=1000)
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat"), hp["N_u"], hp["N_f"], noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    log = Logger(hp)
    log.set_error_fn(lambda: 0.0)                # as the reference: the script installs the error metric
    a = inf.BurgersInformedNN(hp, log, X_f, ub, lb, nu=0.01 / np.pi)
    a.fit(X_u, u)
    ck = a.save_weights(str(tmp_path / "w.npy"))
    la, _ = a.grad(X_u, u)
    b = inf.BurgersInformedNN(dict(hp, init_weights=ck), Logger(hp), X_f, ub, lb, nu=0.01 / np.pi)
    assert np.array_equal(a.get_weights(), b.get_weights())
    lb_, _ = b.grad(X_u, u)
    assert la == lb_


# ---- discrete-time scripts ----------------------------------------------------------------------------------
LAMBDAS = re.compile(r"l1 = (\S+)  l2 = (\S+)$")


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_inf_disc_burgers_adam_run_matches_reference_log(tmp_path, dtype):
    """200 Adam epochs of 1d-burgers/inf_disc_burgers.py (q = 500) against the reference script's printed log."""
    g = json.load(open(golden("burgers_disc_run.json")))
    hp = dict(g["hp"], dtype=dtype)
    out = run_script(os.path.join("1d-burgers", "inf_disc_burgers.py"), hp, tmp_path)
    rows, end = parse(out)
    ref, ref_end = parse("\n".join(g["lines"]))
    assert [(r[0], r[1]) for r in rows] == [(r[0], r[1]) for r in ref]
    tol = 1.5e-4 if dtype == "f64" else 2e-3
    for (kind, ep, loss), (_, _, loss_ref) in zip(rows, ref):
        assert abs(loss - loss_ref) <= tol * loss_ref, (kind, ep, loss, loss_ref)
    assert end is not None and end[0] == ref_end[0]
    assert abs(end[1] - g["final_error"]) <= (2e-4 if dtype == "f64" else 5e-3), (end, g["final_error"])


def test_inf_disc_burgers_lbfgs_converges(tmp_path):
    """Adam + L-BFGS with the true gradient (the reference's L-BFGS closure has none): the t_1 snapshot is
    predicted to a few percent after a short schedule."""
    hp = {"N_n": 250, "q": 100, "layers": [1, 50, 50, 50, 101], "tf_epochs": 100, "tf_lr": 0.001, "tf_b1": 0.9,
          "tf_eps": 1e-08, "nt_epochs": 400, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 50, "dtype": "f64"}
    out = run_script(os.path.join("1d-burgers", "inf_disc_burgers.py"), hp, tmp_path)
    rows, end = parse(out)
    assert any(r[0] == "nt_epoch" for r in rows)
    assert rows[-1][2] < 1e-2 * rows[0][2]
    assert end is not None and end[1] < 0.2, end


def test_ide_disc_burgers_run_matches_reference_log(tmp_path):
    """100 Adam + 60 L-BFGS of 1d-burgers/ide_disc_burgers.py, clean and noisy model, f64: printed losses and
    lambdas against the reference script's log."""
    g = json.load(open(golden("burgers_disc_ide_run.json")))
    hp = dict(g["hp"], dtype="f64")
    out = run_script(os.path.join("1d-burgers", "ide_disc_burgers.py"), hp, tmp_path)
    mine = [l for l in out.splitlines() if l.startswith(("tf_epoch", "nt_epoch"))]
    ref = [l for l in g["lines"] if l.startswith(("tf_epoch", "nt_epoch"))]
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        ma, mb = LINE.match(a), LINE.match(b)
        assert (ma.group(1), ma.group(2)) == (mb.group(1), mb.group(2))
        la, lb_ = float(ma.group(3)), float(mb.group(3))
        nt = ma.group(1) == "nt_epoch"
        assert abs(la - lb_) <= (2e-2 if nt else 1.5e-4) * lb_, (a, b)
        xa, xb = LAMBDAS.search(a), LAMBDAS.search(b)
        assert xa and xb
        for k in (1, 2):
            assert abs(float(xa.group(k)) - float(xb.group(k))) <= (2e-2 if nt else 2e-5), (a, b)
    vals = dict(re.findall(r"^(l1|l2|noisy l1|noisy l2):\s+(\S+)$", out, flags=re.M))
    assert abs(float(vals["l1"]) - g["lambda_1"]) < 0.05 and abs(float(vals["noisy l1"]) - g["lambda_1_noisy"]) < 0.05
