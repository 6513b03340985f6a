"""End-to-end parity of the Burgers inference path against evidence generated from the reference itself
(tests/golden/make_band.py = the unmodified reference script over the test shims; pytest -m gpu).

north_star: "trained u(x,t) field within a stated tolerance, final L2 within 1e-3 of reference"
(inf_cont_burgers.py:114-123, logger.py:56-60).  The reference's schedules are roundoff-chaotic: its OWN final error
moves by up to 4e-2 when the initial weights move by one ulp (burgers_band.json).  What is asserted, with every
number read from a fixture the reference produced:

  1. field level, float64, where float64 implementations still track each other: after the 100 Adam epochs, and
     after 50 / 100 further L-BFGS iterations, the GPU-trained weights, the field u(X_star) on the 25600-point grid
     and the error agree with the reference's to the tolerances in PREFIX_TOL (the 1e-3 of north_star holds
     there, with margin);
  2. the full default schedule (100 Adam + 200 L-BFGS) in float64 AND float32 ends inside the reference's own
     ulp-perturbation ensemble (rule: conftest.ensemble_accepts), and the float64 field is compared with the
     reference's trained field burgers_default_run_w_final.npy -> predict(X_star);
  3. BASELINE configs[0] (Adam x 2000, lr 0.03, reference value 4.3073e-01): float64 log prefix against the
     reference's printed log until its first loss spike, final error inside that configuration's ensemble.
"""
import json
import os
import sys

import numpy as np
import pytest

from conftest import PKG, ensemble_accepts, golden

pytestmark = pytest.mark.gpu

# (relative weight deviation / max-abs, max |u_gpu - u_ref| on the grid, |error_gpu - error_ref|), float64
# measured on MI355X (profiles/r02_parity_measured.jsonl): a100 3.8e-14 / 6.8e-15 / 7e-16; a100_l50 3.8e-10 / 6.0e-9 /
# 1.8e-12; a100_l100 6.6e-6 / 9.1e-5 / 4.1e-6 -- the tolerances leave one to two orders for another GPU / compiler
PREFIX_TOL = {"a100": (1e-11, 1e-11, 1e-12), "a100_l50": (1e-7, 1e-6, 1e-9), "a100_l100": (2e-4, 1e-3, 1e-4)}


def _run(hp, monkeypatch):
    monkeypatch.setenv("PINN_NO_PLOT", "1")
    monkeypatch.setattr(sys, "argv", ["inf_cont_burgers.py"])
    for p in (os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    import neuralnetwork
    inf = importlib.import_module("inf_cont_burgers")
    np.random.seed(1234)
    neuralnetwork.set_seed(1234)
    return inf.run(hp)


def _grid(burgers_sets):
    r = burgers_sets(100, 10000)
    return r[5], r[6]


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize("tag,nt", [("a100", 0), ("a100_l50", 50), ("a100_l100", 100)])
def test_f64_trained_field_tracks_reference_on_schedule_prefixes(burgers_sets, monkeypatch, capsys, record, tag, nt):
    g = np.load(golden("burgers_prefix.npz"))
    hp = dict(json.load(open(golden("burgers_default_run.json")))["hp"], tf_epochs=100, nt_epochs=nt, dtype="f64")
    pinn = _run(hp, monkeypatch)
    X_star, u_star = _grid(burgers_sets)
    u = pinn.predict(X_star)[0][:, 0]
    err = float(np.linalg.norm(u_star[:, 0] - u, 2) / np.linalg.norm(u_star[:, 0], 2))
    dw, du, de = rel(pinn.get_weights(), g["w_" + tag]), float(np.max(np.abs(u - g["u_" + tag]))), abs(err - float(g["err_" + tag]))
    record(tag=tag, w_rel=dw, u_maxabs=du, err_gpu=err, err_ref=float(g["err_" + tag]), err_absdiff=de)
    tw, tu, te = PREFIX_TOL[tag]
    assert dw < tw and du < tu and de < te, (tag, dw, du, de)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_default_schedule_final_error_inside_reference_ensemble(burgers_sets, monkeypatch, record, dtype):
    b = json.load(open(golden("burgers_band.json")))
    errors = [v["final_error"] for v in b["runs"].values()]
    pinn = _run(dict(b["hp"], dtype=dtype), monkeypatch)
    X_star, u_star = _grid(burgers_sets)
    u = pinn.predict(X_star)[0][:, 0]
    err = float(np.linalg.norm(u_star[:, 0] - u, 2) / np.linalg.norm(u_star[:, 0], 2))
    ok, med, radius = ensemble_accepts(errors, err)
    # the reference's own trained field (k = 0 run): how far two members of the ensemble are apart, for the record
    from oracle import mlp
    w_ref = np.load(golden("burgers_default_run_w_final.npy"))
    lb, ub = np.array([-1.0, 0.0]), np.array([1.0, 0.99])
    u_ref = mlp.forward_value(mlp.unpack(w_ref, b["hp"]["layers"]), X_star, lb, ub)[:, 0]
    f = np.load(golden("burgers_band_fields.npz"))
    assert np.max(np.abs(u_ref - f["u_k0_full"])) < 1e-12          # the two fixtures describe the same run
    spread = max(float(np.sqrt(np.mean((f[k] - f["u_k+0"]) ** 2))) for k in f.files if k.startswith("u_k") and k not in ("u_k0_full", "u_k+0"))
    rms = float(np.sqrt(np.mean((u[::5] - f["u_k+0"]) ** 2)))
    record(dtype=dtype, err_gpu=err, ens_median=med, ens_radius=radius, ens_min=min(errors), ens_max=max(errors),
           field_rms_vs_ref=rms, ensemble_field_rms_spread=spread, reference_error=b["reference_final_error"])
    assert ok, (dtype, err, med, radius, sorted(errors))
    assert rms <= 1.5 * spread, (rms, spread)      # as close to the reference's field as its own perturbed runs are


def test_cfg1_adam2000_log_prefix_and_final_error(burgers_sets, monkeypatch, capsys, record):
    """BASELINE configs[0]: 8x20 MLP, N_f = 10000, Adam x 2000 at lr 0.03 (1d-burgers/inf_cont_burgers.py:35-37 with
    tf_epochs = 2000, nt_epochs = 0)."""
    import re
    b = json.load(open(golden("burgers_cfg1_band.json")))
    errors = [v["final_error"] for v in b["runs"].values()]
    pinn = _run(dict(b["hp"], dtype="f64"), monkeypatch)
    out = capsys.readouterr().out
    line = re.compile(r"^tf_epoch =\s+(\d+)\s+elapsed = \S+ \(\S+\)  loss = (\S+)  ")
    mine = {int(m.group(1)): float(m.group(2)) for m in map(line.match, out.splitlines()) if m}
    ref = {int(m.group(1)): float(m.group(2)) for m in map(line.match, b["runs"]["0"]["lines"]) if m}
    assert sorted(mine) == sorted(ref) and len(ref) == 200
    # the two logs must agree (4 printed digits) up to the reference's first loss spike, and for >= 100 epochs
    first_bad = next((ep for ep in sorted(ref) if abs(mine[ep] - ref[ep]) > 2e-4 * ref[ep]), 2000)
    spike = next((ep for ep, nxt in zip(sorted(ref), sorted(ref)[1:]) if ep >= 100 and ref[nxt] > 1.5 * ref[ep]), 2000)
    X_star, u_star = _grid(burgers_sets)
    u = pinn.predict(X_star)[0][:, 0]
    err = float(np.linalg.norm(u_star[:, 0] - u, 2) / np.linalg.norm(u_star[:, 0], 2))
    ok, med, radius = ensemble_accepts(errors, err)
    record(first_disagreeing_epoch=first_bad, reference_first_spike_epoch=spike, err_gpu=err, ens_median=med,
           ens_radius=radius, reference_error=b["reference_final_error"])
    assert first_bad >= min(spike, 100), (first_bad, spike)
    assert ok, (err, med, radius, sorted(errors))
