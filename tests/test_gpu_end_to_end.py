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
  2. the full default schedule (100 Adam + 200 L-BFGS) ends INSIDE [min, max] of the reference's own 25-member
     perturbation ensemble -- float64 against the (1 + k 2^-52) ensemble, float32 against the (1 + k 2^-23) one (a
     float32 implementation perturbs by that much at every step) -- and the trained field is as close to the
     reference's as the reference's own perturbed runs are;
  2b. the GPU engines run under the SAME 25 perturbations: their error distributions must pass a rank test
     (Mann-Whitney) against the reference's -- one sample could sit inside a wide range by luck, 25 cannot;
  3. BASELINE configs[0] (Adam x 2000, lr 0.03, reference value 4.3073e-01): float64 log prefix against the
     reference's printed log until its first loss spike, and the distribution of the GPU's 25 final errors against the
     reference's 25 (rank test) instead of a radius that accepted anything up to 1.0;
  4. for both L-BFGS kernels (mode 0 = the reference's operation order, mode 1 = compact form): the first iteration
     whose float64 loss departs from the reference's full-precision trace (burgers_default_trace.npz) by > 1e-6.
north_star's literal "final L2 error within 1e-3 of reference" is recorded per dtype (`err_absdiff_vs_k0`); at the end of
the full schedule it is a coin toss, not a property (f64 2.7e-3, then 9.9e-4, then 2.8e-3 as the L-BFGS dot products were
re-associated twice at the 1e-16 level; f32 2.3e-2): two runs of the reference itself differ by more (8 vs 4 torch threads:
0.26564 vs 0.26739) -- README.md states it.
"""
import json
import os
import sys

import numpy as np
import pytest

from conftest import PKG, ensemble_accepts, golden, same_distribution_p

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


def _band(dtype):
    return json.load(open(golden("burgers_band.json" if dtype == "f64" else "burgers_band_eps32.json")))


def _final_error(pinn, burgers_sets):
    X_star, u_star = _grid(burgers_sets)
    u = pinn.predict(X_star)[0][:, 0]
    return float(np.linalg.norm(u_star[:, 0] - u, 2) / np.linalg.norm(u_star[:, 0], 2)), u


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_default_schedule_final_error_inside_reference_ensemble(burgers_sets, monkeypatch, record, dtype):
    b = _band(dtype)
    errors = [v["final_error"] for v in b["runs"].values()]
    pinn = _run(dict(b["hp"], dtype=dtype), monkeypatch)
    err, u = _final_error(pinn, burgers_sets)
    assert abs(pinn.error_l2(*_grid(burgers_sets)) - err) <= 1e-13 * err          # the device-side metric agrees
    ok, lo, hi = ensemble_accepts(errors, err)
    # the reference's own trained field (k = 0 run): how far two members of the ensemble are apart, for the record
    from oracle import mlp
    X_star, _ = _grid(burgers_sets)
    w_ref = np.load(golden("burgers_default_run_w_final.npy"))
    lb, ub = np.array([-1.0, 0.0]), np.array([1.0, 0.99])
    u_ref = mlp.forward_value(mlp.unpack(w_ref, b["hp"]["layers"]), X_star, lb, ub)[:, 0]
    f = np.load(golden("burgers_band_fields.npz"))
    assert np.max(np.abs(u_ref - f["u_k0_full"])) < 1e-12          # the two fixtures describe the same run
    spread = max(float(np.sqrt(np.mean((f[k] - f["u_k+0"]) ** 2))) for k in f.files if k.startswith("u_k") and k not in ("u_k0_full", "u_k+0"))
    rms = float(np.sqrt(np.mean((u[::5] - f["u_k+0"]) ** 2)))
    ref0 = json.load(open(golden("burgers_band.json")))["reference_final_error"]
    record(dtype=dtype, err_gpu=err, ens_min=lo, ens_max=hi, members=len(errors), field_rms_vs_ref=rms,
           ensemble_field_rms_spread=spread, reference_error=ref0, err_absdiff_vs_k0=abs(err - ref0),
           within_1e_3_of_reference=bool(abs(err - ref0) <= 1e-3))
    assert ok, (dtype, err, lo, hi)
    assert rms <= 1.5 * spread, (rms, spread)      # as close to the reference's field as its own perturbed runs are


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_gpu_ensemble_ranks_like_the_reference_ensemble(burgers_sets, monkeypatch, record, dtype):
    """the engine under the same 25 perturbations of the initial kernels as the reference (hp["init_scale"] = 1 + k eps):
    its final errors and the reference's must be samples of one distribution (rank test), and every GPU member must
    stay inside a range no wider than the reference's own, stretched by its own width on either side (a blow-up or a
    collapse of a single member is a failure too)"""
    b = _band(dtype)
    ref = [b["runs"][str(k)]["final_error"] for k in b["k_ulp"]]
    mine, restarts = [], {}
    for k in b["k_ulp"]:
        pinn = _run(dict(b["hp"], dtype=dtype, init_scale=1.0 + k * b["eps"]), monkeypatch)
        mine.append(_final_error(pinn, burgers_sets)[0])
        if pinn.nt_restarts:
            restarts[int(k)] = pinn.nt_restarts
    p = same_distribution_p(mine, ref)
    lo, hi = min(ref), max(ref)
    wide_lo, wide_hi = lo - (hi - lo), hi + (hi - lo)
    outside = {int(k): e for k, e in zip(b["k_ulp"], mine) if not (wide_lo <= e <= wide_hi)}
    record(dtype=dtype, p_mannwhitney=p, gpu_min=min(mine), gpu_median=float(np.median(mine)), gpu_max=max(mine),
           ref_min=lo, ref_median=float(np.median(ref)), ref_max=hi, members=len(mine),
           gpu_inside_ref_range=int(sum(lo <= e <= hi for e in mine)), gpu_outside_3x_range=len(outside),
           gpu_diverged=int(sum(e > 1.0 for e in mine)), gpu_errors=" ".join("%.4f" % e for e in mine),
           nt_guard_restarts=json.dumps(restarts))
    assert p >= 1e-3, (dtype, p, sorted(mine), sorted(ref))
    # the same bound for both arithmetics (round 3 allowed float32 two members outside, one of them diverged to 3e6).
    # That divergence was traced (profiles/r04_diag_f32_k-10_shadow.txt): L-BFGS iteration 46 of member k = -10 produces a
    # curvature pair with y.s = 2.6e-5 against |y||s| = 3.0e-3, and the reference's update (no line search,
    # utils/custom_lbfgs.py:159-163) then steps 250x further than before -- the float64 kernels at the same iterates give
    # the same pair and the same step (cos 1.000000, loss 339.2 vs 339.6): the algorithm's hazard, which float32's
    # trajectory happened to meet.  float64 keeps the reference's behaviour; float32, the engine's own mode, runs with
    # the restart guard of NeuralNetwork.nt_optimization (hp["nt_guard"], default 1e3 there) and must not lose a member.
    assert sum(e > 1.0 for e in mine) == 0, (dtype, mine)
    assert not outside, (dtype, outside, lo, hi)
    if dtype == "f64":
        assert not restarts                      # the guard is off in the reference's arithmetic


def test_f32_member_lost_without_the_guard_is_kept_with_it(burgers_sets, monkeypatch, record):
    """member k = -10 of the float32-sized ensemble: with hp["nt_guard"] = 0 (the reference's behaviour) the run is lost at
    L-BFGS iteration 47 with the kernels as committed (final error 3e6); with the default guard the exploding chunk is
    discarded and the run ends inside the reference's range.  Which member meets the hazard depends on every rounding
    of the trajectory, so the first half is recorded, not asserted; the second half is asserted whenever it applies."""
    b = _band("f32")
    lo, hi = min(v["final_error"] for v in b["runs"].values()), max(v["final_error"] for v in b["runs"].values())
    hp = dict(b["hp"], dtype="f32", init_scale=1.0 - 10 * b["eps"])
    bare = _final_error(_run(dict(hp, nt_guard=0), monkeypatch), burgers_sets)[0]
    pinn = _run(hp, monkeypatch)
    kept = _final_error(pinn, burgers_sets)[0]
    record(err_without_guard=bare, err_with_guard=kept, restarts=json.dumps(pinn.nt_restarts), ref_min=lo, ref_max=hi)
    assert kept < 1.0 and lo - (hi - lo) <= kept <= hi + (hi - lo), (kept, lo, hi)
    if bare > 1.0:
        assert pinn.nt_restarts, "the guarded run of a member that diverges unguarded must have restarted"
    else:
        assert not pinn.nt_restarts and kept == bare          # a run that never explodes is untouched by the guard


def test_converged_schedule_ends_within_1e_3_of_the_reference(burgers_sets, monkeypatch, capsys, record):
    """north_star's literal criterion -- "final L2 error within 1e-3 of reference" -- where it is closest to well-posed: the
    default Adam phase followed by L-BFGS run LONG (5000 iterations; the reference's own stopping tests,
    utils/custom_lbfgs.py:200-215, never fire: tolFun = eps on sum|g|, tolX = 1e-19), by which time the error has fallen
    from 0.27 to 1-2.5e-3.  Fixture = the reference script over the shims, 15 members since round 5 (k = 0, +-1 ... +-7;
    tests/golden/make_band.py converged).  What the fixture itself says about the criterion: two of the reference's OWN 15
    float64 runs are lost on the way (6e16 / 3e26: its L-BFGS has no line search), and its 13 survivors span
    1.10e-3 ... 2.50e-3 -- the reference is not within 1e-3 of ITSELF (k = 0: 2.22e-3), and the errors compared are of the
    size of the bound.  So beside the absolute statement a RELATIVE one is asserted, and the field beside the scalar:
      * the engine has no more members lost or ending further than 1e-3 outside the range of the reference's survivors than
        the reference has lost members (reference: 2 of 15 lost for good; engine, round 5's kernel: 1 of 15 -- k = -7 ends at
        6.9e-3 after a late, half-healed explosion; round 6's kernels, other summation orders: 2 of 15 lost -- k = 4 and 7 with
        one build, k = 0 and -5 with the final one);
      * the engine's MEDIAN final error lies inside the reference survivors' [min, max] (relative: an implementation that
        converged to a different field quality would sit outside a range this narrow);
      * the engine's k = 0 field is as close to the reference's k = 0 field (RMS over the 25600-point grid) as the
        reference's own perturbed members are to it;
      * |engine - reference| at k = 0 <= 1e-3: recorded as `k0_absdiff` and asserted -- "met in absolute terms at the 1e-3
        scale of the errors themselves" (README.md), not more;
      * with the restart guard on no engine member is lost, and a member that never explodes is untouched by it."""
    b = json.load(open(golden("burgers_converged_band.json")))
    fields = np.load(golden("burgers_converged_fields.npz"))
    ref = {int(k): v["final_error"] for k, v in b["runs"].items()}
    ref_ok = {k: e for k, e in ref.items() if e < 1.0}
    assert len(ref) >= 15 and 0 in ref_ok and len(ref_ok) >= 10, ref
    lo, hi = min(ref_ok.values()), max(ref_ok.values())
    u_ref0 = fields["u_k0_full"]
    ref_rms = {k: float(np.sqrt(np.mean((fields["u_k%+d" % k].astype(np.float64) - u_ref0[::5]) ** 2))) for k in ref_ok if k != 0}
    mine, guarded, restarts, u0 = {}, {}, {}, None
    for k in b["k_ulp"]:
        hp = dict(b["hp"], dtype="f64", init_scale=1.0 + k * b["eps"])
        mine[k], u = _final_error(_run(hp, monkeypatch), burgers_sets)
        if k == 0:
            u0 = u
        pinn = _run(dict(hp, nt_guard=1e3), monkeypatch)
        guarded[k] = _final_error(pinn, burgers_sets)[0]
        restarts[k] = len(pinn.nt_restarts)
        capsys.readouterr()
    ok = {k: e for k, e in mine.items() if e < 1.0}
    median = float(np.median(list(ok.values())))
    rms0 = float(np.sqrt(np.mean((u0 - u_ref0) ** 2))) if 0 in ok else float("nan")
    record(reference=json.dumps(ref), engine=json.dumps(mine), engine_guarded=json.dumps(guarded), restarts=json.dumps(restarts),
           reference_lost=len(ref) - len(ref_ok), engine_lost=len(mine) - len(ok), ref_min=lo, ref_max=hi,
           engine_median=median, reference_median=float(np.median(list(ref_ok.values()))),
           k0_absdiff=abs(mine[0] - ref_ok[0]) if 0 in ok else float("nan"), k0_guarded_absdiff=abs(guarded[0] - ref_ok[0]),
           k0_field_rms=rms0, reference_field_rms_min=min(ref_rms.values()), reference_field_rms_max=max(ref_rms.values()))
    # members that are lost, or that end outside the survivors' range widened by 1e-3 (an explosion late in the schedule that
    # has not healed by iteration 5000 -- measured in round 5: engine k = -7 ends at 6.9e-3 unguarded, 9.1e-4 with the guard):
    # the engine has no more of them than the reference has lost members (2 of 15)
    astray = [k for k, e in mine.items() if not (lo - 1e-3 <= e <= hi + 1e-3)]
    assert len(astray) <= len(ref) - len(ref_ok), (astray, mine, lo, hi)
    # ... of which LOST for good (error >= 1, the reference's own two end at 6e16 / 3e26) no more than the reference loses either.
    # Which members go is not a property of an implementation: the reference's L-BFGS has no line search, and a change of the
    # summation order inside the gradient kernel moves the casualties (round 5's kernel: k = -7 astray at 6.9e-3, none lost;
    # round 6's: k = 4 and k = 7 lost, profiles/r06_parity_measured.jsonl).  A regression that loses MORE members than the
    # reference's own arithmetic does fails here.
    assert len(mine) - len(ok) <= len(ref) - len(ref_ok), (mine, ref)
    assert lo <= median <= hi, (median, lo, hi)
    if 0 in ok:
        assert abs(ok[0] - ref_ok[0]) <= 1e-3, (ok[0], ref_ok[0])
        assert rms0 <= max(ref_rms.values()), (rms0, ref_rms)
    else:
        # k = 0 itself is among the members this kernel's summation order loses (round 6: k = 0 and k = -5, as k = 1 and k = 2 are
        # the reference's): the absolute criterion is then read on the guarded run, which discards the exploding chunk
        assert abs(guarded[0] - ref_ok[0]) <= 1e-3, (guarded[0], ref_ok[0])
    assert all(lo - 1e-3 <= e <= hi + 1e-3 for e in guarded.values()), (guarded, lo, hi)
    assert all(guarded[k] == mine[k] for k in ok if restarts[k] == 0)       # a run that never explodes is untouched


def test_cfg1_adam2000_log_prefix_and_final_error(burgers_sets, monkeypatch, capsys, record):
    """BASELINE configs[0]: 8x20 MLP, N_f = 10000, Adam x 2000 at lr 0.03 (1d-burgers/inf_cont_burgers.py:35-37 with
    tf_epochs = 2000, nt_epochs = 0)."""
    import re
    b = json.load(open(golden("burgers_cfg1_band.json")))
    errors = [b["runs"][str(k)]["final_error"] for k in b["k_ulp"]]
    pinn = _run(dict(b["hp"], dtype="f64"), monkeypatch)
    out = capsys.readouterr().out
    line = re.compile(r"^tf_epoch =\s+(\d+)\s+elapsed = \S+ \(\S+\)  loss = (\S+)  ")
    mine = {int(m.group(1)): float(m.group(2)) for m in map(line.match, out.splitlines()) if m}
    ref = {int(m.group(1)): float(m.group(2)) for m in map(line.match, b["runs"]["0"]["lines"]) if m}
    assert sorted(mine) == sorted(ref) and len(ref) == 200
    # the two logs must agree (4 printed digits) up to the reference's first loss spike, and for >= 100 epochs
    first_bad = next((ep for ep in sorted(ref) if abs(mine[ep] - ref[ep]) > 2e-4 * ref[ep]), 2000)
    spike = next((ep for ep, nxt in zip(sorted(ref), sorted(ref)[1:]) if ep >= 100 and ref[nxt] > 1.5 * ref[ep]), 2000)
    err = _final_error(pinn, burgers_sets)[0]
    ok, lo, hi = ensemble_accepts(errors, err)
    # distribution overlap: the engine under the reference's 25 perturbations (2000 Adam steps each: 0.1 s of GPU time)
    gpu = [err]
    for k in b["k_ulp"][1:]:
        gpu.append(_final_error(_run(dict(b["hp"], dtype="f64", init_scale=1.0 + k * b.get("eps", 2.0 ** -52)), monkeypatch),
                                burgers_sets)[0])
    capsys.readouterr()
    p = same_distribution_p(gpu, errors)
    record(first_disagreeing_epoch=first_bad, reference_first_spike_epoch=spike, err_gpu=err, ens_min=lo, ens_max=hi,
           reference_error=b["reference_final_error"], p_mannwhitney=p, gpu_min=min(gpu),
           gpu_median=float(np.median(gpu)), gpu_max=max(gpu), ref_median=float(np.median(errors)))
    assert first_bad >= min(spike, 100), (first_bad, spike)
    assert ok, (err, lo, hi)
    assert p >= 1e-3, (p, sorted(gpu), sorted(errors))


@pytest.mark.parametrize("mode", [0, 1])
def test_first_lbfgs_iteration_departing_from_the_reference_trace(burgers_sets, record, mode):
    """the default schedule on the engine (float64) against the reference's full-precision trace of the same run
    (tests/golden/burgers_default_trace.npz: the reference script with Logger.log_train_epoch wrapped): every Adam
    loss to 1e-10, then the first L-BFGS iteration whose loss is off by more than 1e-6 (relative), for the kernel that
    follows the reference's operation order (mode 0) and for the compact form (mode 1, the default).  The schedule is
    roundoff-chaotic (a 1-ulp change of the initial weights moves the final error by 4e-2), so a departure is expected
    -- the assertion is that it comes late"""
    from pinn_native import Engine
    t = np.load(golden("burgers_default_trace.npz"))
    g = np.load(golden("burgers_eval.npz"))
    r = burgers_sets(100, 10000)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    eng = Engine([2] + [20] * 8 + [1], lb, ub, pde="burgers", dtype="f64")
    eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(0.01 / np.pi)
    eng.set_weights(g["w0"])
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    la = eng.adam_run(100)
    adam_dev = float(np.max(np.abs(la - t["adam_losses"]) / t["adam_losses"]))
    eng.lbfgs_set_mode(mode)
    eng.lbfgs_begin(200, 0.8, 50, float(np.finfo(float).eps))
    its, ll, done = eng.lbfgs_run(200)
    assert done == 1 and np.array_equal(its, t["lbfgs_iters"])
    dev = np.abs(ll - t["lbfgs_losses"]) / t["lbfgs_losses"]
    first = int(its[np.argmax(dev > 1e-6)]) if np.any(dev > 1e-6) else 0
    first9 = int(its[np.argmax(dev > 1e-9)]) if np.any(dev > 1e-9) else 0
    err = eng.error_l2(*_grid(burgers_sets))
    record(mode=mode, adam_loss_dev=adam_dev, first_iter_off_by_1e_9=first9, first_iter_off_by_1e_6=first,
           dev_at_50=float(dev[49]), dev_at_100=float(dev[99]), dev_at_150=float(dev[149]), dev_last=float(dev[-1]),
           final_error=err, reference_final_error=float(t["final_error"]))
    assert adam_dev < 1e-10
    assert first == 0 or first >= 60, (mode, first)
    eng.close()
