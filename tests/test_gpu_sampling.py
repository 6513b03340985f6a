"""GPU: device-side Latin hypercube collocation sets (csrc/kernels_sampling.h) against the numpy restatement of the
same integer pipeline (oracle/lhs.py): bit-exact in f64, exact float32 rounding in f32, the stratification property,
shard consistency across "ranks", in-place re-draws, and the loss/gradient on a device-drawn set."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAYERS = [2, 20, 20, 20, 20, 20, 20, 20, 20, 1]
LB, UB = np.array([-1.0, 0.0]), np.array([1.0, 0.99])
NU = 0.01 / np.pi


def engine(dtype):
    import pinn_native
    from oracle import init
    eng = pinn_native.Engine(LAYERS, LB, UB, pde="burgers", dtype=dtype)
    rs = np.random.RandomState(0)
    Xu = np.column_stack([rs.uniform(-1, 1, 50), np.zeros(50)])
    eng.set_data(Xu, np.sin(np.pi * Xu[:, 0:1]))
    eng.set_pde_params(NU)
    eng.set_weights(init.glorot_flat(LAYERS))
    return eng, Xu


@pytest.mark.parametrize("n", [1, 1000, 10000, 65537])
def test_device_lhs_matches_restatement_bit_for_bit(n):
    from oracle import lhs
    eng, _ = engine("f64")
    eng.lhs_collocation(n, seed=0x1234ABCD5678)
    X = eng.get_collocation()
    ref, (p0, p1) = lhs.lhs_points(n, 0x1234ABCD5678, LB, UB)
    assert np.array_equal(X, ref)
    # one point per stratum in each dimension
    assert np.array_equal(np.sort(p0), np.arange(n, dtype=np.uint64))
    assert np.array_equal(np.sort(p1), np.arange(n, dtype=np.uint64))
    for d in range(2):
        strata = np.floor((X[:, d] - LB[d]) / (UB[d] - LB[d]) * n).astype(np.int64)
        assert np.array_equal(np.sort(np.clip(strata, 0, n - 1)), np.arange(n))
    eng32, _ = engine("f32")
    eng32.lhs_collocation(n, seed=0x1234ABCD5678)
    assert np.array_equal(eng32.get_collocation(), ref.astype(np.float32).astype(np.float64))


def test_shards_of_one_design_and_redraw_in_place():
    from oracle import lhs
    n = 5000
    full, _ = lhs.lhs_points(n, 77, LB, UB)
    eng, _ = engine("f64")
    eng.lhs_collocation(n, seed=77, first=0, count=1800)
    a = eng.get_collocation()
    eng.lhs_collocation(n, seed=77, first=1800, count=3200)
    b = eng.get_collocation()
    assert np.array_equal(np.vstack([a, b]), full)
    eng.lhs_collocation(n, seed=78, first=1800, count=3200)          # same shape: re-drawn in place
    c = eng.get_collocation()
    assert np.array_equal(c, lhs.lhs_points(n, 78, LB, UB, first=1800, count=3200)[0]) and not np.array_equal(b, c)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_loss_grad_on_a_device_drawn_set(dtype):
    from oracle import pde, init, lhs
    n = 4096
    eng, Xu = engine(dtype)
    eng.lhs_collocation(n, seed=5)
    loss, grad, _ = eng.loss_grad()
    Xf = lhs.lhs_points(n, 5, LB, UB)[0]
    if dtype == "f32":
        Xf = Xf.astype(np.float32).astype(np.float64)
    lo, go, _ = pde.burgers_loss_grad(init.glorot_flat(LAYERS), LAYERS, LB, UB, Xf, Xu, np.sin(np.pi * Xu[:, 0:1]), NU)
    tl, tg = (1e-12, 1e-11) if dtype == "f64" else (1e-5, 2e-5)
    assert abs(loss - lo) <= tl * max(1.0, abs(lo))
    assert np.max(np.abs(grad - go)) <= tg * np.max(np.abs(go))
    # a shard normalises by the design size: two shards add up to the whole
    eng.lhs_collocation(n, seed=5, first=0, count=1000)
    l1, g1, t1 = eng.loss_grad()
    eng.lhs_collocation(n, seed=5, first=1000, count=3096)
    l2, g2, t2 = eng.loss_grad()
    assert abs((t1[0] + t2[0]) - (loss - t1[1])) <= (1e-12 if dtype == "f64" else 1e-5)


def test_resample_every_in_the_training_loop(tmp_path):
    import json
    import os
    import subprocess
    import sys
    from conftest import PKG
    hp = {"N_u": 100, "N_f": 4000, "layers": LAYERS, "tf_epochs": 60, "tf_lr": 0.01, "tf_b1": 0.9, "tf_eps": None,
          "nt_epochs": 0, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10, "dtype": "f32", "resample_every": 7}
    f = tmp_path / "hp.json"
    f.write_text(json.dumps(hp))
    res = subprocess.run([sys.executable, os.path.join(PKG, "1d-burgers", "inf_cont_burgers.py"), str(f)], cwd=PKG,
                         env=dict(os.environ, PINN_NO_PLOT="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    losses = [float(l.split("loss = ")[1].split()[0]) for l in res.stdout.splitlines() if l.startswith("tf_epoch")]
    assert len(losses) == 6 and losses[-1] < losses[0]
