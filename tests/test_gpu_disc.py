"""GPU: discrete-time (IRK) Burgers models -- the four-launch MFMA evaluation of csrc/kernels_disc.h against the
oracle (oracle/disc.py, itself pinned to the reference scripts run over the shims: tests/golden/burgers_disc_*.npz).
Tolerances: f64 1e-11 (relative to the largest gradient entry), f32 2e-4 gradient / 2e-5 loss -- the losses are
sums over up to 250 x 501 squared residuals of magnitude 1e4..1e5, so f32 is looser than on the continuous models."""
import json

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

NU = 0.01 / np.pi
TOL = {"f64": (1e-12, 1e-11), "f32": (2e-5, 2e-4)}


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def inference_case(tag):
    from oracle import disc
    z = np.load(golden("burgers_disc_eval%s.npz" % tag))
    hp = json.loads(str(z["hp"]))
    W, _ = disc.irk_tables_like_reference(hp["q"])
    sets = disc.inference_sets(z["x_0"], z["u_0"], z["x_1"], z["dt"], W)
    return z, hp, sets


def make_engine(layers, sets, dtype, identify=False):
    import pinn_native
    eng = pinn_native.Engine(layers, [-1.0], [1.0], pde="burgers_disc_ide" if identify else "burgers_disc",
                             dtype=dtype)
    for s, (x, t, M) in enumerate(sets):
        eng.disc_set_stage(s, x, t, M)
    if not identify:
        eng.set_pde_params(NU)
    return eng


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag", ["_small", ""])
def test_inference_loss_grad_vs_golden_and_oracle(tag, dtype):
    from oracle import disc
    z, hp, sets = inference_case(tag)
    eng = make_engine(hp["layers"], sets, dtype)
    assert eng.n_params == z["w0"].size
    eng.set_weights(z["w0"])
    loss, grad, terms = eng.loss_grad()
    lo, go, ex = disc.disc_loss_grad(z["w0"], hp["layers"], [-1.0], [1.0], sets, nu=NU)
    tl, tg = TOL[dtype]
    assert abs(loss - float(z["loss"])) <= tl * abs(float(z["loss"]))
    assert rel(grad, z["grad"]) <= tg
    assert abs(loss - lo) <= tl * abs(lo) and rel(grad, go) <= tg
    assert abs(terms[0] - ex["sse"][0]) <= tl * abs(lo) and abs(terms[1] - ex["sse"][1]) <= tl * abs(lo)
    # predict = last output column (inf_disc_burgers.py:126-129)
    xs = np.linspace(-1.0, 1.0, 37)
    U = eng.predict(xs[:, None])
    ref = disc.predict_last(z["w0"], hp["layers"], [-1.0], [1.0], xs[:, None])
    assert np.max(np.abs(U[:, -1] - ref)) <= (1e-12 if dtype == "f64" else 2e-5)
    # bit-reproducible
    loss2, grad2, _ = eng.loss_grad()
    assert loss2 == loss and np.array_equal(grad, grad2)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag", ["_small", ""])
def test_identification_loss_grad_vs_golden_and_oracle(tag, dtype):
    from oracle import disc
    z = np.load(golden("burgers_disc_ide_eval%s.npz" % tag))
    q = int(z["q"])
    layers = [int(v) for v in z["layers"]]
    W, _ = disc.irk_tables_like_reference(q)
    sets = disc.identification_sets(z["x_0"], z["u_0"], z["x_1"], z["u_1"], z["dt"], W[:-1], W[-1:])
    eng = make_engine(layers, sets, dtype, identify=True)
    assert eng.n_params == z["w0"].size
    eng.set_weights(z["w0"])
    loss, grad, _ = eng.loss_grad()
    tl, tg = TOL[dtype]
    assert abs(loss - float(z["loss"])) <= tl * abs(float(z["loss"]))
    assert rel(grad, z["grad"]) <= tg
    # the two lambda entries against their own scale (they are 1e3 below the largest weight gradient)
    assert np.max(np.abs(grad[-2:] - z["grad"][-2:])) <= (1e-9 if dtype == "f64" else 5e-3) * np.max(np.abs(z["grad"][-2:]))
    # U_0_model / U_1_model at the script's prediction points (ide_disc_burgers.py:188-193)
    xs = np.linspace(-1.0, 1.0, 8)
    for s, key in ((0, "U0_first"), (1, "U1_first")):
        P = eng.disc_predict(s, xs)
        params_w = z["w0"]
        from oracle import mlp
        pred, _, _, _ = disc.stage_prediction(mlp.unpack(params_w[:-2], layers), xs[:, None], np.array([-1.0]),
                                              np.array([1.0]), sets[s][2], params_w[-2], np.exp(params_w[-1]))
        assert np.max(np.abs(P - pred)) <= (1e-10 if dtype == "f64" else 1e-3) * max(1.0, np.max(np.abs(pred)))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_adam_trajectory_matches_reference_run(dtype):
    z, hp, sets = inference_case("_small")
    eng = make_engine(hp["layers"], sets, dtype)
    eng.set_weights(z["w0"])
    eng.adam_init(hp["tf_lr"], hp["tf_b1"], 0.999, hp["tf_eps"])
    losses = eng.adam_run(10)
    tol = 1e-10 if dtype == "f64" else 1e-4
    assert np.max(np.abs(losses - z["adam_losses"]) / z["adam_losses"]) <= tol
    w = eng.get_weights()
    assert np.max(np.abs(w - z["w_after_10"])) <= (1e-10 if dtype == "f64" else 2e-4)


def test_ragged_groups_and_single_set():
    """point counts that are not multiples of 16, one set only, widths that are not multiples of 16"""
    from oracle import disc, init
    rs = np.random.RandomState(3)
    q, layers = 5, [1, 23, 23, 7]
    A, b, c = disc.gauss_legendre_butcher(q)
    M = 0.3 * np.vstack([A, b[None, :], 0.5 * b[None, :]])        # [n_out = 7, q = 5]
    for n0, n1 in ((1, 0), (17, 3), (33, 16)):
        x0, t0 = rs.uniform(-1, 1, n0), rs.standard_normal(n0)
        x1, t1 = rs.uniform(-1, 1, n1), rs.standard_normal(n1)
        sets = [(x0[:, None], t0[:, None], M), (x1[:, None], t1[:, None], None)]
        w = 0.5 * rs.standard_normal(sum(a * b + b for a, b in zip(layers[:-1], layers[1:])))
        eng = make_engine(layers, sets, "f64")
        eng.set_weights(w)
        loss, grad, _ = eng.loss_grad()
        lo, go, _ = disc.disc_loss_grad(w, layers, [-1.0], [1.0], [s for s in sets if len(s[0])], nu=NU)
        assert abs(loss - lo) <= 1e-12 * abs(lo) and rel(grad, go) <= 1e-11


def test_wide_hidden_layers_f32():
    """hidden width 100 (> 64): the NT = 8 instantiation, float32 only; float64 refuses it with a message"""
    import pinn_native
    from oracle import disc
    rs = np.random.RandomState(5)
    q, layers = 12, [1, 100, 100, 13]
    A, b, c = disc.gauss_legendre_butcher(q)
    M = 0.2 * np.vstack([A, b[None, :]])
    x0, t0 = rs.uniform(-1, 1, 40), rs.standard_normal(40)
    sets = [(x0[:, None], t0[:, None], M), (np.array([[-1.0], [1.0]]), np.zeros((2, 1)), None)]
    w = 0.3 * rs.standard_normal(sum(a * b + b for a, b in zip(layers[:-1], layers[1:])))
    eng = make_engine(layers, sets, "f32")
    eng.set_weights(w)
    loss, grad, _ = eng.loss_grad()
    lo, go, _ = disc.disc_loss_grad(w, layers, [-1.0], [1.0], sets, nu=NU)
    assert abs(loss - lo) <= 2e-5 * abs(lo) and rel(grad, go) <= 2e-4
    with pytest.raises(pinn_native.PinnNativeError, match="hidden width"):
        pinn_native.Engine(layers, [-1.0], [1.0], pde="burgers_disc", dtype="f64")


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_many_groups(dtype):
    """a few thousand points (hundreds of 16-point groups, two sets): more workgroups than compute units"""
    from oracle import disc
    rs = np.random.RandomState(11)
    q, layers = 20, [1, 50, 50, 50, 21]
    A, b, c = disc.gauss_legendre_butcher(q)
    M = 0.05 * np.vstack([A, b[None, :]])
    n0, n1 = 4100, 777
    x0, x1 = rs.uniform(-1, 1, (n0, 1)), rs.uniform(-1, 1, (n1, 1))
    sets = [(x0, -np.sin(np.pi * x0), M), (x1, 0.1 * rs.standard_normal((n1, 1)), -0.5 * M)]
    w = 0.4 * rs.standard_normal(sum(a * b + b for a, b in zip(layers[:-1], layers[1:])))
    eng = make_engine(layers, sets, dtype)
    eng.set_weights(w)
    loss, grad, terms = eng.loss_grad()
    lo, go, ex = disc.disc_loss_grad(w, layers, [-1.0], [1.0], sets, nu=NU)
    tl, tg = TOL[dtype]
    assert abs(loss - lo) <= tl * abs(lo) and rel(grad, go) <= tg
    assert abs(terms[0] - ex["sse"][0]) <= tl * abs(lo) and abs(terms[1] - ex["sse"][1]) <= tl * abs(lo)
