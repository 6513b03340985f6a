"""CPU: the oracle (numpy f64 restatement) against the golden fixtures that were produced by
running the reference's own Python over the test shims (tests/golden/make_golden.py).
This is what pins the oracle to the reference."""
import json

import numpy as np
import pytest

from conftest import golden, BURGERS_MAT
from oracle import init, mlp, optim, pde

NU = 0.01 / np.pi


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("tag,N_u,N_f", [("_small", 64, 2048), ("", 100, 10000)])
def test_burgers_eval(burgers_sets, tag, N_u, N_f):
    g = np.load(golden("burgers_eval%s.npz" % tag))
    hp = json.loads(str(g["hp"]))
    r = burgers_sets(N_u, N_f)
    X_star, X_u, u, X_f, ub, lb = r[5], r[7], r[8], r[9], r[10], r[11]
    assert np.array_equal(init.glorot_flat(hp["layers"]), g["w0"])       # canonical init
    assert mlp.n_params(hp["layers"]) == 3021
    loss, grad, ex = pde.burgers_loss_grad(g["w0"], hp["layers"], lb, ub, X_f, X_u, u, NU)
    assert abs(loss - float(g["loss"])) < 1e-14
    assert rel(grad, g["grad"]) < 1e-13
    assert np.max(np.abs(ex["f"][:64, 0] - g["f_first"])) < 1e-14
    assert abs(ex["mse_u"] - float(g["mse_u"])) < 1e-14
    up = mlp.forward_value(mlp.unpack(g["w0"], hp["layers"]), X_star, lb, ub)
    assert np.max(np.abs(up[::257, 0] - g["u_pred_stride"])) < 1e-14


def test_burgers_eval_1e6_blocks(burgers_sets):
    """BASELINE configs[4] (N_f = 10^6): the oracle, evaluated in chunks with the global denominator, against the
    reference's own closure on the 125 000-point blocks (tests/golden/make_golden.py gen_burgers_eval_1e6): the full
    set at the canonical weights, two single blocks at the perturbed ones, and the base class's data term"""
    g = np.load(golden("burgers_eval_1e6.npz"))
    hp = json.loads(str(g["hp"]))
    layers, N = hp["layers"], hp["N_f"]
    r = burgers_sets(hp["N_u"], N)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    assert np.array_equal(X_f[:4], g["X_f_first"]) and np.array_equal(X_f[-4:], g["X_f_last"])
    w0 = init.glorot_flat(layers)

    def residual_part(w, lo, hi):
        L, G = 0.0, 0.0
        for a in range(lo, hi, 62500):
            l, gr, _ = pde.burgers_loss_grad(w, layers, lb, ub, X_f[a:min(a + 62500, hi)], X_u, u, NU, n_f_total=N,
                                             with_data=False)
            L, G = L + l, G + gr
        return L, G

    def data_part(w):
        l, gr, _ = pde.burgers_loss_grad(w, layers, lb, ub, X_f[:0], X_u, u, NU, n_f_total=N, with_data=True)
        return l, gr
    # full set, canonical weights
    (Lr, Gr), (Lu, Gu) = residual_part(w0, 0, N), data_part(w0)
    assert abs(Lu - float(g["mse_u_w0"])) < 1e-15 and rel(Gu, g["grad_mse_u_w0"]) < 1e-13
    assert abs(Lr + Lu - float(g["loss_w0"])) < 1e-14
    assert rel(Gr + Gu, g["grad_w0"]) < 1e-13
    # single blocks at the perturbed weights: the reference's block model gives mse_u + mean_c(f^2)
    w1 = g["w1"]
    Lu, Gu = data_part(w1)
    assert abs(Lu - float(g["mse_u_w1"])) < 1e-15
    for k in (2, 7):
        lo, hi = (int(v) for v in g["bounds"][k])
        Lr, Gr = residual_part(w1, lo, hi)
        share = (hi - lo) / N
        assert abs(Lr / share + Lu - float(g["block_loss_w1"][k])) < 1e-14
        assert rel(Gr / share + Gu, g["block_grad_w1"][k]) < 1e-13


def test_burgers_known_values_from_survey(burgers_sets):
    """SURVEY.md Appendix C.2 figures (recorded independently of this repo's fixtures)."""
    g = np.load(golden("burgers_eval.npz"))
    assert abs(float(g["loss"]) - 0.26786867333002784) < 1e-15
    assert abs(np.abs(g["grad"]).sum() - 31.975903361883283) < 1e-11
    assert abs(g["grad"][0] - (-0.034737901273344848)) < 1e-15
    assert abs(g["grad"][3020] - 0.071065101852763018) < 1e-15
    assert abs(float(g["err0"]) - 0.91265619374884654) < 1e-14


def test_flat_layout_roundtrip():
    layers = [2, 7, 7, 7, 3]
    rs = np.random.RandomState(0)
    w = rs.standard_normal(mlp.n_params(layers))
    params = mlp.unpack(w, layers)
    assert [p[0].shape for p in params] == [(2, 7), (7, 7), (7, 7), (7, 3)]
    assert np.array_equal(mlp.pack(params), w)
    # W.flatten() row-major then b (utils/neuralnetwork.py:68-78)
    assert params[0][0][1, 3] == w[1 * 7 + 3] and params[0][1][2] == w[14 + 2]


def test_gradient_matches_finite_differences(burgers_sets):
    r = burgers_sets(64, 2048)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9][:256], r[10], r[11]
    layers = [2, 6, 6, 6, 1]
    rs = np.random.RandomState(3)
    w = 0.5 * rs.standard_normal(mlp.n_params(layers))
    _, g, _ = pde.burgers_loss_grad(w, layers, lb, ub, X_f, X_u, u, NU)
    for seed in range(3):
        v = np.random.RandomState(seed).standard_normal(w.size)
        v /= np.linalg.norm(v)
        h = 1e-6
        fp = pde.burgers_loss_grad(w + h * v, layers, lb, ub, X_f, X_u, u, NU)[0]
        fm = pde.burgers_loss_grad(w - h * v, layers, lb, ub, X_f, X_u, u, NU)[0]
        assert abs((fp - fm) / (2 * h) - g @ v) < 1e-7 * max(1.0, abs(g @ v))


def test_residual_of_exact_travelling_solution():
    """u = tanh-free analytic check: for u(x,t) = a x + b t the Burgers residual is b + a(ax+bt);
    a 1-layer 'network' with tiny weights is linear to O(w^3) -> Taylor channels are consistent."""
    layers = [2, 4, 1]
    lb, ub = np.array([-1.0, 0.0]), np.array([1.0, 1.0])
    rs = np.random.RandomState(5)
    w = 1e-3 * rs.standard_normal(mlp.n_params(layers))
    X = rs.uniform(-1, 1, size=(50, 2)) * [1, .5] + [0, .5]
    (h, p, q, r), _ = mlp.taylor_forward(mlp.unpack(w, layers), X, lb, ub)
    eps = 1e-5
    up = mlp.forward_value(mlp.unpack(w, layers), X + [eps, 0], lb, ub)
    um = mlp.forward_value(mlp.unpack(w, layers), X - [eps, 0], lb, ub)
    u0 = mlp.forward_value(mlp.unpack(w, layers), X, lb, ub)
    assert np.max(np.abs((up - um) / (2 * eps) - p)) < 1e-9
    assert np.max(np.abs((up - 2 * u0 + um) / eps ** 2 - r)) < 1e-5
    tp = mlp.forward_value(mlp.unpack(w, layers), X + [0, eps], lb, ub)
    tm = mlp.forward_value(mlp.unpack(w, layers), X - [0, eps], lb, ub)
    assert np.max(np.abs((tp - tm) / (2 * eps) - q)) < 1e-9


@pytest.mark.parametrize("tag,N_u,N_f", [("_small", 64, 2048)])
def test_adam_trajectory(burgers_sets, tag, N_u, N_f):
    g = np.load(golden("burgers_eval%s.npz" % tag))
    ga = np.load(golden("burgers_adam%s.npz" % tag))
    hp = json.loads(str(ga["hp"]))
    r = burgers_sets(N_u, N_f)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    w = g["w0"].copy()
    adam = optim.Adam(hp["tf_lr"], hp["tf_b1"], eps=hp["tf_eps"])
    losses = []
    for it in range(30):
        lo, gr, _ = pde.burgers_loss_grad(w, hp["layers"], lb, ub, X_f, X_u, u, NU)
        losses.append(lo)
        w = adam.step(w, gr)
        if it == 0:
            assert rel(w, ga["w_after_1"]) < 1e-13
    assert np.max(np.abs(np.array(losses) - ga["losses"]) / ga["losses"]) < 1e-10
    assert rel(w, ga["w_after_30"]) < 1e-10


@pytest.mark.parametrize("tag,N_u,N_f", [("_small", 64, 2048)])
def test_lbfgs_trajectory_and_last_iteration_quirk(burgers_sets, tag, N_u, N_f):
    g = np.load(golden("burgers_eval%s.npz" % tag))
    gl = np.load(golden("burgers_lbfgs%s.npz" % tag))
    hp = json.loads(str(gl["hp"]))
    r = burgers_sets(N_u, N_f)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    res = optim.lbfgs(lambda x: pde.burgers_loss_grad(x, hp["layers"], lb, ub, X_f, X_u, u, NU)[:2],
                      g["w0"], int(gl["max_iter"]), float(gl["lr"]), int(gl["n_corr"]))
    assert res["n_eval"] == int(gl["n_eval"]) == 25
    assert [l[0] for l in res["logs"]] == gl["log_iters"].tolist() == list(range(1, 25))
    assert np.max(np.abs(np.array(res["f_hist"]) - gl["f_hist"]) / gl["f_hist"]) < 1e-10
    assert rel(res["x"], gl["x_returned"]) < 1e-10
    assert rel(res["x_model"], gl["w_model"]) < 1e-10
    assert abs(res["final_loss"] - float(gl["final_loss_global"])) < 1e-12


def test_lbfgs_known_answer():
    k = json.load(open(golden("lbfgs_kat.json")))
    A = np.diag(np.arange(1.0, 7.0)) + 0.1 * np.ones((6, 6))
    b = np.arange(1.0, 7.0)
    args = []

    def opfunc(x):
        args.append(x.copy())
        return 0.5 * x @ A @ x - b @ x + 0.25 * np.sum(x ** 4), A @ x - b + x ** 3
    res = optim.lbfgs(opfunc, np.zeros(6), 8, 0.8, 3)
    assert np.allclose(res["f_hist"], k["f_hist"], rtol=0, atol=1e-13)
    assert np.allclose(res["x"], k["x_returned"], rtol=0, atol=1e-13)
    assert np.allclose(args[-1], k["last_opfunc_arg"], rtol=0, atol=1e-13)
    assert len(args) == k["n_opfunc_calls"] == 8 and res["n_eval"] == k["n_eval"]
    assert [l[0] for l in res["logs"]] == [l[0] for l in k["logs"]]
    assert optim.lbfgs(opfunc, np.zeros(6), 0, 0.8, 3) is None


@pytest.mark.parametrize("tag", ["_small", ""])
def test_burgers_identification(tag):
    """oracle vs the reference's own ide_cont_burgers.py (whitespace-repaired, run over the shims): loss, gradient
    incl. the lambda entries, residual, predict (u AND f at X_star, :169-172), Adam and L-BFGS trajectories"""
    from oracle import mlp, optim
    g = np.load(golden("burgers_ide_eval%s.npz" % tag))
    assert "reference ide_cont_burgers.py" in str(g["source"])
    X_u, u = g["X_u"], g["u"]
    lb, ub = np.array([-1.0, 0.0]), np.array([1.0, 0.99])
    layers = [2] + [20] * 8 + [1]
    loss, grad, ex = pde.burgers_ide_loss_grad(g["w0"], layers, lb, ub, X_u, u)
    assert abs(loss - float(g["loss"])) < 1e-14
    assert rel(grad, g["grad"]) < 1e-12
    assert abs(grad[-2] - g["grad"][-2]) < 1e-15 and abs(grad[-1] - g["grad"][-1]) < 1e-15
    assert np.max(np.abs(ex["f"][:64, 0] - g["f_first"])) < 1e-13
    # the second model of the script draws from the continuing streams: its data is NOT the seed-1234 sample
    import burgersutil
    np.random.seed(1234)
    r = burgersutil.prep_data(BURGERS_MAT, int(g["N_u"]), noise=0.0)
    r2 = burgersutil.prep_data(BURGERS_MAT, int(g["N_u"]), noise=0.01)
    assert not np.array_equal(r[7], X_u) and np.array_equal(r2[7], X_u) and np.array_equal(r2[8], u)
    # Adam: 10 steps, lr 1e-3 (ide_cont_burgers.py:36-39)
    adam = optim.Adam(1e-3, 0.9, 0.999, None)
    w = g["w0"].copy()
    losses = []
    for it in range(10):
        lo, gr, _ = pde.burgers_ide_loss_grad(w, layers, lb, ub, X_u, u)
        losses.append(lo)
        w = adam.step(w, gr)
        if it == 0:
            assert rel(w, g["adam_w_after_1"]) < 1e-13
    assert np.max(np.abs(np.array(losses) - g["adam_losses"]) / g["adam_losses"]) < 1e-10
    assert rel(w, g["adam_w_after_10"]) < 1e-10
    # L-BFGS: 25 iterations through the reference driver semantics
    args = []

    def opfunc(x):
        args.append(x.copy())
        lo, gr, _ = pde.burgers_ide_loss_grad(x, layers, lb, ub, X_u, u)
        return lo, gr
    res = optim.lbfgs(opfunc, g["w0"].copy(), int(g["lbfgs_max_iter"]), 0.8, int(g["lbfgs_n_corr"]))
    assert np.max(np.abs(np.array(res["f_hist"]) - g["lbfgs_f_hist"]) / np.abs(g["lbfgs_f_hist"])) < 1e-7
    assert rel(res["x"], g["lbfgs_x_returned"]) < 1e-6 and rel(args[-1], g["lbfgs_w_model"]) < 1e-6


def test_schrodinger_small(schrodinger_sets):
    g = np.load(golden("schrodinger_eval_small.npz"))
    hp = json.loads(str(g["hp"]))
    r = schrodinger_sets(50, 50, 1024)
    X_f, ub, lb, tb, x0, u0, v0, X0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17], r[18]
    X_lb = np.concatenate((0 * tb + lb[0], tb), 1)
    X_ub = np.concatenate((0 * tb + ub[0], tb), 1)
    uv0 = np.concatenate([u0, v0], 1)
    for mode, Xin in (("compat", np.concatenate([x0, x0], 1)), ("intent", X0)):
        loss, grad, ex = pde.schrodinger_loss_grad(g["w0"], hp["layers"], lb, ub, X_f, X_lb, X_ub,
                                                   Xin, uv0)
        assert abs(loss - float(g["loss_" + mode])) < 1e-13
        assert rel(grad, g["grad_" + mode]) < 1e-12
    assert np.max(np.abs(ex["f_u"][:64, 0] - g["f_u_first"])) < 1e-12
    # Adam (lr .05, b1 .99, eps .1) five steps, compat input as the script does
    w = g["w0"].copy()
    adam = optim.Adam(hp["tf_lr"], hp["tf_b1"], eps=hp["tf_eps"])
    for it in range(5):
        lo, gr, _ = pde.schrodinger_loss_grad(w, hp["layers"], lb, ub, X_f, X_lb, X_ub,
                                              np.concatenate([x0, x0], 1), uv0)
        assert abs(lo - g["adam_losses_compat"][it]) / g["adam_losses_compat"][it] < 1e-10
        w = adam.step(w, gr)
    assert rel(w, g["w_after_5"]) < 1e-10


def test_shard_sums_equal_full_batch(burgers_sets):
    """Data-parallel contract: un-normalised shard evaluations (global 1/N) add up to the
    full-batch loss and gradient."""
    g = np.load(golden("burgers_eval_small.npz"))
    r = burgers_sets(64, 2048)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    layers = [2] + [20] * 8 + [1]
    full = pde.burgers_loss_grad(g["w0"], layers, lb, ub, X_f, X_u, u, NU)
    tot_l, tot_g = 0.0, 0.0
    for k, sl in enumerate((slice(0, 700), slice(700, 2048))):
        lo, gr, _ = pde.burgers_loss_grad(g["w0"], layers, lb, ub, X_f[sl], X_u, u, NU,
                                          n_f_total=2048, with_data=(k == 0))
        tot_l, tot_g = tot_l + lo, tot_g + gr
    assert abs(tot_l - full[0]) < 1e-14 and rel(tot_g, full[1]) < 1e-13


# ---- discrete-time (IRK) Burgers models ---------------------------------------------------------------------
def _irk_sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("tag", ["_small", ""])
def test_disc_inference_eval(tag):
    """oracle/disc.py vs 1d-burgers/inf_disc_burgers.py run unmodified over the shims."""
    from oracle import disc
    g = np.load(golden("burgers_disc_eval%s.npz" % tag))
    hp = json.loads(str(g["hp"]))
    W, times = disc.irk_tables_like_reference(hp["q"])
    assert W.dtype == np.float32 and tuple(g["irk_shape"]) == W.shape and _irk_sha(W) == str(g["irk_sha"])
    sets = disc.inference_sets(g["x_0"], g["u_0"], g["x_1"], g["dt"], W)
    loss, grad, ex = disc.disc_loss_grad(g["w0"], hp["layers"], [-1.0], [1.0], sets, nu=NU)
    assert abs(loss - float(g["loss"])) <= 1e-13 * abs(loss)
    assert rel(grad, g["grad"]) < 1e-12
    # the closure the reference hands to L-BFGS computes its loss outside the tape (inf_disc_burgers.py:104-116):
    # same loss value, but not the gradient (recorded in the fixture so the divergence is visible)
    assert abs(float(g["closure_loss"]) - loss) <= 1e-13 * abs(loss) and float(g["closure_grad_maxdiff"]) > 1e-6
    params = mlp.unpack(g["w0"], hp["layers"])
    U0, _, _, _ = disc.stage_prediction(params, g["x_0"], np.array([-1.0]), np.array([1.0]), sets[0][2], 1.0, NU)
    assert np.max(np.abs(U0[:8, :8] - g["U0_first"])) < 1e-12
    # Adam trajectory (inf_disc_burgers.py hp: lr 1e-3, eps 1e-8)
    adam = optim.Adam(hp["tf_lr"], hp["tf_b1"], 0.999, hp["tf_eps"])
    w = g["w0"].copy()
    for it in range(10):
        lv, gr, _ = disc.disc_loss_grad(w, hp["layers"], [-1.0], [1.0], sets, nu=NU)
        assert abs(lv - g["adam_losses"][it]) <= 1e-11 * abs(lv)
        w = adam.step(w, gr)
    assert np.max(np.abs(w - g["w_after_10"])) < 1e-12


@pytest.mark.parametrize("tag", ["_small", ""])
def test_disc_identification_eval(tag):
    """oracle/disc.py vs 1d-burgers/ide_disc_burgers.py (second, noisy model of the script)."""
    from oracle import disc
    g = np.load(golden("burgers_disc_ide_eval%s.npz" % tag))
    q = int(g["q"])
    assert q == 81                                    # ceil(0.5 log(eps) / log(0.8)), burgersutil.py:90
    layers = [int(v) for v in g["layers"]]
    W, _ = disc.irk_tables_like_reference(q)
    assert _irk_sha(np.concatenate([W[:-1], W[-1:]])) == str(g["irk_sha"])
    sets = disc.identification_sets(g["x_0"], g["u_0"], g["x_1"], g["u_1"], g["dt"], W[:-1], W[-1:])
    loss, grad, _ = disc.disc_loss_grad(g["w0"], layers, [-1.0], [1.0], sets, identify=True)
    assert abs(loss - float(g["loss"])) <= 1e-13 * abs(loss)
    assert rel(grad, g["grad"]) < 1e-12
    assert np.max(np.abs(grad[-2:] - g["grad"][-2:])) <= 1e-10 * np.max(np.abs(g["grad"][-2:]))
    adam = optim.Adam(0.001, 0.9, 0.999, None)
    w = g["w0"].copy()
    for it in range(10):
        lv, gr, _ = disc.disc_loss_grad(w, layers, [-1.0], [1.0], sets, identify=True)
        assert abs(lv - g["adam_losses"][it]) <= 1e-11 * abs(lv)
        w = adam.step(w, gr)
    assert np.max(np.abs(w - g["w_after_10"])) < 1e-12


@pytest.mark.parametrize("q", [1, 2, 3, 8, 81, 500])
def test_gauss_legendre_butcher_conditions(q):
    """The Butcher files of the un-vendored PINNs submodule are restated, not copied: check the construction
    through the collocation / order / symplecticity conditions and against the product's independent generator."""
    from oracle import disc
    import irk
    A, b, c = disc.gauss_legendre_butcher(q)
    assert abs(b.sum() - 1.0) < 1e-13 and np.all(b > 0) and np.all(np.diff(c) > 0)
    for k in range(1, min(q, 10) + 1):
        assert np.max(np.abs(A @ c ** (k - 1) - c ** k / k)) < 1e-13          # C(q)
        assert abs(b @ c ** (k - 1) - 1.0 / k) < 1e-13                        # B(q..)
    assert np.max(np.abs(b[:, None] * A + (b[:, None] * A).T - np.outer(b, b))) < 1e-15   # symplectic
    if q == 2:
        assert abs(A[0, 1] - (0.25 - np.sqrt(3.0) / 6.0)) < 1e-15
    if q == 3:        # textbook known-answer: the 3-stage, order-6 Gauss method (Hairer & Wanner, Solving ODEs II, IV.5)
        r15 = np.sqrt(15.0)
        A3 = np.array([[5 / 36, 2 / 9 - r15 / 15, 5 / 36 - r15 / 30],
                       [5 / 36 + r15 / 24, 2 / 9, 5 / 36 - r15 / 24],
                       [5 / 36 + r15 / 30, 2 / 9 + r15 / 15, 5 / 36]])
        assert np.max(np.abs(A - A3)) < 1e-15
        assert np.max(np.abs(b - np.array([5 / 18, 4 / 9, 5 / 18]))) < 1e-15
        assert np.max(np.abs(c - np.array([0.5 - r15 / 10, 0.5, 0.5 + r15 / 10]))) < 1e-15
    A2, b2, c2 = irk.gauss_legendre_butcher(q)
    assert np.max(np.abs(A - A2)) < 1e-13 and np.max(np.abs(b - b2)) < 1e-13 and np.max(np.abs(c - c2)) < 1e-13


def test_disc_prep_data_matches_reference_draws():
    """Product burgersutil.prep_data (discrete branches) reproduces the reference's RNG draw order."""
    import burgersutil
    g = np.load(golden("burgers_disc_eval.npz"))
    np.random.seed(1234)
    r = burgersutil.prep_data(BURGERS_MAT, N_n=250, q=500, lb=np.array([-1.0]), ub=np.array([1.0]), noise=0.0,
                              idx_t_0=10, idx_t_1=90)
    assert len(r) == 11 and np.array_equal(r[4], g["x_0"]) and np.array_equal(r[5], g["u_0"])
    assert np.array_equal(r[6], g["x_1"]) and r[9].shape == (501, 500) and r[9].dtype == np.float32
    assert abs(float(r[2][0]) - float(g["dt"][0])) == 0.0
    g = np.load(golden("burgers_disc_ide_eval.npz"))
    np.random.seed(1234)
    kw = dict(N_0=199, N_1=201, lb=np.array([-1.0]), ub=np.array([1.0]), idx_t_0=10, idx_t_1=90)
    burgersutil.prep_data(BURGERS_MAT, noise=0.0, **kw)          # the script's first (clean) call
    r = burgersutil.prep_data(BURGERS_MAT, noise=0.01, **kw)
    assert len(r) == 11 and r[7] == 81 and r[9].shape == (81, 81) and r[10].shape == (1, 81)
    for i, key in enumerate(("x_0", "u_0", "x_1", "u_1")):
        assert np.array_equal(r[i], g[key])


# ---- device Latin-hypercube generator: the numpy restatement the GPU test compares against ------------------
def test_lhs_restatement_philox_known_answers_and_bijection():
    from oracle import lhs
    z = np.array([0], dtype=np.uint64)
    f = np.array([0xFFFFFFFF], dtype=np.uint64)
    # Random123 known-answer vectors for philox4x32-10
    assert [int(v[0]) for v in lhs.philox4x32_10(z, z, z, z, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(v[0]) for v in lhs.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    for n in (1, 2, 3, 17, 1000, 4097):
        i = np.arange(n, dtype=np.uint64)
        p = lhs.permute(i, n, 0x9ABCDEF0, 0x12345678)
        assert np.array_equal(np.sort(p), i)
    X, _ = lhs.lhs_points(2048, 99, [-1.0, 0.0], [1.0, 0.99])
    assert X[:, 0].min() >= -1.0 and X[:, 0].max() < 1.0 and X[:, 1].min() >= 0.0 and X[:, 1].max() < 0.99
    for d, (lo, hi) in enumerate(((-1.0, 1.0), (0.0, 0.99))):
        strata = np.floor((X[:, d] - lo) / (hi - lo) * 2048).astype(int)
        assert np.array_equal(np.sort(strata), np.arange(2048))
    # a different seed is a different design
    assert not np.array_equal(X, lhs.lhs_points(2048, 100, [-1.0, 0.0], [1.0, 0.99])[0])
