"""GPU parity at BASELINE configs[4]'s size (pytest -m gpu): N_f = 10^6 collocation points, the size bench.py's cfg5_leg times
and the regime no smaller test reaches -- 61 tiles per workgroup, the LDS gradient accumulators of k_fused20d / k_fused20m
carried over 61 tiles (csrc/kernels_fused20d.h, kernels_fused20m.h).

Checked against
  * tests/golden/burgers_eval_1e6.npz: the reference's own get_loss_and_flat_grad (1d-burgers/inf_cont_burgers.py:59-90 on
    burgersutil.py:122's Latin hypercube) evaluated block by block by tests/golden/make_golden.py -- the full set and every
    125 000-point shard of an 8-rank launch, at the canonical and a perturbed weight vector;
  * oracle.pde.burgers_loss_grad evaluated in chunks (n_f_total = 10^6, with_data = False) + the data term once;
  * oracle.optim.Adam over that chunked evaluation for five steps.
Tolerances: float64 loss 1e-12 / gradient 1e-11 (relative to the gradient's max-abs); float32 1e-5 / 2e-5.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

NU = 0.01 / np.pi
LAYERS = [2] + [20] * 8 + [1]
N_F = 1000000
TOL = {"f64": dict(loss=1e-12, grad=1e-11), "f32": dict(loss=1e-5, grad=2e-5)}
PATH = {"f64": 7, "f32": 2}
CHUNK = 62500


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def chunked_oracle(w, lb, ub, X_f, X_u, u, n_total=N_F, with_data=True):
    """oracle.pde.burgers_loss_grad over X_f in chunks (the loss is a sum over points): -> (loss, grad)"""
    from oracle import pde
    L, G = 0.0, 0.0
    for a in range(0, X_f.shape[0], CHUNK):
        l, g, _ = pde.burgers_loss_grad(w, LAYERS, lb, ub, X_f[a:a + CHUNK], X_u, u, NU, n_f_total=n_total, with_data=False)
        L, G = L + l, G + g
    if with_data:
        l, g, _ = pde.burgers_loss_grad(w, LAYERS, lb, ub, X_f[:0], X_u, u, NU, n_f_total=n_total, with_data=True)
        L, G = L + l, G + g
    return L, G


@pytest.fixture(scope="module")
def cfg5(burgers_sets):
    import hashlib
    g = np.load(golden("burgers_eval_1e6.npz"))
    r = burgers_sets(100, N_F)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    assert sha(X_f) == str(g["sha_X_f"]) and sha(X_u) == str(g["sha_X_u"])       # the set the reference evaluated
    w0 = np.load(golden("burgers_eval.npz"))["w0"]
    assert sha(w0) == str(g["sha_w0"])
    return g, {"w0": w0, "w1": g["w1"]}, (lb, ub, X_f, X_u, u)


def engine(dtype, lb, ub, X_f, X_u, u, n_total=None):
    from pinn_native import Engine
    eng = Engine(LAYERS, lb, ub, pde="burgers", dtype=dtype)
    eng.set_collocation(X_f, n_total=n_total)
    eng.set_data(X_u, u)
    eng.set_pde_params(NU)
    assert eng.kernel_path() == PATH[dtype]          # the kernels cfg5_leg times: k_fused20d<0,8,false> / k_fused20m<0,8,false>
    return eng


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cfg5_full_set_vs_reference_golden_and_oracle(cfg5, record, dtype):
    g, ws, (lb, ub, X_f, X_u, u) = cfg5
    tol = TOL[dtype]
    eng = engine(dtype, lb, ub, X_f, X_u, u)
    for name, w in ws.items():
        eng.set_weights(w)
        loss, grad, terms = eng.loss_grad()
        # the reference's own closure (block average of its eight 125 000-point models)
        dl, dg = abs(loss - float(g["loss_" + name])) / float(g["loss_" + name]), rel(grad, g["grad_" + name])
        assert dl < tol["loss"] and dg < tol["grad"], (name, dl, dg)
        assert abs(terms[1] - float(g["mse_u_" + name])) / float(g["mse_u_" + name]) < tol["loss"] * 10
        # the oracle, chunked
        lo, go = chunked_oracle(w, lb, ub, X_f, X_u, u)
        do, dgo = abs(loss - lo) / lo, rel(grad, go)
        assert do < tol["loss"] and dgo < tol["grad"], (name, do, dgo)
        record(dtype=dtype, w=name, loss_vs_golden=dl, grad_vs_golden=dg, loss_vs_oracle=do, grad_vs_oracle=dgo)
        loss_b, grad_b, _ = eng.loss_grad()                   # bit-reproducible run to run over 61 tiles per workgroup
        assert loss_b == loss and np.array_equal(grad_b, grad)
    eng.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("block", [0, 3, 7])
def test_cfg5_shard_vs_reference_golden_and_oracle(cfg5, record, dtype, block):
    """one rank's share of an 8-rank launch: 125 000 points, denominators global (n_total = 10^6).  The reference's block
    model returns L_c = mse_u + mean_c(f^2); the engine's shard returns mse_u + (1/N) sum_c f^2."""
    g, ws, (lb, ub, X_f, X_u, u) = cfg5
    from pinn_native.parallel import shard_bounds
    lo_, hi_ = shard_bounds(N_F, 8, block)
    assert (lo_, hi_) == tuple(int(v) for v in g["bounds"][block])
    tol = TOL[dtype]
    eng = engine(dtype, lb, ub, X_f[lo_:hi_], X_u, u, n_total=N_F)
    share = (hi_ - lo_) / N_F
    for name, w in ws.items():
        eng.set_weights(w)
        loss, grad, terms = eng.loss_grad()
        mse_u, g_u = float(g["mse_u_" + name]), g["grad_mse_u_" + name]
        want_l = mse_u + share * (float(g["block_loss_" + name][block]) - mse_u)
        want_g = g_u + share * (g["block_grad_" + name][block] - g_u)
        dl, dg = abs(loss - want_l) / want_l, rel(grad, want_g)
        assert dl < tol["loss"] and dg < tol["grad"], (name, dl, dg)
        # the residual term on its own (the data term above is 70x larger at the initial weights)
        want_r = share * (float(g["block_loss_" + name][block]) - mse_u)
        assert abs(terms[0] - want_r) / want_r < tol["loss"] * 100
        lo, go = chunked_oracle(w, lb, ub, X_f[lo_:hi_], X_u, u)
        assert abs(loss - lo) / lo < tol["loss"] and rel(grad, go) < tol["grad"]
        record(dtype=dtype, w=name, block=block, loss_vs_golden=dl, grad_vs_golden=dg)
    eng.close()


@pytest.mark.parametrize("dtype,wtol,ltol", [("f64", 1e-10, 1e-11), ("f32", 3e-4, 5e-5)])
@pytest.mark.parametrize("shard", [False, True])
def test_cfg5_adam_steps_vs_oracle(cfg5, record, dtype, wtol, ltol, shard):
    """five Adam steps (the reference's lr .03, utils/neuralnetwork.py:19-22,105-116) at N_f = 10^6 (and on the
    125 000-point shard) against oracle.optim.Adam over the chunked oracle evaluation"""
    from oracle import optim
    g, ws, (lb, ub, X_f, X_u, u) = cfg5
    Xs, n_total = (X_f[:125000], N_F) if shard else (X_f, None)
    eng = engine(dtype, lb, ub, Xs, X_u, u, n_total=n_total)
    eng.set_weights(ws["w0"])
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    losses = eng.adam_run(5)
    w_dev = eng.get_weights()
    adam, w, want = optim.Adam(0.03, 0.9, 0.999, None), ws["w0"].copy(), []
    for _ in range(5):
        lo, go = chunked_oracle(w, lb, ub, Xs, X_u, u)
        want.append(lo)
        w = adam.step(w, go)
    dl, dw = float(np.max(np.abs(losses - want) / np.array(want))), rel(w_dev, w)
    record(dtype=dtype, shard=shard, adam5_loss_dev=dl, adam5_w_dev=dw)
    assert dl < ltol and dw < wtol, (dl, dw)
    eng.close()
