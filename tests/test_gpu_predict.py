"""GPU: the post-training half of the path -- `self.model(X_star)` (utils/neuralnetwork.py:151-153),
f_model at caller-supplied points (1d-burgers/ide_cont_burgers.py:169-172) and the scripts' error metric
(1d-burgers/inf_cont_burgers.py:114-116, utils/logger.py:56-60) -- on the MFMA forward sweeps of
csrc/kernels_predict20.h and the device-side reduction pinn_error_l2, through the C ABI, against the oracle / numpy."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

NU = 0.01 / np.pi
LB, UB = np.array([-1.0, 0.0]), np.array([1.0, 0.99])


def _net(H, seed, ide=False):
    rs = np.random.RandomState(seed)
    layers = [2] + [20] * H + [1]
    P = sum(a * b + b for a, b in zip(layers[:-1], layers[1:]))
    w = 0.9 / np.sqrt(20.0) * rs.standard_normal(P)
    if ide:
        w = np.concatenate([w, [0.7, -4.0]])
    return layers, w, rs


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("H", [1, 2, 3, 8, 11])
def test_width20_forward_sweeps_any_depth(dtype, H):
    """k_fwd20d / k_fwd20f, value channel and Taylor channels, depths 1..11, ragged point counts incl. > one pass of
    the persistent grid (f32: 8 x 256 workgroups of 64 points = 131072)"""
    import pinn_native
    from oracle import mlp, pde
    layers, w, rs = _net(H, 100 + H)
    eng = pinn_native.Engine(layers, LB, UB, pde="burgers", dtype=dtype)
    eng.set_pde_params(NU)
    eng.set_weights(w)
    tol_u, tol_f = (1e-13, 1e-11) if dtype == "f64" else (3e-6, 1e-4)
    for n in (1, 65, 1000, 140001 if H == 8 else 777):
        X = np.column_stack([rs.uniform(LB[0], UB[0], n), rs.uniform(LB[1], UB[1], n)])
        u = eng.predict(X)
        assert u.shape == (n, 1)
        ref = mlp.forward_value(mlp.unpack(w, layers), X, LB, UB)
        assert np.max(np.abs(u - ref)) <= tol_u * max(1.0, np.max(np.abs(ref))), (n, np.max(np.abs(u - ref)))
        if n <= 1000:
            Xu = X[:1]
            _, _, ex = pde.burgers_loss_grad(w, layers, LB, UB, X, Xu, np.zeros((1, 1)), NU)
            f = eng.residual_at(X)
            scale = max(1.0, np.max(np.abs(ex["f"])))
            assert np.max(np.abs(f - ex["f"])) <= tol_f * scale, (n, np.max(np.abs(f - ex["f"])) / scale)
            eng.set_collocation(X); eng.set_data(Xu, np.zeros((1, 1)))
            assert np.max(np.abs(eng.residual() - ex["f"])) <= tol_f * scale
    eng.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_identification_residual_at_uses_the_lambdas(dtype):
    import pinn_native
    from oracle import pde
    layers, w, rs = _net(8, 7, ide=True)
    X = np.column_stack([rs.uniform(LB[0], UB[0], 300), rs.uniform(LB[1], UB[1], 300)])
    u = rs.standard_normal((300, 1))
    eng = pinn_native.Engine(layers, LB, UB, pde="burgers_ide", dtype=dtype)
    eng.set_data(X, u); eng.set_weights(w)
    _, _, ex = pde.burgers_ide_loss_grad(w, layers, LB, UB, X, u)
    tol = 1e-11 if dtype == "f64" else 1e-4
    scale = max(1.0, np.max(np.abs(ex["f"])))
    assert np.max(np.abs(eng.residual_at(X) - ex["f"])) <= tol * scale
    assert np.max(np.abs(eng.residual() - ex["f"])) <= tol * scale
    eng.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_error_l2_on_the_device_equals_numpy(burgers_sets, dtype, record):
    """pinn_error_l2 against np.linalg.norm on the predicted field (the value the reference's error() returns):
    1e-14 relative in float64 -- both sides are float64 sums of the same 25600 float64 terms -- repeat calls
    (cached grid) bit-identical, and a changed reference field or grid is noticed"""
    import pinn_native
    g = np.load(golden("burgers_eval.npz"))
    r = burgers_sets(100, 10000)
    X_star, u_star, lb, ub = r[5], r[6], r[11], r[10]
    eng = pinn_native.Engine([2] + [20] * 8 + [1], lb, ub, pde="burgers", dtype=dtype)
    eng.set_weights(g["w0"])
    up = eng.predict(X_star)
    want = float(np.linalg.norm(u_star - up, 2) / np.linalg.norm(u_star, 2))
    got = eng.error_l2(X_star, u_star)
    record(dtype=dtype, err_device=got, err_numpy=want, rel_dev=abs(got - want) / want)
    assert abs(got - want) <= 1e-14 * want, (got, want)
    assert abs(got - float(g["err0"])) < (1e-11 if dtype == "f64" else 2e-5)      # the reference's own error at w0
    assert eng.error_l2(X_star, u_star) == got
    u2 = u_star + 0.125
    want2 = float(np.linalg.norm(u2 - up, 2) / np.linalg.norm(u2, 2))
    assert abs(eng.error_l2(X_star, u2) - want2) <= 1e-14 * want2
    n = 1001                                                                      # another grid, ragged
    want3 = float(np.linalg.norm(u_star[:n] - up[:n], 2) / np.linalg.norm(u_star[:n], 2))
    assert abs(eng.error_l2(X_star[:n], u_star[:n]) - want3) <= 1e-14 * want3
    eng.set_weights(g["w0"] * 1.001)                                              # new weights, same cached grid
    up4 = eng.predict(X_star[:n])
    want4 = float(np.linalg.norm(u_star[:n] - up4, 2) / np.linalg.norm(u_star[:n], 2))
    assert abs(eng.error_l2(X_star[:n], u_star[:n]) - want4) <= 1e-14 * want4 and want4 != want3
    eng.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_error_l2_modulus_kind_schrodinger(schrodinger_sets, dtype):
    """|h| = sqrt(u^2 + v^2) against h_star (1dcomplex-schrodinger/inf_cont_schrodinger.py:155-158), two-output net on
    the shape-generic forward sweep + k_pick_values"""
    import json
    import pinn_native
    g = np.load(golden("schrodinger_eval_small.npz"))
    hp = json.loads(str(g["hp"]))
    r = schrodinger_sets(50, 50, 1024)
    X_star, h_star, ub, lb = r[7], r[10], r[12], r[13]
    eng = pinn_native.Engine(hp["layers"], lb, ub, pde="schrodinger", dtype=dtype)
    eng.set_weights(g["w0"])
    uv = eng.predict(X_star)
    h = np.sqrt(uv[:, 0:1] ** 2 + uv[:, 1:2] ** 2)
    want = float(np.linalg.norm(h_star - h, 2) / np.linalg.norm(h_star, 2))
    got = eng.error_l2(X_star, h_star, modulus=True)
    assert abs(got - want) <= 1e-13 * want, (got, want)
    uv_ref = np.concatenate([r[8], r[9]], axis=1)                                  # element-wise kind over [n][2]
    want0 = float(np.linalg.norm(uv_ref - uv) / np.linalg.norm(uv_ref))
    assert abs(eng.error_l2(X_star, uv_ref) - want0) <= 1e-13 * want0
    eng.close()


def test_status_records_the_first_nonfinite_evaluation(burgers_sets):
    """SURVEY 5 failure detection: the reference lets a NaN loss propagate (custom_lbfgs.py:154); the engine does the
    same and additionally records WHEN it happened"""
    import pinn_native
    g = np.load(golden("burgers_eval_small.npz"))
    r = burgers_sets(64, 2048)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    for dtype in ("f64", "f32"):
        eng = pinn_native.Engine([2] + [20] * 8 + [1], lb, ub, pde="burgers", dtype=dtype)
        eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(NU)
        eng.set_weights(g["w0"])
        eng.loss_grad(); eng.loss_grad()
        assert eng.status() == (2, 0)
        w = g["w0"].copy(); w[100] = np.nan
        eng.set_weights(w)
        loss, _, _ = eng.loss_grad()
        assert not np.isfinite(loss)
        eng.set_weights(g["w0"])
        eng.adam_init(0.03); eng.adam_run(2)
        assert eng.status() == (5, 3)                       # sticky: the FIRST bad evaluation
        eng.close()
