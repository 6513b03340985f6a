"""Oracle (test infrastructure): PDE residuals, losses and flat gradients.  numpy float64.

Reference being restated:
  Burgers inference       1d-burgers/inf_cont_burgers.py:59-62 (loss), :65-90 (f_model)
  Burgers identification  1d-burgers/ide_cont_burgers.py:56-85 (f_model), :88-91 (loss),
                          :93-107 (lambda_1, lambda_2 appended to the flat vector)
  Schrodinger             1dcomplex-schrodinger/inf_cont_schrodinger.py:60-76 (uvx_model),
                          :79-105 (f_model), :107-129 (loss)
Adjoint seeds: SURVEY.md Appendix A.2.

Every function takes the flat weight vector in the reference layout and returns
(loss, flat_grad, extras).  `n_f_total` lets a caller evaluate a *shard* of the
collocation set while normalising by the global N_f (multi-GPU equivalence tests);
`with_data=False` drops the replicated data term so shard results can be summed.
"""
import numpy as np

from . import mlp


def burgers_residual(params, X, lb, ub, c1, c2):
    (h, p, q, r), cache = mlp.taylor_forward(params, X, lb, ub)
    u, u_x, u_t, u_xx = h, p, q, r
    f = u_t + c1 * u * u_x - c2 * u_xx          # inf_cont_burgers.py:90 / ide_cont_burgers.py:85
    return f, (u, u_x, u_t, u_xx), cache


def burgers_loss_grad(w, layers, lb, ub, X_f, X_u, u_data, nu, n_f_total=None, with_data=True):
    """Burgers inference (inf_cont_burgers.py:59-62): mean((u-u_pred)^2) + mean(f^2)."""
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    params = mlp.unpack(w, layers)
    N_f = X_f.shape[0] if n_f_total is None else n_f_total
    f, (u, u_x, u_t, u_xx), cache = burgers_residual(params, X_f, lb, ub, 1.0, nu)
    mse_f = np.sum(f * f) / N_f
    fb = 2.0 * f / N_f
    grads = mlp.taylor_backward(params, cache, fb * u_x, fb * u, fb, -nu * fb)
    mse_u = 0.0
    if with_data:
        u_pred = mlp.forward_value(params, X_u, lb, ub)
        d = u_pred - u_data
        mse_u = np.mean(d * d)
        grads = mlp.add_grads(grads, mlp.value_backward(params, X_u, lb, ub, 2.0 * d / d.size))
    loss = mse_u + mse_f
    return loss, mlp.pack(grads), {"f": f, "mse_u": mse_u, "mse_f": mse_f}


def burgers_ide_loss_grad(w, layers, lb, ub, X_u, u_data):
    """Burgers identification: flat vector = network weights + [lambda_1, lambda_2]
    (ide_cont_burgers.py:98-107); residual at the data points with c1=l1, c2=exp(l2)."""
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    l1, l2 = w[-2], w[-1]
    c2 = np.exp(l2)
    params = mlp.unpack(w[:-2], layers)
    N = X_u.shape[0]
    f, (u, u_x, u_t, u_xx), cache = burgers_residual(params, X_u, lb, ub, l1, c2)
    d = u - u_data
    loss = np.mean(d * d) + np.mean(f * f)
    fb = 2.0 * f / N
    hb = fb * l1 * u_x + 2.0 * d / N
    grads = mlp.taylor_backward(params, cache, hb, fb * l1 * u, fb, -c2 * fb)
    dl1 = np.sum(fb * u * u_x)
    dl2 = np.sum(fb * (-c2) * u_xx)
    g = np.concatenate([mlp.pack(grads), [dl1, dl2]])
    return loss, g, {"f": f, "mse_u": np.mean(d * d), "mse_f": np.mean(f * f)}


def schrodinger_residual(params, X, lb, ub):
    (h, p, q, r), cache = mlp.taylor_forward(params, X, lb, ub)
    u, v = h[:, 0:1], h[:, 1:2]
    u_t, v_t = q[:, 0:1], q[:, 1:2]
    u_xx, v_xx = r[:, 0:1], r[:, 1:2]
    h2 = u * u + v * v
    f_u = u_t + 0.5 * v_xx + h2 * v             # inf_cont_schrodinger.py:101-103
    f_v = v_t - 0.5 * u_xx - h2 * u
    return f_u, f_v, (h, p, q, r), cache


def schrodinger_loss_grad(w, layers, lb, ub, X_f, X_lb, X_ub, X0, uv0, n_f_total=None,
                          with_small_terms=True):
    """inf_cont_schrodinger.py:107-129.  X0 is the [N0,2] input of the IC term: the reference
    script passes x0 of shape [N0,1] which broadcasts to (x0,x0) ("compat"); the evident intent
    is (x0,0).  The caller chooses by constructing X0."""
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    params = mlp.unpack(w, layers)
    N_f = X_f.shape[0] if n_f_total is None else n_f_total
    f_u, f_v, (h, p, q, r), cache = schrodinger_residual(params, X_f, lb, ub)
    u, v = h[:, 0:1], h[:, 1:2]
    mse_f = (np.sum(f_u * f_u) + np.sum(f_v * f_v)) / N_f
    gu, gv = 2.0 * f_u / N_f, 2.0 * f_v / N_f
    hb = np.concatenate([gu * 2 * u * v - gv * (3 * u * u + v * v),
                         gu * (u * u + 3 * v * v) - gv * 2 * u * v], axis=1)
    qb = np.concatenate([gu, gv], axis=1)
    rb = np.concatenate([-0.5 * gv, 0.5 * gu], axis=1)
    grads = mlp.taylor_backward(params, cache, hb, np.zeros_like(hb), qb, rb)
    mse_0 = mse_b = 0.0
    if with_small_terms:
        # initial condition (loss():109-118 with uv_pred = model(X) from grad(), neuralnetwork.py:57)
        pred0 = mlp.forward_value(params, X0, lb, ub)
        d0 = pred0 - uv0
        N0 = X0.shape[0]
        mse_0 = np.sum(d0[:, 0] ** 2) / N0 + np.sum(d0[:, 1] ** 2) / N0
        grads = mlp.add_grads(grads, mlp.value_backward(params, X0, lb, ub, 2.0 * d0 / N0))
        # periodic boundary: u, v, u_x, v_x at (lb_x, tb) vs (ub_x, tb)   (loss():119-123)
        (hl, pl, ql, rl), cl = mlp.taylor_forward(params, X_lb, lb, ub)
        (hu, pu, qu, ru), cu = mlp.taylor_forward(params, X_ub, lb, ub)
        Nb = X_lb.shape[0]
        dh, dp = hl - hu, pl - pu
        mse_b = (np.sum(dh ** 2) + np.sum(dp ** 2)) / Nb
        z = np.zeros_like(dh)
        grads = mlp.add_grads(grads, mlp.taylor_backward(params, cl, 2 * dh / Nb, 2 * dp / Nb, z, z))
        grads = mlp.add_grads(grads, mlp.taylor_backward(params, cu, -2 * dh / Nb, -2 * dp / Nb, z, z))
    loss = mse_0 + mse_b + mse_f
    return loss, mlp.pack(grads), {"f_u": f_u, "f_v": f_v, "mse_0": mse_0, "mse_b": mse_b,
                                   "mse_f": mse_f}
