"""Oracle (test infrastructure): discrete-time (implicit Runge-Kutta) Burgers models.  numpy float64.

Reference being restated:
  inference       1d-burgers/inf_disc_burgers.py:57-89 (U_0_model), :92-95 (loss, SUMS of squares),
                  :126-129 (predict = last output column)
  identification  1d-burgers/ide_disc_burgers.py:81-108 (U_0_model / U_1_model), :111-115 (loss),
                  :122-134 (lambda_1, lambda_2 appended to the flat vector)
  data            1d-burgers/burgersutil.py:43-61 (inference branch), :78-98 (identification branch)

Both models are the same computation over *stage sets*: a set is (x [n,1], target [n,1], M [n_out,q] or None)
and contributes  sum((U_full + N(U) M^T - target)^2)  with  N = c1 U U_x - c2 U_xx  on the first q outputs.
  inference:       sets = [(x_0, u_0, dt*IRK_weights), (x_1, 0, None)],  c1 = 1, c2 = nu
  identification:  sets = [(x_0, u_0, dt*alpha), (x_1, u_1, -dt*(beta-alpha))],  c1 = l1, c2 = exp(l2)
(ide_disc_burgers.py:104-106 builds N with the opposite sign for the second set; folded into M here.)

Third-party data absent from the reference tree: `PINNs/Utilities/IRK_weights/Butcher_IRK<q>.txt` (git submodule of
maziarraissi/PINNs, not vendored; burgersutil.py:58,91).  Those files hold the Gauss-Legendre IRK tableau
[A (q x q) | b (q) | c (q)] flattened; `gauss_legendre_butcher` below restates that published construction
(collocation at the Gauss nodes: A_ij = int_0^{c_i} l_j, b_j = int_0^1 l_j) -- parity of the table itself is
unpinned upstream, it is checked here through the order/collocation conditions.
"""
import numpy as np

from . import mlp


def gauss_legendre_butcher(q):
    """(A, b, c) of the q-stage Gauss-Legendre collocation method, by direct quadrature of the Lagrange basis
    (barycentric form) -- deliberately a different algorithm from the product's utils/irk.py."""
    xi, wq = np.polynomial.legendre.leggauss(q)
    c = 0.5 * (xi + 1.0)
    b = 0.5 * wq
    # barycentric weights of the nodes c
    d = c[:, None] - c[None, :]
    np.fill_diagonal(d, 1.0)
    # scale to avoid under/overflow of the product for large q
    logw = -np.sum(np.log(np.abs(d)), axis=1)
    sign = np.prod(np.sign(d), axis=1)
    bw = sign * np.exp(logw - logw.max())
    A = np.empty((q, q))
    for i in range(q):
        tau = c[i] * c                                  # Gauss nodes of [0, c_i]
        diff = tau[:, None] - c[None, :]                # [node, j]
        hit = np.abs(diff) < 1e-300
        diff[hit] = 1.0
        terms = bw[None, :] / diff
        ell = terms / terms.sum(axis=1, keepdims=True)  # l_j(tau_m)
        rows = hit.any(axis=1)
        ell[rows] = hit[rows].astype(float)
        A[i] = c[i] * (b @ ell)
    return A, b, c


def irk_tables_like_reference(q):
    """What burgersutil.py:58-61 would hold after loading Butcher_IRK<q>.txt: float32-rounded values.
    Returns (IRK_weights [(q+1), q] = [A; b], IRK_times [q, 1])."""
    A, b, c = gauss_legendre_butcher(q)
    tmp = np.float32(np.concatenate([A.ravel(), b, c])[:, None])
    return np.reshape(tmp[0:q * q + q], (q + 1, q)), tmp[q * q + q:]


def inference_sets(x_0, u_0, x_1, dt, IRK_weights):
    """inf_disc_burgers.py:89,92-95."""
    M = float(np.asarray(dt).ravel()[0]) * np.asarray(IRK_weights, dtype=np.float64)
    return [(np.asarray(x_0, float), np.asarray(u_0, float), M),
            (np.asarray(x_1, float), np.zeros((len(x_1), 1)), None)]


def identification_sets(x_0, u_0, x_1, u_1, dt, IRK_alpha, IRK_beta):
    """ide_disc_burgers.py:92,104-108,111-115."""
    dt = float(np.asarray(dt).ravel()[0])
    al = np.asarray(IRK_alpha, dtype=np.float64)
    # ide_disc_burgers.py:108 subtracts the two tables as loaded, i.e. in float32 (burgersutil.py:91), before the
    # product with the float64 N: keep that rounding
    diff = np.asarray(np.asarray(IRK_beta) - np.asarray(IRK_alpha), dtype=np.float64)
    return [(np.asarray(x_0, float), np.asarray(u_0, float), dt * al),
            (np.asarray(x_1, float), np.asarray(u_1, float), -dt * diff)]


def stage_prediction(params, x, lb, ub, M, c1, c2):
    (h, p, _, r), cache = mlp.taylor_forward(params, x, lb, ub)
    if M is None:
        return h, None, (h, p, r), cache
    q = M.shape[1]
    U, U_x, U_xx = h[:, :q], p[:, :q], r[:, :q]
    Nn = c1 * U * U_x - c2 * U_xx
    return h + Nn @ M.T, Nn, (h, p, r), cache


def disc_loss_grad(w, layers, lb, ub, sets, nu=None, identify=False):
    """Returns (loss, flat_grad, extras).  identify=True: w carries [lambda_1, lambda_2] at the end."""
    lb = np.asarray(lb, dtype=np.float64).reshape(-1)
    ub = np.asarray(ub, dtype=np.float64).reshape(-1)
    w = np.asarray(w, dtype=np.float64)
    if identify:
        l1, l2 = w[-2], w[-1]
        c1, c2 = l1, np.exp(l2)
        params = mlp.unpack(w[:-2], layers)
    else:
        c1, c2 = 1.0, float(nu)
        params = mlp.unpack(w, layers)
    loss, grads, dl1, dl2, sse = 0.0, None, 0.0, 0.0, []
    for x, target, M in sets:
        pred, Nn, (h, p, r), cache = stage_prediction(params, x, lb, ub, M, c1, c2)
        res = pred - target
        sse.append(float(np.sum(res * res)))
        loss += sse[-1]
        rb = 2.0 * res
        hb, pb, rrb = rb.copy(), np.zeros_like(rb), np.zeros_like(rb)
        if M is not None:
            q = M.shape[1]
            Nb = rb @ M
            hb[:, :q] += Nb * c1 * p[:, :q]
            pb[:, :q] = Nb * c1 * h[:, :q]
            rrb[:, :q] = -c2 * Nb
            dl1 += np.sum(Nb * h[:, :q] * p[:, :q])
            dl2 += np.sum(Nb * (-c2) * r[:, :q])
        g = mlp.taylor_backward(params, cache, hb, pb, np.zeros_like(rb), rrb)
        grads = g if grads is None else mlp.add_grads(grads, g)
    flat = mlp.pack(grads)
    if identify:
        flat = np.concatenate([flat, [dl1, dl2]])
    return loss, flat, {"sse": sse}


def predict_last(w, layers, lb, ub, x_star):
    """inf_disc_burgers.py:126-129."""
    params = mlp.unpack(w, layers)
    return mlp.forward_value(params, np.asarray(x_star, float), np.asarray(lb, float).reshape(-1),
                             np.asarray(ub, float).reshape(-1))[:, -1]
