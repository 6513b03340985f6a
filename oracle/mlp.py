"""Oracle (test infrastructure): tanh MLP, flat weight layout, Taylor-mode forward and
hand-derived reverse sweep.  numpy float64 only.

Reference being restated:
  * model: utils/neuralnetwork.py:24-37 -- Lambda 2(X-lb)/(ub-lb)-1, Dense(tanh) x (L-2),
    Dense(linear)
  * flat layout: utils/neuralnetwork.py:68-89 -- per Dense layer W.flatten() (row-major
    [fan_in, fan_out]) then b
  * derivatives: the nested tapes of 1d-burgers/inf_cont_burgers.py:65-90 compute
    u, u_x, u_xx, u_t per sample; here they are carried forward as four channels
    (h, p, q, r) = (value, d/dx, d/dt, d2/dx2)   [SURVEY.md Appendix A.1]
  * outer tape (utils/neuralnetwork.py:55-59): reverse sweep through the 4-channel forward
    [SURVEY.md Appendix A.3]
"""
import numpy as np


def layer_shapes(layers):
    return [(layers[i], layers[i + 1]) for i in range(len(layers) - 1)]


def n_params(layers):
    return sum(fi * fo + fo for fi, fo in layer_shapes(layers))


def unpack(w, layers):
    """flat vector -> [(W [fi,fo], b [fo]), ...]   (neuralnetwork.py:80-89)"""
    w = np.asarray(w, dtype=np.float64)
    out, off = [], 0
    for fi, fo in layer_shapes(layers):
        W = w[off:off + fi * fo].reshape(fi, fo)
        off += fi * fo
        b = w[off:off + fo]
        off += fo
        out.append((W, b))
    return out


def pack(params):
    """[(W,b),...] -> flat vector   (neuralnetwork.py:68-78)"""
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in params])


def forward_value(params, X, lb, ub):
    """self.model(X)  (neuralnetwork.py:29-37)."""
    h = 2.0 * (X - lb) / (ub - lb) - 1.0
    for i, (W, b) in enumerate(params):
        h = h @ W + b
        if i < len(params) - 1:
            h = np.tanh(h)
    return h


def taylor_forward(params, X, lb, ub):
    """Returns (H, P, Q, R) of the output layer, each [N, n_out], and a cache for the reverse sweep.
    P = d/dx (input column 0), Q = d/dt (input column 1), R = d2/dx2."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    s = 2.0 / (ub - lb)
    h = s * (X - lb) - 1.0
    p = np.zeros_like(h)
    q = np.zeros_like(h)
    r = np.zeros_like(h)
    p[:, 0] = s[0]
    if h.shape[1] > 1:          # 1-input nets (discrete-time models): no t channel
        q[:, 1] = s[1]
    cache = []
    L = len(params)
    for i, (W, b) in enumerate(params):
        z = h @ W + b
        zp = p @ W
        zq = q @ W
        zr = r @ W
        if i < L - 1:
            a = np.tanh(z)
            d1 = 1.0 - a * a
            d2 = -2.0 * a * d1
            cache.append((h, p, q, r, a, zp, zq, zr))
            h, p, q, r = a, d1 * zp, d1 * zq, d2 * zp * zp + d1 * zr
        else:
            cache.append((h, p, q, r, None, zp, zq, zr))
            h, p, q, r = z, zp, zq, zr
    return (h, p, q, r), cache


def taylor_backward(params, cache, hb, pb, qb, rb):
    """Reverse sweep.  hb..rb: adjoints of the output channels [N, n_out].
    Returns [(dW, db), ...] (sums over the N points)."""
    L = len(params)
    grads = [None] * L
    for i in range(L - 1, -1, -1):
        W, _ = params[i]
        h, p, q, r, a, zp, zq, zr = cache[i]
        if a is not None:
            d1 = 1.0 - a * a
            d2 = -2.0 * a * d1
            d3 = -2.0 * d1 * (1.0 - 3.0 * a * a)
            zb = d1 * hb + d2 * (zp * pb + zq * qb + zr * rb) + d3 * zp * zp * rb
            zpb = d1 * pb + 2.0 * d2 * zp * rb
            zqb = d1 * qb
            zrb = d1 * rb
        else:
            zb, zpb, zqb, zrb = hb, pb, qb, rb
        dW = h.T @ zb + p.T @ zpb + q.T @ zqb + r.T @ zrb
        db = zb.sum(axis=0)
        grads[i] = (dW, db)
        if i > 0:
            hb, pb, qb, rb = zb @ W.T, zpb @ W.T, zqb @ W.T, zrb @ W.T
    return grads


def value_backward(params, X, lb, ub, out_bar):
    """Plain (value-only) backprop for data terms: returns [(dW, db)] for sum(out * out_bar)."""
    h = 2.0 * (X - lb) / (ub - lb) - 1.0
    acts = []
    L = len(params)
    for i, (W, b) in enumerate(params):
        z = h @ W + b
        acts.append(h)
        h = np.tanh(z) if i < L - 1 else z
        if i < L - 1:
            acts[-1] = (acts[-1], h)
        else:
            acts[-1] = (acts[-1], None)
    grads = [None] * L
    gb = out_bar
    for i in range(L - 1, -1, -1):
        hin, a = acts[i]
        zb = gb if a is None else gb * (1.0 - a * a)
        grads[i] = (hin.T @ zb, zb.sum(axis=0))
        if i > 0:
            gb = zb @ params[i][0].T
    return grads


def add_grads(ga, gb):
    return [(a[0] + b[0], a[1] + b[1]) for a, b in zip(ga, gb)]
