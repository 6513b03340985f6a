"""L-BFGS without line search on a flat numpy vector, with the call contract of the
reference's utils/custom_lbfgs.py (:39-236): `lbfgs(opfunc, x, config, state, do_verbose,
log_fn)` and the Lua-like `Struct` whose unset attributes read as 0.

This is the *portable* driver: `opfunc` may be any callable x -> (f, g) -- in this
repository it is the GPU closure returned by NeuralNetwork.get_loss_and_flat_grad.
NeuralNetwork.nt_optimization does not go through here: it runs the same algorithm
device-resident (csrc/kernels_optim.h).  Semantics kept from the reference: first step
min(1, 1/|g|_1), fixed step `learningRate` afterwards, curvature pairs accepted only when
y.s > 1e-10, limited history `nCorrection`, no re-evaluation on the final iteration, the
(:192-215) stopping tests, log_fn after the tests, module globals `final_loss`/`times`.
"""
import time
from collections import deque

import numpy as np

final_loss = None
times = []


class Struct(object):
    """Attribute bag; reading an attribute that was never set yields 0."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return 0


def dot(a, b):
    return float(np.dot(np.ravel(a), np.ravel(b)))


def _as_vector(x):
    return np.array(np.asarray(x, dtype=np.float64).ravel(), copy=True)


def _two_loop(grad, pairs, h_diag):
    """Inverse-Hessian-vector product -H*grad from the stored (s, y, 1/y.s) triples."""
    q = -grad
    alphas = []
    for s, y, rho in reversed(pairs):
        a = dot(s, q) * rho
        alphas.append(a)
        q = q - a * y
    r = q * h_diag
    for (s, y, rho), a in zip(pairs, reversed(alphas)):
        r = r + (a - dot(y, r) * rho) * s
    return r


def lbfgs(opfunc, x, config, state, do_verbose, log_fn):
    global final_loss, times

    max_iter = config.maxIter
    if max_iter == 0:
        return None
    max_eval = config.maxEval or max_iter * 1.25
    tol_fun = config.tolFun or 1e-5
    tol_x = config.tolX or 1e-19
    history = config.nCorrection or 100
    step = config.learningRate or 1
    say = print if config.verbose else (lambda msg: None)

    x = _as_vector(x)
    f, g = opfunc(x)
    f, g = float(f), _as_vector(g)
    f_hist = [f]
    n_eval = 1
    state.funcEval = state.funcEval + 1

    if np.abs(g).sum() <= tol_fun:
        say("optimality condition below tolFun")
        return x, f_hist

    pairs = deque(maxlen=history)
    h_diag = 1.0
    d = t = g_prev = f_prev = None
    times = []
    tick = time.perf_counter()
    n_iter = 0
    while n_iter < max_iter:
        n_iter += 1
        state.nIter = state.nIter + 1

        if state.nIter == 1:
            d = -g
            pairs.clear()
            h_diag = 1.0
        else:
            y = g - g_prev
            s = d * t
            ys = dot(y, s)
            if ys > 1e-10:
                pairs.append((s, y, 1.0 / ys))     # deque drops the oldest pair when full
                h_diag = ys / dot(y, y)
            d = _two_loop(g, list(pairs), h_diag)
        g_prev, f_prev = g, f

        if dot(g, d) > -tol_x:
            say("Can not make progress along direction.")
            break
        t = min(1.0, 1.0 / np.abs(g).sum()) if state.nIter == 1 else step

        x = x + t * d
        evaluated = 0
        if n_iter != max_iter:
            f, g = opfunc(x)
            f, g = float(f), _as_vector(g)
            f_hist.append(f)
            evaluated = 1
        n_eval += evaluated
        state.funcEval = state.funcEval + evaluated

        if n_iter == max_iter:
            break
        if n_eval >= max_eval:
            say("max nb of function evals")
            break
        if np.abs(g).sum() <= tol_fun:
            say("optimality condition below tolFun")
            break
        if np.abs(d * t).sum() <= tol_x:
            say("step size below tolX")
            break
        if abs(f - f_prev) < tol_x:
            say("function value changing less than tolX")
            break

        if do_verbose:
            log_fn(n_iter, f, True)
            now = time.perf_counter()
            times.append(1000.0 * (now - tick))
            tick = now
        if n_iter == max_iter - 1:
            final_loss = f

    state.old_dirs = [p[0] for p in pairs]
    state.old_stps = [p[1] for p in pairs]
    state.Hdiag = h_diag
    state.g_old = g_prev
    state.f_old = f_prev
    state.t = t
    state.d = d
    return x, f_hist, n_eval
