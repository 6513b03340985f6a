"""Data preparation for the continuous-time Burgers scripts (host side, numpy).

Mirrors the two continuous branches of the reference's `prep_data`
(1d-burgers/burgersutil.py:27-36, 63-75, 99-131): same argument names, same return
tuples, same order of draws from numpy's global RNG, so that for the same
`np.random.seed` the training sets are bit-identical (golden hashes in
tests/golden/burgers_data.json).  The discrete-time (IRK) branches need Butcher tables
from a submodule the reference does not vendor and are out of scope.
"""
import os
import sys

import numpy as np
import scipy.io

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(os.path.dirname(_HERE), "utils"))
from sampling import lhs  # noqa: E402
from plotting import newfig, savefig, saveResultDir  # noqa: E402,F401


def prep_data(path, N_u=None, N_f=None, N_n=None, q=None, ub=None, lb=None, noise=0.0,
              idx_t_0=None, idx_t_1=None, N_0=None, N_1=None):
    if N_n is not None or q is not None or N_0 is not None or N_1 is not None:
        raise NotImplementedError(
            "discrete-time (IRK) data preparation is outside this engine's scope")
    mat = scipy.io.loadmat(path)
    t = mat["t"].reshape(-1, 1)
    x = mat["x"].reshape(-1, 1)
    Exact_u = np.real(mat["usol"]).T                  # [T, N]

    X, T = np.meshgrid(x, t)
    X_star = np.column_stack((X.ravel(), T.ravel()))
    u_star = Exact_u.reshape(-1, 1)

    # draw 1: interior samples of the full field (the identification training set;
    # discarded -- but still drawn -- in the inference branch)
    pick = np.random.choice(X_star.shape[0], N_u, replace=False)
    X_u_train, u_train = X_star[pick, :], u_star[pick, :]

    lb = X_star.min(axis=0)
    ub = X_star.max(axis=0)
    if N_f is None:
        return x, t, X, T, Exact_u, X_star, u_star, X_u_train, u_train, ub, lb

    # initial condition t = 0, then the two walls x = -1 and x = +1
    ic = (np.column_stack((X[0, :], T[0, :])), Exact_u[0, :].reshape(-1, 1))
    left = (np.column_stack((X[:, 0], T[:, 0])), Exact_u[:, 0].reshape(-1, 1))
    right = (np.column_stack((X[:, -1], T[:, -1])), Exact_u[:, -1].reshape(-1, 1))
    X_cand = np.vstack([ic[0], left[0], right[0]])
    u_cand = np.vstack([ic[1], left[1], right[1]])

    # draw 2: collocation points; draw 3: the N_u boundary/initial points
    X_f_train = lb + (ub - lb) * lhs(2, N_f)
    pick = np.random.choice(X_cand.shape[0], N_u, replace=False)
    return (x, t, X, T, Exact_u, X_star, u_star, X_cand[pick, :], u_cand[pick, :],
            X_f_train, ub, lb)


def _field_figure(X_star, u_pred, X, T, x, t, X_u_train, title):
    import matplotlib
    matplotlib.use("Agg")
    from scipy.interpolate import griddata
    U_pred = griddata(X_star, np.asarray(u_pred).ravel(), (X, T), method="cubic")
    fig, ax = newfig(1.0, 1.1)
    im = ax.imshow(U_pred.T, interpolation="nearest", cmap="rainbow",
                   extent=[t.min(), t.max(), x.min(), x.max()], origin="lower", aspect="auto")
    fig.colorbar(im)
    if X_u_train is not None:
        ax.plot(X_u_train[:, 1], X_u_train[:, 0], "kx", markersize=3, clip_on=False)
    ax.set_xlabel("t")
    ax.set_ylabel("x")
    ax.set_title(title)
    return fig


def plot_inf_cont_results(X_star, u_pred, X_u_train, u_train, Exact_u, X, T, x, t,
                          save_path=None, save_hp=None):
    """Headless (no-LaTeX) counterpart of burgersutil.py:133-206: u(t,x) heat-map with the
    training points; persisted through saveResultDir like the reference."""
    _field_figure(X_star, u_pred, X, T, x, t, X_u_train, "u(t,x)")
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp)


def plot_ide_cont_results(X_star, u_pred, X_u_train, u_train, Exact_u, X, T, x, t,
                          lambda_1_value, lambda_1_value_noisy, lambda_2_value,
                          lambda_2_value_noisy, save_path=None, save_hp=None):
    """Headless counterpart of burgersutil.py:208-263 (field + identified PDE in the title)."""
    title = "u_t + %.5f u u_x - %.7f u_xx = 0" % (lambda_1_value, lambda_2_value)
    _field_figure(X_star, u_pred, X, T, x, t, X_u_train, title)
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp)
