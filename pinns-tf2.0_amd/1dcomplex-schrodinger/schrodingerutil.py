"""Data preparation for the Schrodinger script (host side, numpy).

Mirrors the reference's `prep_data` (1dcomplex-schrodinger/schrodingerutil.py:21-61):
same arguments, same 20-value return tuple, same order of global-RNG draws
(choice over x, choice over t, then lhs), so the sets are bit-identical for the same
seed (hashes in tests/golden/schrodinger_data.json).
"""
import os
import sys

import numpy as np
import scipy.io

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(os.path.dirname(_HERE), "utils"))
from sampling import lhs  # noqa: E402
from plotting import newfig, savefig, saveResultDir  # noqa: E402,F401


def prep_data(path, N_0, N_b, N_f, noise):
    mat = scipy.io.loadmat(path)
    t = mat["tt"].reshape(-1, 1)
    x = mat["x"].reshape(-1, 1)
    psi = mat["uu"]                                   # complex [Nx, Nt]
    Exact_u, Exact_v = np.real(psi), np.imag(psi)
    Exact_h = np.sqrt(Exact_u ** 2 + Exact_v ** 2)

    X, T = np.meshgrid(x, t)
    X_star = np.column_stack((X.ravel(), T.ravel()))
    u_star = Exact_u.T.reshape(-1, 1)
    v_star = Exact_v.T.reshape(-1, 1)
    h_star = Exact_h.T.reshape(-1, 1)

    lb = np.array([-5.0, 0.0])
    ub = np.array([5.0, np.pi / 2])

    rows = np.random.choice(x.shape[0], N_0, replace=False)
    x0, u0, v0 = x[rows, :], Exact_u[rows, 0:1], Exact_v[rows, 0:1]
    cols = np.random.choice(t.shape[0], N_b, replace=False)
    tb = t[cols, :]

    X0 = np.column_stack((x0[:, 0], np.zeros(N_0)))   # (x0, 0)
    H0 = np.column_stack((u0[:, 0], v0[:, 0]))
    X_f = lb + (ub - lb) * lhs(2, N_f)
    return (x, t, X, T, Exact_u, Exact_v, Exact_h, X_star, u_star, v_star, h_star, X_f,
            ub, lb, tb, x0, u0, v0, X0, H0)


def plot_inf_cont_results(X_star, u_pred, v_pred, h_pred, Exact_h, X, T, x, t, ub, lb, x0, tb,
                          save_path=None, save_hp=None):
    """Headless counterpart of schrodingerutil.py:64-147: |h(t,x)| heat-map."""
    import matplotlib
    matplotlib.use("Agg")
    from scipy.interpolate import griddata
    H_pred = griddata(X_star, np.asarray(h_pred).ravel(), (X, T), method="cubic")
    fig, ax = newfig(1.0, 0.9)
    im = ax.imshow(H_pred.T, interpolation="nearest", cmap="YlGnBu",
                   extent=[lb[1], ub[1], lb[0], ub[0]], origin="lower", aspect="auto")
    fig.colorbar(im)
    ax.set_xlabel("t")
    ax.set_ylabel("x")
    ax.set_title("|h(t,x)|")
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp)
