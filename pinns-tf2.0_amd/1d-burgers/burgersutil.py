"""Data preparation for the Burgers scripts (host side, numpy).

Mirrors the four branches of the reference's `prep_data` (1d-burgers/burgersutil.py:27-131): same argument
names, same return tuples, same order of draws from numpy's global RNG, so that for the same `np.random.seed`
the training sets are bit-identical (golden hashes in tests/golden/burgers_data.json, burgers_disc_*.npz).
The discrete-time branches need the Butcher tables of a submodule the reference does not vendor; utils/irk.py
generates them (or reads the original file when `PINNs/Utilities` exists next to the working directory).
"""
import os
import sys

import numpy as np
import scipy.io

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(os.path.dirname(_HERE), "utils"))
from sampling import lhs  # noqa: E402
from plotting import newfig, savefig, saveResultDir  # noqa: E402,F401
from irk import load_butcher  # noqa: E402

# burgersutil.py:17-19 of the reference: where the original Butcher files would live
utilsPath = os.path.join(".", "PINNs", "Utilities")


def prep_data(path, N_u=None, N_f=None, N_n=None, q=None, ub=None, lb=None, noise=0.0,
              idx_t_0=None, idx_t_1=None, N_0=None, N_1=None):
    mat = scipy.io.loadmat(path)
    t = mat["t"].reshape(-1, 1)
    x = mat["x"].reshape(-1, 1)
    Exact_u = np.real(mat["usol"]).T                  # [T, N]

    if None not in (N_n, q, idx_t_0, idx_t_1) and ub is not None and lb is not None:
        # discrete-time inference (burgersutil.py:43-61): N_n points of the snapshot idx_t_0, the two walls,
        # the whole snapshot idx_t_1 as test data, and the q-stage table as [(q+1), q] = [A; b]
        dt = t[idx_t_1] - t[idx_t_0]
        pick = np.random.choice(Exact_u.shape[1], N_n, replace=False)
        x_0 = x[pick, :]
        u_0 = Exact_u[idx_t_0:idx_t_0 + 1, pick].T
        u_0 = u_0 + noise * np.std(u_0) * np.random.randn(u_0.shape[0], u_0.shape[1])
        x_1 = np.vstack((lb, ub))
        tmp = load_butcher(q, utilsPath)
        IRK_weights = np.reshape(tmp[0:q ** 2 + q], (q + 1, q))
        IRK_times = tmp[q ** 2 + q:]
        return x, t, dt, Exact_u, x_0, u_0, x_1, x, Exact_u[idx_t_1, :], IRK_weights, IRK_times

    X, T = np.meshgrid(x, t)
    X_star = np.column_stack((X.ravel(), T.ravel()))
    u_star = Exact_u.reshape(-1, 1)

    # draw 1: interior samples of the full field (the identification training set;
    # discarded -- but still drawn -- in the inference branch)
    pick = np.random.choice(X_star.shape[0], N_u, replace=False)
    X_u_train, u_train = X_star[pick, :], u_star[pick, :]

    if N_0 is not None and N_1 is not None:
        # discrete-time identification (burgersutil.py:78-98): two noisy snapshots, q from the step size
        E = Exact_u.T
        pick = np.random.choice(E.shape[0], N_0, replace=False)
        x_0 = x[pick, :]
        u_0 = E[pick, idx_t_0][:, None]
        u_0 = u_0 + noise * np.std(u_0) * np.random.randn(u_0.shape[0], u_0.shape[1])
        pick = np.random.choice(E.shape[0], N_1, replace=False)
        x_1 = x[pick, :]
        u_1 = E[pick, idx_t_1][:, None]
        u_1 = u_1 + noise * np.std(u_1) * np.random.randn(u_1.shape[0], u_1.shape[1])
        dt = (t[idx_t_1] - t[idx_t_0]).item()
        q = int(np.ceil(0.5 * np.log(np.finfo(float).eps) / np.log(dt)))
        tmp = load_butcher(q, utilsPath)
        weights = np.reshape(tmp[0:q ** 2 + q], (q + 1, q))
        return x_0, u_0, x_1, u_1, x, t, dt, q, E, weights[0:-1, :], weights[-1:, :]

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
                          save_path=None, save_hp=None, weights=None):
    """Headless (no-LaTeX) counterpart of burgersutil.py:133-206: u(t,x) heat-map with the
    training points; persisted through saveResultDir like the reference.  `weights` (optional, not in the
    reference): the trained flat vector, written next to the figure as weights.npy."""
    _field_figure(X_star, u_pred, X, T, x, t, X_u_train, "u(t,x)")
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp, weights=weights)


def plot_ide_cont_results(X_star, u_pred, X_u_train, u_train, Exact_u, X, T, x, t,
                          lambda_1_value, lambda_1_value_noisy, lambda_2_value,
                          lambda_2_value_noisy, save_path=None, save_hp=None):
    """Headless counterpart of burgersutil.py:208-263 (field + identified PDE in the title)."""
    title = "u_t + %.5f u u_x - %.7f u_xx = 0" % (lambda_1_value, lambda_2_value)
    _field_figure(X_star, u_pred, X, T, x, t, X_u_train, title)
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp)


def _snapshot_figure(x, Exact_u, t, idx_t_0, idx_t_1, x_0, u_0, x_1, u_1, x_star, u_1_pred, title):
    """Exact field with the two time slices marked, and the two snapshots (data / prediction) below."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig = plt.figure(figsize=(7.5, 6.0))
    ax = fig.add_subplot(2, 1, 1)
    im = ax.imshow(Exact_u.T, interpolation="nearest", cmap="rainbow", origin="lower", aspect="auto",
                   extent=[t.min(), t.max(), x.min(), x.max()])
    fig.colorbar(im)
    for idx in (idx_t_0, idx_t_1):
        ax.axvline(float(t[idx]), color="w", linewidth=1)
    ax.set_xlabel("t")
    ax.set_ylabel("x")
    ax.set_title(title)
    ax = fig.add_subplot(2, 2, 3)
    ax.plot(x, Exact_u[idx_t_0, :], "b-", linewidth=2, label="Exact")
    ax.plot(x_0, u_0, "rx", linewidth=2, label="Data")
    ax.set_title("t = %.2f" % float(t[idx_t_0]))
    ax.set_xlabel("x")
    ax.legend(frameon=False, loc="best")
    ax = fig.add_subplot(2, 2, 4)
    ax.plot(x, Exact_u[idx_t_1, :], "b-", linewidth=2, label="Exact")
    if x_1 is not None and u_1 is not None:
        ax.plot(x_1, u_1, "rx", linewidth=2, label="Data")
    if u_1_pred is not None:
        ax.plot(x_star, u_1_pred, "r--", linewidth=2, label="Prediction")
    ax.set_title("t = %.2f" % float(t[idx_t_1]))
    ax.set_xlabel("x")
    ax.legend(frameon=False, loc="best")
    fig.tight_layout()
    return fig


def plot_inf_disc_results(x_star, idx_t_0, idx_t_1, x_0, u_0, ub, lb, u_1_pred, Exact_u, x, t,
                          save_path=None, save_hp=None):
    """Headless counterpart of burgersutil.py:208-261: exact field, the t_0 data and the predicted t_1 snapshot."""
    _snapshot_figure(x, Exact_u, t, idx_t_0, idx_t_1, x_0, u_0, None, None, x_star, u_1_pred, "u(t,x)")
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp)


def plot_ide_disc_results(x_star, t_star, idx_t_0, idx_t_1, x_0, u_0, x_1, u_1, ub, lb, U_1_pred, Exact_u,
                          lambda_1_value, lambda_1_value_noisy, lambda_2_value, lambda_2_value_noisy, x, t,
                          save_path=None, save_hp=None):
    """Headless counterpart of burgersutil.py:265-324: both data snapshots and the identified PDE."""
    title = "u_t + %.5f u u_x - %.7f u_xx = 0   (1%% noise: %.5f, %.7f)" % (
        lambda_1_value, lambda_2_value, lambda_1_value_noisy, lambda_2_value_noisy)
    E = Exact_u.T if Exact_u.shape[0] == np.size(x) and Exact_u.shape[1] == np.size(t) else Exact_u
    _snapshot_figure(x, E, t, idx_t_0, idx_t_1, x_0, u_0, x_1, u_1, x_star, None, title)
    if save_path is not None and save_hp is not None:
        saveResultDir(save_path, save_hp)
