"""Nonlinear Schrodinger continuous-time inference on the MI355X engine -- drop-in for the
reference's 1dcomplex-schrodinger/inf_cont_schrodinger.py: same CLI, hp keys/defaults
(:19-41), class and method names.  h = u + iv, i h_t + h_xx/2 + |h|^2 h = 0; the loss is
IC misfit + periodic boundary (u, v, u_x, v_x) + residual (:107-129), all on the GPU.
"""
import json
import os
import sys

import numpy as np

eqnPath = "1dcomplex-schrodinger"
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.append(os.path.join(_root, eqnPath))
sys.path.append(os.path.join(_root, "utils"))
from schrodingerutil import prep_data, plot_inf_cont_results  # noqa: E402
from logger import Logger  # noqa: E402
from neuralnetwork import NeuralNetwork, set_seed  # noqa: E402

set_seed(1234)           # the reference's tf.random.set_seed(1234)

np.random.seed(1234)

if len(sys.argv) > 1:
    with open(sys.argv[1]) as hpFile:
        hp = json.load(hpFile)
else:
    hp = {
        "N_0": 50, "N_b": 50, "N_f": 20000,
        "layers": [2, 100, 100, 100, 100, 2],
        "tf_epochs": 200, "tf_lr": 0.05, "tf_b1": 0.99, "tf_eps": 1e-1,
        "nt_epochs": 0, "nt_lr": 1.2, "nt_ncorr": 50,
        "log_frequency": 10,
    }


class SchrodingerInformedNN(NeuralNetwork):
    pde = "schrodinger"

    def __init__(self, hp, logger, X_f, tb, ub, lb):
        super().__init__(hp, logger, ub, lb)
        self._quiet_parts = bool(hp.get("quiet_loss_parts", False))   # hp switch: drop the per-epoch mse lines
        tb = np.asarray(tb, dtype=np.float64)
        X_lb = np.concatenate((0 * tb + lb[0], tb), 1)     # (lb_x, tb)
        X_ub = np.concatenate((0 * tb + ub[0], tb), 1)     # (ub_x, tb)
        self.X_lb = self.tensor(X_lb)
        self.X_ub = self.tensor(X_ub)
        X_f = np.asarray(X_f, dtype=np.float64)
        self.x_f = self.tensor(X_f[:, 0:1])
        self.t_f = self.tensor(X_f[:, 1:2])
        self._set_collocation(X_f)               # these two are split over the ranks of a data-parallel launch
        self._set_boundary(X_lb, X_ub)

    def f_model(self):
        """(f_u, f_v) at the collocation points, each [N_f, 1]."""
        f = self._residual_collocation()
        return f[:, 0:1], f[:, 1:2]

    def loss(self, uv, uv_pred):
        """Total loss at the current weights for the bound IC set; the three parts are printed
        like the reference does on every evaluation (:128)."""
        total, _, terms = self._engine.loss_grad(want_grad=False)
        if self.is_root:
            print(f"mse_0 {terms[1]}    mse_b {terms[2]}    mse_f    {terms[0]}")
        return total

    def _adam_chunk(self, n):
        """The reference prints the three parts inside loss(), i.e. once per epoch (:128): same lines, same order
        relative to the progress lines, emitted when the chunk's losses come back from the device."""
        terms = self._engine.adam_run_terms(n)
        if not self._quiet_parts and self.is_root:
            for res, data, bnd in terms:
                print(f"mse_0 {data}    mse_b {bnd}    mse_f    {res}")
        return terms.sum(axis=1)

    def _adam_enqueue(self, n):                      # the same chunk one chunk behind the GPU (utils/neuralnetwork.py _pipelined)
        return self._engine.adam_enqueue(n, terms=True)

    def _adam_collect(self, ticket):
        terms = self._engine.adam_collect(ticket)
        if not self._quiet_parts and self.is_root:
            for res, data, bnd in terms:
                print(f"mse_0 {data}    mse_b {bnd}    mse_f    {res}")
        return terms.sum(axis=1)

    def predict(self, X_star):
        h_pred = self.model(X_star)
        return h_pred[:, 0:1], h_pred[:, 1:2]


if __name__ == "__main__":
    path = os.path.join(_root, eqnPath, "data", "NLS.mat")
    x, t, X, T, Exact_u, Exact_v, Exact_h, \
        X_star, u_star, v_star, h_star, X_f, \
        ub, lb, tb, x0, u0, v0, X0, H0 = prep_data(path, hp["N_0"], hp["N_b"], hp["N_f"],
                                                   noise=0.0)
    logger = Logger(hp)
    pinn = SchrodingerInformedNN(hp, logger, X_f, tb, ub, lb)

    def error():      # ||h_star - |h_pred| ||_2 / ||h_star||_2 (:155-158), reduced on the device
        return pinn.error_l2(X_star, h_star, modulus=True)

    logger.set_error_fn(error)
    # same call as the reference (:164): x0 is [N_0, 1]; see neuralnetwork._as_points
    pinn.fit(x0, np.concatenate([u0, v0], axis=1))

    u_pred, v_pred = pinn.predict(X_star)
    h_pred = np.sqrt(u_pred ** 2 + v_pred ** 2)
    if not os.environ.get("PINN_NO_PLOT") and pinn.is_root:
        plot_inf_cont_results(X_star, u_pred, v_pred, h_pred, Exact_h, X, T, x, t, ub, lb, x0, tb,
                              save_path=os.path.join(_root, eqnPath), save_hp=hp)
