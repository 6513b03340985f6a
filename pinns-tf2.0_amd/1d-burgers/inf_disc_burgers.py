"""Burgers discrete-time inference (q-stage implicit Runge-Kutta) on the MI355X engine -- drop-in for the
reference's 1d-burgers/inf_disc_burgers.py: same CLI (`python 1d-burgers/inf_disc_burgers.py [hp.json]`), same hp
keys and defaults (:25-45), same class and method names, same stdout log.

The network maps x to the q stage values and the solution at t_1 (q+1 outputs); the loss (:92-95) is
    sum((U_0_model(x_0) - u_0)^2) + sum(model(x_1)^2),   U_0_model = U_1 + dt * N(U) IRK_weights^T,  N = U U_x - nu U_xx
(:57-89).  U, U_x, U_xx, the 500 x 501 IRK contraction and the whole reverse sweep run in the HIP engine
(csrc/kernels_disc.h); there is no TensorFlow.

Two deliberate differences from the reference file:
  * its L-BFGS closure (:104-116) evaluates the loss outside the gradient tape, which under TensorFlow yields no
    gradient at all; here L-BFGS uses the same (true) gradient as Adam -- the evident intent;
  * the Butcher table comes from utils/irk.py when the un-vendored PINNs/Utilities files are absent.
"""
import json
import os
import sys

import numpy as np

np.random.seed(1234)

eqnPath = "1d-burgers"
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.append(os.path.join(_root, eqnPath))
sys.path.append(os.path.join(_root, "utils"))
from logger import Logger  # noqa: E402
from neuralnetwork import NeuralNetwork, set_seed  # noqa: E402
from burgersutil import prep_data, plot_inf_disc_results  # noqa: E402

set_seed(1234)           # the reference's tf.random.set_seed(1234)

if len(sys.argv) > 1:
    with open(sys.argv[1]) as hpFile:
        hp = json.load(hpFile)
else:
    hp = {
        "N_n": 250,            # data points on the t_0 snapshot
        "q": 500,              # Runge-Kutta stages
        "layers": [1, 50, 50, 50, 501],     # [x] -> [u_1^n(x), ..., u_q^n(x), u^{n+1}(x)]
        "tf_epochs": 200, "tf_lr": 0.001, "tf_b1": 0.9, "tf_eps": 1e-08,     # Adam
        "nt_epochs": 1000, "nt_lr": 0.8, "nt_ncorr": 50,                     # L-BFGS
        "log_frequency": 10,
    }


class BurgersInformedNN(NeuralNetwork):
    pde = "burgers_disc"

    def __init__(self, hp, logger, dt, x_1, lb, ub, nu, IRK_weights, IRK_times):
        super().__init__(hp, logger, ub, lb)
        self.nu = nu
        self.dt = float(np.asarray(dt).ravel()[0])
        self.q = max(hp["q"], 1)
        self.IRK_weights = IRK_weights
        self.IRK_times = IRK_times
        self.x_1 = self.tensor(x_1)
        self._engine.set_pde_params(nu)

    def _bind(self, x_0, u_0):
        """Stage set 0 = the t_0 snapshot through the IRK table, stage set 1 = the two walls (U itself -> 0)."""
        x_0 = np.asarray(x_0, dtype=np.float64).reshape(-1)
        u_0 = np.asarray(u_0, dtype=np.float64).reshape(-1)
        key = (x_0.tobytes(), u_0.tobytes())
        if key != self._bound:
            M = self.dt * np.asarray(self.IRK_weights, dtype=np.float64)       # [q+1, q]
            self._engine.disc_set_stage(0, x_0, u_0, M)
            self._engine.disc_set_stage(1, self.x_1.reshape(-1), np.zeros(self.x_1.size), None)
            self._bound = key

    def U_0_model(self, x):
        """U_1 + dt * N(U) IRK_weights^T at the points x, [len(x), q+1]  (:57-89)."""
        if self._bound is None:
            raise RuntimeError("U_0_model needs the training set bound first (fit / grad)")
        return self._engine.disc_predict(0, np.asarray(x, dtype=np.float64).reshape(-1))

    def loss(self, u_0, u_0_pred):
        u_1_pred = self.model(self.x_1)
        return float(np.sum(np.square(np.asarray(u_0_pred) - np.asarray(u_0))) + np.sum(np.square(u_1_pred)))

    def get_params(self, numpy=False):
        return self.nu

    def fit(self, x_0, u_0):
        self.dummy_x0_tf = np.ones([np.shape(x_0)[0], self.q])      # kept for interface parity (:119-121)
        super().fit(x_0, u_0)

    def predict(self, x_star):
        return self.model(x_star)[:, -1]


if __name__ == "__main__":
    lb = np.array([-1.0])
    ub = np.array([1.0])
    idx_t_0 = 10
    idx_t_1 = 90
    nu = 0.01 / np.pi

    path = os.path.join(_root, eqnPath, "data", "burgers_shock.mat")
    x, t, dt, \
        Exact_u, x_0, u_0, x_1, x_star, u_star, \
        IRK_weights, IRK_times = prep_data(path, N_n=hp["N_n"], q=hp["q"], lb=lb, ub=ub, noise=0.0,
                                           idx_t_0=idx_t_0, idx_t_1=idx_t_1)

    logger = Logger(hp)
    pinn = BurgersInformedNN(hp, logger, dt, x_1, lb, ub, nu, IRK_weights, IRK_times)

    def error():
        u_pred = pinn.predict(x_star)
        return np.linalg.norm(u_pred - u_star, 2) / np.linalg.norm(u_star, 2)

    logger.set_error_fn(error)
    pinn.fit(x_0, u_0)

    u_1_pred = pinn.predict(x_star)
    if not os.environ.get("PINN_NO_PLOT") and pinn.is_root:
        plot_inf_disc_results(x_star, idx_t_0, idx_t_1, x_0, u_0, ub, lb, u_1_pred, Exact_u, x, t,
                              save_path=os.path.join(_root, eqnPath), save_hp=hp)
