"""Burgers discrete-time identification on the MI355X engine -- drop-in for the reference's
1d-burgers/ide_disc_burgers.py: same CLI, hp keys and defaults (:28-44), class and method names, and the same
progress lines (`l1 = ...  l2 = ...` appended, :164-167).

Two snapshots (x_0, u_0) at t_0 and (x_1, u_1) at t_1 = t_0 + dt constrain the q stage values the network
outputs (q from dt, burgersutil.py:90):
    U_0_model = U + dt N(U) alpha^T,   U_1_model = U - dt N(U) (beta - alpha)^T,   N = l1 U U_x - exp(l2) U_xx
(:81-108) and the loss is the sum of both squared misfits (:111-115) with l1, l2 trained along with the weights.
Forward, derivatives, IRK contractions, reverse sweep and both optimisers run in the HIP engine.
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
from burgersutil import prep_data, plot_ide_disc_results  # noqa: E402

set_seed(1234)           # the reference's tf.random.set_seed(1234)

if len(sys.argv) > 1:
    with open(sys.argv[1]) as hpFile:
        hp = json.load(hpFile)
else:
    hp = {
        "N_0": 199, "N_1": 201,            # data points on the two snapshots
        "layers": [1, 50, 50, 50, 0],      # the output size (q) is set from dt below
        "tf_epochs": 100, "tf_lr": 0.001, "tf_b1": 0.9, "tf_eps": None,
        "nt_epochs": 2000, "nt_lr": 0.8, "nt_ncorr": 50,
    }
hp.setdefault("log_frequency", 10)         # the reference builds Logger(frequency=10) (:225)


class BurgersInformedNN(NeuralNetwork):
    pde = "burgers_disc_ide"

    def __init__(self, hp, logger, dt, lb, ub, q, IRK_alpha, IRK_beta):
        super().__init__(hp, logger, ub, lb)
        self.dt = float(dt)
        self.q = max(q, 1)
        self.IRK_alpha = IRK_alpha
        self.IRK_beta = IRK_beta

    def _extra_params(self):
        return np.array([0.0, -6.0])            # lambda_1, lambda_2 initial values (:147-148)

    @property
    def lambda_1(self):
        return self._engine.get_weights()[-2:-1]

    @property
    def lambda_2(self):
        return self._engine.get_weights()[-1:]

    def _bind4(self, x_0, u_0, x_1, u_1):
        arrs = [np.asarray(a, dtype=np.float64).reshape(-1) for a in (x_0, u_0, x_1, u_1)]
        key = tuple(a.tobytes() for a in arrs)
        if key != self._bound:
            al = np.asarray(self.IRK_alpha, dtype=np.float64)
            # the reference subtracts the two tables as loaded (float32) before the product (:108)
            diff = np.asarray(np.asarray(self.IRK_beta) - np.asarray(self.IRK_alpha), dtype=np.float64)
            self._engine.disc_set_stage(0, arrs[0], arrs[1], self.dt * al)
            self._engine.disc_set_stage(1, arrs[2], arrs[3], -self.dt * diff)
            self._bound = key

    def U_0_model(self, x, customDummy=None):
        return self._engine.disc_predict(0, np.asarray(x, dtype=np.float64).reshape(-1))

    def U_1_model(self, x, customDummy=None):
        return self._engine.disc_predict(1, np.asarray(x, dtype=np.float64).reshape(-1))

    def loss(self, x_0, u_0, x_1, u_1):
        self._bind4(x_0, u_0, x_1, u_1)
        return self._engine.loss_grad(want_grad=False)[0]

    def grad(self, x_0, u_0, x_1, u_1):
        self._bind4(x_0, u_0, x_1, u_1)
        loss_value, flat, _ = self._engine.loss_grad()
        return loss_value, self._split(flat)

    def get_params(self, numpy=False):
        w = self._engine.get_weights()
        l1, l2 = w[-2], np.exp(w[-1])
        return (float(l1), float(l2)) if numpy else (l1, l2)

    def _log_custom(self):
        l1, l2 = self.get_params(numpy=True)
        return f"l1 = {l1:5f}  l2 = {l2:8f}"

    def createDummy(self, x):
        return np.ones([np.shape(x)[0], self.q])

    def fit(self, x_0, u_0, x_1, u_1):
        self.logger.log_train_start(self)
        self._bind4(x_0, u_0, x_1, u_1)
        # both optimiser loops are the base class's device-resident ones; _bind() must keep the four-array set
        self._bind = lambda X, u: None
        try:
            self.tf_optimization(x_0, u_0)
            self.nt_optimization(x_0, u_0)
        finally:
            del self._bind
        self.logger.log_train_end(self.tf_epochs, self._log_custom())

    def predict(self, x_star):
        return self.U_0_model(x_star), self.U_1_model(x_star)


if __name__ == "__main__":
    lb = np.array([-1.0])
    ub = np.array([1.0])
    idx_t_0 = 10
    skip = 80
    idx_t_1 = idx_t_0 + skip

    path = os.path.join(_root, eqnPath, "data", "burgers_shock.mat")
    x_0, u_0, x_1, u_1, x_star, t_star, dt, q, \
        Exact_u, IRK_alpha, IRK_beta = prep_data(path, N_0=hp["N_0"], N_1=hp["N_1"], lb=lb, ub=ub, noise=0.0,
                                                 idx_t_0=idx_t_0, idx_t_1=idx_t_1)
    lambdas_star = (1.0, 0.01 / np.pi)
    hp["layers"][-1] = q

    logger = Logger(hp)
    pinn = BurgersInformedNN(hp, logger, dt, lb, ub, q, IRK_alpha, IRK_beta)

    def error():
        l1, l2 = pinn.get_params(numpy=True)
        l1_star, l2_star = lambdas_star
        return (abs(l1 - l1_star) / l1_star + abs(l2 - l2_star) / l2_star) / 2

    logger.set_error_fn(error)
    pinn.fit(x_0, u_0, x_1, u_1)
    U_0_pred, U_1_pred = pinn.predict(x_star)
    lambda_1_pred, lambda_2_pred = pinn.get_params(numpy=True)

    # noisy case (same as before with 1 % noise), as the reference (:243-251)
    x_0, u_0, x_1, u_1, x_star, t_star, dt, q, \
        Exact_u, IRK_alpha, IRK_beta = prep_data(path, N_0=hp["N_0"], N_1=hp["N_1"], lb=lb, ub=ub, noise=0.01,
                                                 idx_t_0=idx_t_0, idx_t_1=idx_t_1)
    hp["layers"][-1] = q
    pinn = BurgersInformedNN(hp, logger, dt, lb, ub, q, IRK_alpha, IRK_beta)
    pinn.fit(x_0, u_0, x_1, u_1)
    U_0_pred, U_1_pred = pinn.predict(x_star)
    lambda_1_pred_noisy, lambda_2_pred_noisy = pinn.get_params(numpy=True)

    if pinn.is_root:                 # under torchrun the discrete-time models are replicas; rank 0 reports
        print("l1: ", lambda_1_pred)
        print("l2: ", lambda_2_pred)
        print("noisy l1: ", lambda_1_pred_noisy)
        print("noisy l2: ", lambda_2_pred_noisy)

    if not os.environ.get("PINN_NO_PLOT") and pinn.is_root:
        plot_ide_disc_results(x_star, t_star, idx_t_0, idx_t_1, x_0, u_0, x_1, u_1, ub, lb, U_1_pred, Exact_u,
                              lambda_1_pred, lambda_1_pred_noisy, lambda_2_pred, lambda_2_pred_noisy,
                              x_star, t_star, save_path=os.path.join(_root, eqnPath), save_hp=hp)
