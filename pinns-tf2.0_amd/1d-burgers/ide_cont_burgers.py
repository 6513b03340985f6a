"""Burgers continuous-time identification on the MI355X engine -- counterpart of the
reference's 1d-burgers/ide_cont_burgers.py (whose source has inconsistent indentation and
does not parse as shipped; a whitespace-only repair, tests/golden/repair_ide_cont.py, runs and is
what the fixtures of this model come from): learn lambda_1 and lambda_2 with
f = u_t + lambda_1 u u_x - exp(lambda_2) u_xx evaluated at the data points (:56-91), the two
scalars appended to the flat weight vector (:93-107), error = mean relative lambda error
(:187-192), then a second run on a fresh sample (the reference's "noise" run, :200-205).
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
from burgersutil import prep_data, plot_ide_cont_results  # noqa: E402
from neuralnetwork import NeuralNetwork, set_seed  # noqa: E402
from logger import Logger  # noqa: E402

set_seed(1234)           # the reference's tf.random.set_seed(1234)

if len(sys.argv) > 1:
    with open(sys.argv[1]) as hpFile:
        hp = json.load(hpFile)
else:
    hp = {
        "N_u": 2000,
        "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1],
        "tf_epochs": 100, "tf_lr": 0.001, "tf_b1": 0.9, "tf_eps": None,
        "nt_epochs": 500, "nt_lr": 0.8, "nt_ncorr": 50,
        "log_frequency": 10,
    }


class BurgersInformedNN(NeuralNetwork):
    pde = "burgers_ide"

    def __init__(self, hp, logger, ub, lb):
        super().__init__(hp, logger, ub, lb)

    def _extra_params(self):
        return np.array([0.0, -6.0])            # lambda_1, lambda_2 initial values (:52-53)

    @property
    def lambda_1(self):
        return self._engine.get_weights()[-2:-1]

    @property
    def lambda_2(self):
        return self._engine.get_weights()[-1:]

    def f_model(self, X_u=None):
        """Residual f = u_t + l1 u u_x - exp(l2) u_xx at the points X_u [N, 2] (:56-85); without an argument, at
        the bound data points.  [N, 1]."""
        if X_u is None:
            if self._dp:                         # a rank holds its block of the data set: evaluate the full set
                return self._engine.residual_at(self._X_bound)
            return self._engine.residual()
        return self._engine.residual_at(np.asarray(X_u, dtype=np.float64))

    def loss(self, u, u_pred):
        f_pred = self.f_model()
        return float(np.mean(np.square(np.asarray(u) - np.asarray(u_pred))) +
                     np.mean(np.square(f_pred)))

    def get_params(self, numpy=False):
        w = self._engine.get_weights()
        l1, l2 = w[-2], np.exp(w[-1])
        return (float(l1), float(l2)) if numpy else (l1, l2)

    def fit(self, X_u, u):
        self.X_u = self.tensor(X_u)
        super().fit(X_u, u)

    def predict(self, X_star):
        """(u, f) at X_star, as the reference (:169-172)."""
        return self.model(X_star), self.f_model(X_star)


if __name__ == "__main__":
    path = os.path.join(_root, eqnPath, "data", "burgers_shock.mat")
    x, t, X, T, Exact_u, X_star, u_star, \
        X_u_train, u_train, ub, lb = prep_data(path, hp["N_u"], noise=0.0)
    lambdas_star = (1.0, 0.01 / np.pi)

    logger = Logger(hp)
    pinn = BurgersInformedNN(hp, logger, ub, lb)

    def error():
        l1, l2 = pinn.get_params(numpy=True)
        l1_star, l2_star = lambdas_star
        return (abs(l1 - l1_star) / l1_star + abs(l2 - l2_star) / l2_star) / 2

    logger.set_error_fn(error)
    pinn.fit(X_u_train, u_train)
    u_pred, f_pred = pinn.predict(X_star)
    lambda_1_pred, lambda_2_pred = pinn.get_params(numpy=True)

    x, t, X, T, Exact_u, X_star, u_star, \
        X_u_train, u_train, ub, lb = prep_data(path, hp["N_u"], noise=0.01)
    pinn = BurgersInformedNN(hp, logger, ub, lb)
    pinn.fit(X_u_train, u_train)
    lambda_1_pred_noise, lambda_2_pred_noise = pinn.get_params(numpy=True)

    if pinn.is_root:
        print("l1: ", lambda_1_pred)
        print("l2: ", lambda_2_pred)
        print("l1_noise: ", lambda_1_pred_noise)
        print("l2_noise: ", lambda_2_pred_noise)

    if not os.environ.get("PINN_NO_PLOT") and pinn.is_root:
        plot_ide_cont_results(X_star, u_pred, X_u_train, u_train, Exact_u, X, T, x, t,
                              lambda_1_pred, lambda_1_pred_noise, lambda_2_pred,
                              lambda_2_pred_noise, save_path=os.path.join(_root, eqnPath),
                              save_hp=hp)
