"""Burgers continuous-time inference on the MI355X engine -- drop-in for the reference's
1d-burgers/inf_cont_burgers.py: same CLI (`python 1d-burgers/inf_cont_burgers.py [hp.json]`,
run from the package root), same hp keys and defaults (:23-43), same class and method names,
same stdout log.  The PDE residual f = u_t + u u_x - nu u_xx (:65-90) is evaluated by the HIP
engine; there is no TensorFlow.
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
from burgersutil import prep_data, plot_inf_cont_results  # noqa: E402

set_seed(1234)           # the reference's tf.random.set_seed(1234)

if len(sys.argv) > 1:
    with open(sys.argv[1]) as hpFile:
        hp = json.load(hpFile)
else:
    hp = {
        "N_u": 100,            # data points on the initial/boundary conditions
        "N_f": 10000,          # collocation points
        "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1],
        "tf_epochs": 100, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None,      # Adam
        "nt_epochs": 200, "nt_lr": 0.8, "nt_ncorr": 50,                     # L-BFGS
        "log_frequency": 10,
    }


class BurgersInformedNN(NeuralNetwork):
    pde = "burgers"

    def __init__(self, hp, logger, X_f, ub, lb, nu):
        super().__init__(hp, logger, ub, lb)
        self.nu = nu
        X_f = np.asarray(X_f, dtype=np.float64)
        self.x_f = self.tensor(X_f[:, 0:1])
        self.t_f = self.tensor(X_f[:, 1:2])
        self._set_collocation(X_f)               # this rank's block when launched data-parallel
        self._engine.set_pde_params(nu)

    def loss(self, u, u_pred):
        """mean((u-u_pred)^2) + mean(f^2), the reference's custom loss (:59-62)."""
        f_pred = self.f_model()
        return float(np.mean(np.square(np.asarray(u) - np.asarray(u_pred))) +
                     np.mean(np.square(f_pred)))

    def f_model(self):
        """Residual at the collocation points, [N_f, 1]."""
        return self._residual_collocation()

    def get_params(self, numpy=False):
        return self.nu

    def predict(self, X_star):
        u_star = self.model(X_star)
        f_star = self.f_model()
        return u_star, f_star


def relative_l2(reference, prediction):
    return np.linalg.norm(reference - prediction, 2) / np.linalg.norm(reference, 2)


def run(hp):
    """The script body of the reference (:100-127): data, model, training, prediction, figure."""
    data_file = os.path.join(_root, eqnPath, "data", "burgers_shock.mat")
    (x, t, X, T, Exact_u, X_star, u_star,
     X_u_train, u_train, X_f, ub, lb) = prep_data(data_file, hp["N_u"], hp["N_f"], noise=0.0)

    logger = Logger(hp)
    pinn = BurgersInformedNN(hp, logger, X_f, ub, lb, nu=0.01 / np.pi)
    logger.set_error_fn(lambda: pinn.error_l2(X_star, u_star))     # = relative_l2(u_star, model(X_star)), on the device
    pinn.fit(X_u_train, u_train)

    u_pred = pinn.predict(X_star)[0]
    if not os.environ.get("PINN_NO_PLOT") and pinn.is_root:
        plot_inf_cont_results(X_star, u_pred.flatten(), X_u_train, u_train, Exact_u, X, T, x, t,
                              save_path=os.path.join(_root, eqnPath), save_hp=hp, weights=pinn.get_weights())
    return pinn


if __name__ == "__main__":
    pinn = run(hp)
