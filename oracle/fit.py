"""Oracle (test infrastructure): the reference's training driver restated over the numpy oracle -- what bench.py's
`cpu_baseline` (kind "port") times on the GPU box's host cores, and what a test compares with the reference's own
printed log.

Reference being restated (pierremtb/PINNs-TF2.0):
  utils/neuralnetwork.py:138-149   fit: log_train_start, Adam loop, L-BFGS, log_train_end
  utils/neuralnetwork.py:105-116   tf_optimization: one loss+grad evaluation and one Adam update per epoch, every
                                   epoch's loss handed to the logger
  utils/neuralnetwork.py:118-136   nt_optimization: custom_lbfgs.lbfgs on the flat-vector closure, the logger as log_fn
  utils/logger.py:45-51            a progress line every `log_frequency` epochs (elapsed, lap, loss)
  utils/logger.py:56-60            the closing line calls the error function once
  1d-burgers/inf_cont_burgers.py:114-116   error = ||u_star - model(X_star)||_2 / ||u_star||_2

Like the reference, the model ends at the last EVALUATED L-BFGS iterate, not at the returned x.
"""
import time

import numpy as np

from . import mlp, optim, pde


def burgers_fit(w0, layers, lb, ub, X_f, X_u, u, nu, X_star, u_star, tf_epochs=100, nt_epochs=200, tf_lr=0.03, tf_b1=0.9,
                tf_eps=None, nt_lr=0.8, nt_ncorr=50, log_frequency=10, out=None):
    """-> dict(w = final model weights, lines = the progress lines, error = final relative L2 error, evals = number of
    loss+grad evaluations, fit_seconds = wall time of the whole call)"""
    t0 = t_last = time.time()
    lines = []

    def emit(text):
        lines.append(text)
        if out is not None:
            out.write(text + "\n")

    def log_epoch(epoch, loss, is_iter=False):                       # utils/logger.py:45-51
        nonlocal t_last
        if epoch % log_frequency:
            return
        now = time.time()
        emit("%s = %6d  elapsed = %s (+%s)  loss = %.4e  " % (
            "nt_epoch" if is_iter else "tf_epoch", epoch, time.strftime("%M:%S", time.gmtime(now - t0)),
            ("%09.6f" % (now - t_last))[:4], loss))
        t_last = now

    n_evals = [0]

    def closure(w):                                                  # utils/neuralnetwork.py:91-103
        n_evals[0] += 1
        loss, grad, _ = pde.burgers_loss_grad(w, layers, lb, ub, X_f, X_u, u, nu)
        return loss, grad

    emit("\nTraining started\n================")
    emit("-- Starting Adam optimization --")
    w = np.array(w0, dtype=np.float64)
    adam = optim.Adam(tf_lr, tf_b1, 0.999, tf_eps)
    for epoch in range(tf_epochs):                                   # utils/neuralnetwork.py:105-109
        loss, grad = closure(w)
        w = adam.step(w, grad)
        log_epoch(epoch, loss)
    emit("-- Starting LBFGS optimization --")
    res = optim.lbfgs(closure, w, nt_epochs, nt_lr, nt_ncorr, tol_fun=1.0 * np.finfo(float).eps,
                      log_fn=lambda it, f, is_iter: log_epoch(it, f, True))
    if res is not None:
        w = res["x_model"]
    up = mlp.forward_value(mlp.unpack(w, layers), X_star, lb, ub)     # inf_cont_burgers.py:114-116
    err = float(np.linalg.norm(u_star - up, 2) / np.linalg.norm(u_star, 2))
    emit("==================")
    emit("Training finished (epoch %d): duration = %s  error = %.4e  " % (
        tf_epochs + nt_epochs, time.strftime("%M:%S", time.gmtime(time.time() - t0)), err))
    return {"w": w, "lines": lines, "error": err, "evals": n_evals[0], "fit_seconds": time.time() - t0}
