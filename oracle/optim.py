"""Oracle (test infrastructure): optimisers, numpy float64.

Reference being restated:
  Adam    utils/neuralnetwork.py:19-22 (hyper-parameters; epsilon=None -> Keras 1e-7),
          :105-116 (full-batch loop).  Update = TF-2.0 ResourceApplyAdam:
          alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (1-b1)(g-m); v += (1-b2)(g^2-v);
          theta -= alpha*m/(sqrt(v)+eps)        [SURVEY.md Appendix A.4]
  L-BFGS  utils/custom_lbfgs.py:39-236, statement by statement, including
          - first step t = min(1, 1/sum|g|), later t = learningRate (:159-163)
          - curvature pair accepted only if y.s > 1e-10 (:102)
          - no re-evaluation on the last iteration (:176), so the "model" ends at the last
            *evaluated* point while the returned x is one step further
          - log_fn is called after the break tests (:217-218)
"""
import numpy as np


class Adam(object):
    def __init__(self, lr, b1=0.9, b2=0.999, eps=None):
        self.lr, self.b1, self.b2 = lr, b1, b2
        self.eps = 1e-7 if eps is None else eps
        self.t = 0
        self.m = None
        self.v = None

    def step(self, w, g):
        if self.m is None:
            self.m = np.zeros_like(w)
            self.v = np.zeros_like(w)
        self.t += 1
        alpha = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        self.m += (1.0 - self.b1) * (g - self.m)
        self.v += (1.0 - self.b2) * (g * g - self.v)
        return w - alpha * self.m / (np.sqrt(self.v) + self.eps)


def lbfgs(opfunc, x, max_iter, lr, n_corr, tol_fun=np.finfo(float).eps, tol_x=1e-19,
          max_eval=None, log_fn=None):
    """Returns dict(x=returned x, x_model=last evaluated x, f_hist, n_eval, logs, final_loss)."""
    if max_iter == 0:
        return None                                            # :43-44
    max_eval = max_eval or max_iter * 1.25                     # :50
    x = np.array(x, dtype=np.float64)
    f, g = opfunc(x)                                           # :65
    x_model = x.copy()
    f_hist = [f]
    n_eval = 1
    logs = []
    final_loss = None
    if np.sum(np.abs(g)) <= tol_fun:                           # :72-76
        return dict(x=x, x_model=x_model, f_hist=f_hist, n_eval=n_eval, logs=logs,
                    final_loss=final_loss)
    n_iter = 0
    S, Y = [], []                                              # old_dirs (s), old_stps (y)
    Hdiag = 1.0
    d = t = g_old = f_old = None
    while n_iter < max_iter:                                   # :81
        n_iter += 1
        if n_iter == 1:                                        # :90-95
            d = -g
            S, Y = [], []
            Hdiag = 1.0
        else:
            y = g - g_old                                      # :98-100
            s = d * t
            ys = np.sum(y * s)
            if ys > 1e-10:                                     # :102-114
                if len(S) == n_corr:
                    del S[0]
                    del Y[0]
                S.append(s)
                Y.append(y)
                Hdiag = ys / np.sum(y * y)
            k = len(S)
            ro = [1.0 / np.sum(Y[i] * S[i]) for i in range(k)]  # :121-123
            al = [0.0] * k
            q = -g
            for i in range(k - 1, -1, -1):                     # :130-133
                al[i] = np.sum(S[i] * q) * ro[i]
                q = q - al[i] * Y[i]
            r = q * Hdiag                                      # :136
            for i in range(k):                                 # :137-139
                be_i = np.sum(Y[i] * r) * ro[i]
                r = r + (al[i] - be_i) * S[i]
            d = r
        g_old = g
        f_old = f
        gtd = np.sum(g * d)                                    # :151
        if gtd > -tol_x:                                       # :154-156
            break
        if n_iter == 1:                                        # :159-163
            t = min(1.0, 1.0 / np.sum(np.abs(g)))
        else:
            t = lr
        x = x + t * d                                          # :174
        ls_eval = 0
        if n_iter != max_iter:                                 # :176-182
            f, g = opfunc(x)
            x_model = x.copy()
            ls_eval = 1
            f_hist.append(f)
        n_eval += ls_eval
        if n_iter == max_iter:                                 # :192
            break
        if n_eval >= max_eval:                                 # :195
            break
        if np.sum(np.abs(g)) <= tol_fun:                       # :200-203
            break
        if np.sum(np.abs(d * t)) <= tol_x:                     # :206-209
            break
        if abs(f - f_old) < tol_x:                             # :212-215
            break
        logs.append((n_iter, f))                               # :217-218
        if log_fn is not None:
            log_fn(n_iter, f, True)
        if n_iter == max_iter - 1:                             # :223-224
            final_loss = f
    return dict(x=x, x_model=x_model, f_hist=f_hist, n_eval=n_eval, logs=logs,
                final_loss=final_loss)
