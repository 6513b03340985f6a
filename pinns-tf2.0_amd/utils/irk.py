"""Implicit Runge-Kutta (Gauss-Legendre) Butcher tables for the discrete-time models (host side, numpy).

The reference loads `PINNs/Utilities/IRK_weights/Butcher_IRK<q>.txt` (1d-burgers/burgersutil.py:58-60, 91-94),
a file of the un-vendored maziarraissi/PINNs submodule holding the q-stage Gauss-Legendre collocation tableau
flattened as [A (q x q, row-major) | b (q) | c (q)].  That file is absent, so the table is generated here:
with c_j the Gauss nodes of [0, 1] and b_j their weights, the Lagrange basis expands in shifted Legendre
polynomials as  l_j(tau) = sum_k (2k+1) b_j P~_k(c_j) P~_k(tau)  (exact: Gauss quadrature integrates degree
2q-1), and  int_0^x P~_k = (P~_{k+1}(x) - P~_{k-1}(x)) / (2 (2k+1)),  so  A = I^T diag(2k+1) P diag(b)  --
one q x q matrix product, stable at q = 500 (row sums reproduce c to 1e-14).

If the reference's file *is* available, `load_butcher(q, utils_path)` reads it instead, so a user holding the
original tables gets exactly the reference's numbers.
"""
import os

import numpy as np
import scipy.special


def gauss_legendre_butcher(q):
    """(A [q,q], b [q], c [q]) in float64."""
    xi, wq = scipy.special.roots_legendre(q)
    c = 0.5 * (xi + 1.0)
    b = 0.5 * wq
    P = np.empty((q + 1, q))
    P[0] = 1.0
    P[1] = xi
    for k in range(1, q):
        P[k + 1] = ((2 * k + 1) * xi * P[k] - k * P[k - 1]) / (k + 1)
    integ = np.empty((q, q))                       # integ[k][i] = int_0^{c_i} P~_k
    integ[0] = c
    for k in range(1, q):
        integ[k] = (P[k + 1] - P[k - 1]) / (2.0 * (2 * k + 1))
    A = (integ.T * (2 * np.arange(q) + 1)) @ P[:q] * b[None, :]
    return A, b, c


def load_butcher(q, utils_path=None):
    """The column vector burgersutil.py:58 gets from np.loadtxt(..., ndmin=2), as float32 like the reference."""
    if utils_path is not None:
        path = os.path.join(utils_path, "IRK_weights", "Butcher_IRK%d.txt" % q)
        if os.path.exists(path):
            return np.float32(np.loadtxt(path, ndmin=2))
    A, b, c = gauss_legendre_butcher(q)
    return np.float32(np.concatenate([A.ravel(), b, c])[:, None])
