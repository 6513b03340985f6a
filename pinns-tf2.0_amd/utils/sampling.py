"""Latin-hypercube sampling from numpy's legacy global RNG.

Stands in for `pyDOE.lhs(n, samples)` as the reference calls it
(1d-burgers/burgersutil.py:122, 1dcomplex-schrodinger/schrodingerutil.py:58): one
uniform draw of shape (samples, n), then one permutation per column.  Keeping the draw
order identical keeps the collocation set bit-identical to the reference's for the
same `np.random.seed`.
"""
import numpy as np


def lhs(n, samples):
    edges = np.linspace(0.0, 1.0, samples + 1)
    lo, width = edges[:-1], edges[1:] - edges[:-1]
    jitter = np.random.rand(samples, n)
    strata = jitter * width[:, None] + lo[:, None]
    out = np.empty_like(strata)
    for col in range(n):
        out[:, col] = strata[np.random.permutation(samples), col]
    return out
