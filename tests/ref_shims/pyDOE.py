"""Test-only stand-in for pyDOE==0.3.8 `lhs` (classic LHS from the global numpy RNG).

TEST INFRASTRUCTURE.  Restates pyDOE's `_lhsclassic`: one `rand(samples, n)`
draw, then one `permutation(samples)` per column (reference call sites:
1d-burgers/burgersutil.py:122, 1dcomplex-schrodinger/schrodingerutil.py:58).
"""
import numpy as np


def lhs(n, samples=None, criterion=None, iterations=None):
    if samples is None:
        samples = n
    cut = np.linspace(0, 1, samples + 1)
    u = np.random.rand(samples, n)
    a = cut[:samples]
    b = cut[1:samples + 1]
    rdpoints = np.zeros_like(u)
    for j in range(n):
        rdpoints[:, j] = u[:, j] * (b - a) + a
    H = np.zeros_like(rdpoints)
    for j in range(n):
        order = np.random.permutation(range(samples))
        H[:, j] = rdpoints[order, j]
    return H
