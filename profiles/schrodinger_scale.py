"""Wide (width-100) MFMA sweeps: forward / reverse kernel time vs number of 16-point groups per workgroup.
    python profiles/schrodinger_scale.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.path.insert(0, os.path.join(bench.PKG, "1dcomplex-schrodinger"))
import schrodingerutil, pinn_native
from oracle import init
layers = [2, 100, 100, 100, 100, 2]
for nf in (3946, 8042, 12138, 20000, 32000):
    np.random.seed(1234)
    r = schrodingerutil.prep_data(os.path.join(bench.PKG, "1dcomplex-schrodinger", "data", "NLS.mat"), 50, 50, nf, noise=0.0)
    X_f, ub, lb, tb, x0, u0, v0, X0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17], r[18]
    eng = pinn_native.Engine(layers, lb, ub, pde="schrodinger", dtype="f32")
    eng.set_collocation(X_f); eng.set_boundary(np.concatenate((0 * tb + lb[0], tb), 1), np.concatenate((0 * tb + ub[0], tb), 1))
    eng.set_data(X0, np.concatenate([u0, v0], 1)); eng.set_weights(init.glorot_flat(layers))
    for _ in range(3): eng.loss_grad()
    eng.timing_enable(20, 1)
    for _ in range(20): eng.loss_grad()
    t = eng.timing_read()
    ngrp = (nf + 150 + 63) // 64 * 4
    print("N_f=%6d groups=%5d (%.2f per WG): fwd %.1f us, bwd %.1f us  (empty bracket %.1f us)" % (
        nf, ngrp, ngrp / 256.0, (t["fwd_ms"] - t["empty_bracket_ms"]) * 1e3, (t["sweeps_ms"] - t["fwd_ms"] - t["empty_bracket_ms"]) * 1e3, t["empty_bracket_ms"] * 1e3))
    eng.close()
