import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.path.insert(0, os.path.join(bench.PKG, "1dcomplex-schrodinger"))
import schrodingerutil, pinn_native
from oracle import init
np.random.seed(1234)
r = schrodingerutil.prep_data(os.path.join(bench.PKG, "1dcomplex-schrodinger", "data", "NLS.mat"), 50, 50, 20000, noise=0.0)
X_f, ub, lb, tb, x0, u0, v0, X0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17], r[18]
layers = [2, 100, 100, 100, 100, 2]
eng = pinn_native.Engine(layers, lb, ub, pde="schrodinger", dtype="f32")
eng.set_collocation(X_f); eng.set_boundary(np.concatenate((0 * tb + lb[0], tb), 1), np.concatenate((0 * tb + ub[0], tb), 1))
eng.set_data(X0, np.concatenate([u0, v0], 1)); eng.set_weights(init.glorot_flat(layers))
eng.adam_init(0.05, 0.99, 0.999, 0.1)
eng.adam_run(5, want_losses=False); eng.sync()
t0 = time.perf_counter(); eng.adam_run(40, want_losses=False); eng.sync()
print("us/step", (time.perf_counter() - t0) / 40 * 1e6)
