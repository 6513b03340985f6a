"""How does the reference's schedule behave when L-BFGS runs long (a converged comparison)?
float64 engine, 100 Adam epochs then nt L-BFGS iterations, initial kernels scaled by (1 + k 2^-52):
final error on the 25600-point grid, final loss, whether the loss ever exploded, the engine's done code."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, burgersutil, pinn_native  # noqa: E402
from diag_f32_lbfgs import member_weights  # noqa: E402

np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
data, grid = (r[9], r[7], r[8], r[11], r[10]), (r[5], r[6])
for nt in [int(a) for a in sys.argv[1:]] or [500, 1000, 2000, 3000, 5000]:
    for k in (0, 1, -1, 2, -2):
        eng = pinn_native.Engine(bench.LAYERS, data[3], data[4], pde="burgers", dtype="f64")
        eng.set_collocation(data[0]); eng.set_data(data[1], data[2]); eng.set_pde_params(bench.NU)
        eng.set_weights(member_weights(k, 2.0 ** -52))
        eng.adam_init(0.03, 0.9, 0.999, 1e-7)
        eng.adam_run(100, want_losses=False)
        eng.lbfgs_begin(nt, 0.8, 50, float(np.finfo(float).eps))
        done, worst, last = 0, 0.0, None
        while not done:
            _, ll, done = eng.lbfgs_run(500)
            if len(ll):
                worst, last = max(worst, float(np.max(ll))), float(ll[-1])
        print("nt=%5d k=%+d  final error %.6f  last loss %.4e  max loss %.3e  done %d" % (nt, k, eng.error_l2(*grid), last, worst, done), flush=True)
        eng.close()
