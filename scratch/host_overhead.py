import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, burgersutil, pinn_native

def run(n_f):
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, n_f, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype="f32")
    eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU)
    w0 = bench.canonical_weights()
    eng.set_weights(w0); eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    eng.adam_run(50, want_losses=False); eng.sync()
    for rep in range(2):
        eng.set_weights(w0); eng.adam_init(0.03, 0.9, 0.999, 1e-7); eng.sync()
        t0 = time.perf_counter(); eng.adam_run(200, want_losses=False); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        print("N_f=%d adam 200: host issue %.1f us/step, total %.1f us/step" % (n_f, (t1-t0)/200*1e6, (t2-t0)/200*1e6))
        t0 = time.perf_counter(); eng.lbfgs_begin(200, 0.8, 50, 2.2e-16); t1 = time.perf_counter()
        it, ll, done = eng.lbfgs_run(200); t2 = time.perf_counter()
        print("N_f=%d lbfgs 200: begin %.1f us, run total %.1f us/iter  done=%d" % (n_f, (t1-t0)*1e6, (t2-t1)/200*1e6, done))
    eng.close()

run(1000)
run(10000)
