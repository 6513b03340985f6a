"""Per-step timing of the float64 Burgers kernels on one MI355X (Adam steps, inputs resident): the register-stash
kernel k_fused20d (path 7) against the HBM-stash kernel k_fused20<double> (path 1), with the float32 kernel beside
them.   python profiles/time_f64.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import burgersutil, pinn_native


def timeit(eng, n):
    eng.loss_grad(); eng.adam_init(1e-3, 0.9, 0.999, 1e-7); eng.adam_run(200 if n > 20 else 5, want_losses=False); eng.sync()
    t0 = time.perf_counter(); eng.adam_run(n, want_losses=False); eng.sync()
    return (time.perf_counter() - t0) / n


for nf, n in ((10000, 300), (125000, 40), (1000000, 10)):
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, nf, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    for dt, path in (("f64", 7), ("f64", 1), ("f32", 2)):
        eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype=dt)
        eng.set_kernel_path(path)
        eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU); eng.set_weights(bench.canonical_weights())
        s = timeit(eng, n)
        peak = 78.6 if dt == "f64" else 157.3
        tf = nf / s * 68640 / 1e12
        print("burgers %s N_f=%-8d path=%d: %8.1f us/Adam step -> %.3g pts/s (%.1f TFLOP/s = %.1f%% of %s peak)" % (
            dt, nf, eng.kernel_path(), s * 1e6, nf / s, tf, 100 * tf / peak, dt), flush=True)
        eng.close()
