"""Adam-step time of the continuous Burgers model for network shapes other than the reference's 8x20 (which kernel
family serves them and how fast), N_f = 10000, f32 and f64.   python profiles/time_shapes.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import burgersutil, pinn_native
from oracle import init

np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
for layers in ([2] + [20] * 8 + [1], [2] + [20] * 4 + [1], [2] + [20] * 6 + [1], [2] + [20] * 10 + [1], [2] + [32] * 4 + [1], [2] + [50] * 4 + [1],
               [2] + [64] * 6 + [1], [2] + [100] * 4 + [1], [2] + [128] * 3 + [1]):
    for dt in ("f32", "f64"):
        eng = pinn_native.Engine(layers, lb, ub, pde="burgers", dtype=dt)
        eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU); eng.set_weights(init.glorot_flat(layers))
        eng.adam_init(1e-3, 0.9, 0.999, 1e-7); eng.adam_run(5, want_losses=False); eng.sync()
        n = 50 if eng.kernel_path() else 10
        blocks = []
        for _ in range(7):      # median of 7 blocks (a single 2 ms block sees the clock ramp: 10x20 f32 read 37..43 us)
            t0 = time.perf_counter(); eng.adam_run(n, want_losses=False); eng.sync()
            blocks.append((time.perf_counter() - t0) / n)
        s = sorted(blocks)[3]
        mw = sum(a * b for a, b in zip(layers[:-1], layers[1:]))
        print("%-28s %s path=%d: %8.1f us/step  %.3g pts/s  %.2f TFLOP/s" % (
            "x".join(map(str, layers)), dt, eng.kernel_path(), s * 1e6, 10000 / s, 24.0 * mw * 10000 / s / 1e12))
        eng.close()

# steady state (many 16-point groups per workgroup): the shape-generic MFMA path at N_f = 200000
np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 200000, noise=0.0)
X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
for layers, dt in (([2] + [50] * 4 + [1], "f32"), ([2] + [64] * 6 + [1], "f32"), ([2] + [128] * 3 + [1], "f32"),
                   ([2] + [64] * 6 + [1], "f64"), ([2] + [100] * 4 + [1], "f64")):
    eng = pinn_native.Engine(layers, lb, ub, pde="burgers", dtype=dt)
    eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU); eng.set_weights(init.glorot_flat(layers))
    eng.adam_init(1e-3, 0.9, 0.999, 1e-7); eng.adam_run(3, want_losses=False); eng.sync()
    t0 = time.perf_counter(); eng.adam_run(5, want_losses=False); eng.sync()
    s = (time.perf_counter() - t0) / 5
    mw = sum(a * b for a, b in zip(layers[:-1], layers[1:]))
    print("N_f=200000 %-24s %s path=%d: %8.1f us/step  %.3g pts/s  %.2f TFLOP/s" % (
        "x".join(map(str, layers)), dt, eng.kernel_path(), s * 1e6, 200000 / s, 24.0 * mw * 200000 / s / 1e12))
    eng.close()
