"""BASELINE configs[3] (1dcomplex-schrodinger/inf_cont_schrodinger.py: 2-100-100-100-100-2 net, N_f = 20000, N_0 = N_b = 50)
Adam-step time in float64 and float32; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
    python profiles/time_cfg4.py [f64|f32|both] [steps] [kernel path]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.path.insert(0, os.path.join(bench.PKG, "1dcomplex-schrodinger"))
import schrodingerutil, pinn_native
from oracle import init

which = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
force_path = int(sys.argv[3]) if len(sys.argv) > 3 else -1
np.random.seed(1234)
r = schrodingerutil.prep_data(os.path.join(bench.PKG, "1dcomplex-schrodinger", "data", "NLS.mat"), 50, 50, 20000, noise=0.0)
X_f, ub, lb, tb, x0, u0, v0, X0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17], r[18]
layers = [2, 100, 100, 100, 100, 2]
M_W = sum(a * b for a, b in zip(layers[:-1], layers[1:]))
flop = 24.0 * M_W * 20000 + 6.0 * M_W * (50 + 100)
for dt in (("f64", "f32") if which == "both" else (which,)):
    eng = pinn_native.Engine(layers, lb, ub, pde="schrodinger", dtype=dt)
    eng.set_collocation(X_f); eng.set_boundary(np.concatenate((0 * tb + lb[0], tb), 1), np.concatenate((0 * tb + ub[0], tb), 1))
    eng.set_data(X0, np.concatenate([u0, v0], 1)); eng.set_weights(init.glorot_flat(layers))
    if force_path >= 0:
        eng.set_kernel_path(force_path)
    eng.loss_grad(); eng.adam_init(1e-3, 0.9, 0.999, 1e-7); eng.adam_run(5, want_losses=False); eng.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); eng.adam_run(n, want_losses=False); eng.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    peak = 78.6 if dt == "f64" else 157.3
    print("cfg4 schrodinger %s path=%d: %.1f us/Adam step -> %.3g pts/s, %.1f TFLOP/s = %.1f %% of the %s peak" % (
        dt, eng.kernel_path(), best * 1e6, 20000 / best, flop / best / 1e12, 100 * flop / best / 1e12 / peak, dt), flush=True)
    eng.close()
