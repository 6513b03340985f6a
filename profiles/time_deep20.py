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
for layers in ([2] + [20] * 10 + [1], [2] + [20] * 12 + [1], [2] + [20] * 9 + [1]):
    for dt in ("f64",):
        for path in (1, 4, 0):
            eng = pinn_native.Engine(layers, lb, ub, pde="burgers", dtype=dt)
            eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU); eng.set_weights(init.glorot_flat(layers))
            try:
                eng.set_kernel_path(path)
            except Exception as e:
                print(layers[1:-1].__len__(), dt, path, "refused:", e); eng.close(); continue
            l0, g0, _ = eng.loss_grad()
            eng.adam_init(1e-3, 0.9, 0.999, 1e-7)
            t_end = time.perf_counter() + 0.1
            while time.perf_counter() < t_end:
                eng.adam_run(50, want_losses=False); eng.sync()
            b = []
            for _ in range(7):
                t0 = time.perf_counter(); eng.adam_run(50, want_losses=False); eng.sync(); b.append((time.perf_counter() - t0) / 50)
            print("%dx20 %s path=%d: %.1f us/step  loss %.12g" % (len(layers) - 2, dt, eng.kernel_path(), sorted(b)[3] * 1e6, l0))
            eng.close()
