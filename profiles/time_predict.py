"""What the post-training half of the path costs: self.model(X_star) on the 25 600-point grid, f_model
at those points and the device-side error metric, per call, wall clock; run under `rocprofv3 --kernel-trace --stats` for
the kernels' own durations (k_fwd20d / k_fwd20f / k_err_partial / k_err_final).   python profiles/time_predict.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402

np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
X_star, u_star = r[5], r[6]
for dtype in ("f64", "f32"):
    eng = pinn_native.Engine(bench.LAYERS, r[11], r[10], pde="burgers", dtype=dtype)
    eng.set_collocation(r[9]); eng.set_data(r[7], r[8]); eng.set_pde_params(bench.NU)
    eng.set_weights(bench.canonical_weights())
    for name, fn in (("predict(X_star) incl. 205 KB D2H", lambda: eng.predict(X_star)),
                     ("error_l2(X_star, u_star)", lambda: eng.error_l2(X_star, u_star)),
                     ("residual_at(X_star) incl. 205 KB D2H", lambda: eng.residual_at(X_star)),
                     ("numpy error from predict (rounds 1-2)", lambda: np.linalg.norm(u_star - eng.predict(X_star), 2) / np.linalg.norm(u_star, 2))):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(50):
            v = fn()
        dt = (time.perf_counter() - t0) / 50
        print("%s %-42s %8.1f us per call" % (dtype, name, dt * 1e6), flush=True)
    eng.close()
