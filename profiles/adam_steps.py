"""300 Adam steps of the headline workload (N_f = 10000, N_u = 100, 8x20, float32), nothing else: the run that
profiles/gaps_rocpd.py looks at."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402

np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
eng = pinn_native.Engine(bench.LAYERS, r[11], r[10], pde="burgers", dtype=sys.argv[1] if len(sys.argv) > 1 else "f32")
eng.set_collocation(r[9]); eng.set_data(r[7], r[8]); eng.set_pde_params(bench.NU)
eng.set_weights(bench.canonical_weights())
eng.adam_init(0.001, 0.9, 0.999, 1e-7)
import time  # noqa: E402
eng.adam_run(100, want_losses=False)
eng.sync()
t0 = time.perf_counter()
for _ in range(3):
    eng.adam_run(100, want_losses=False)
t1 = time.perf_counter()
eng.sync()
t2 = time.perf_counter()
print("host enqueue %.2f us per step, wall %.2f us per step (300 steps)" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
eng.close()
