"""Wall-clock time per Adam step of the headline workload outside bench.py (no sampling, no collectives):
python profiles/step_time.py [f32|f64] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
eng = pinn_native.Engine(bench.LAYERS, r[11], r[10], pde="burgers", dtype=dtype)
eng.set_collocation(r[9]); eng.set_data(r[7], r[8]); eng.set_pde_params(bench.NU)
eng.set_weights(bench.canonical_weights())
eng.adam_init(0.001, 0.9, 0.999, 1e-7)
eng.adam_run(50, want_losses=False); eng.sync()
for rep in range(4):
    t0 = time.perf_counter()
    eng.adam_run(K, want_losses=False)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print("%s: %d steps: %.2f us/step (host enqueue %.2f us/step)" % (dtype, K, (t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6), flush=True)
eng.close()
