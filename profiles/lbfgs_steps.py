"""300 L-BFGS iterations of the headline workload (N_f = 10000, N_u = 100, 8x20), nothing else: wall time per
iteration, and the run that profiles/gaps_rocpd.py looks at under rocprofv3 --kernel-trace.
    python profiles/lbfgs_steps.py [f32|f64] [iterations]"""
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
for rep in range(3):
    eng.set_weights(bench.canonical_weights())
    eng.sync()
    t0 = time.perf_counter()
    bench.run_steps(eng, 0, K)
    eng.sync()
    print("%s: %d L-BFGS iterations: %.2f us/iteration" % (dtype, K, (time.perf_counter() - t0) / K * 1e6), flush=True)
eng.close()
