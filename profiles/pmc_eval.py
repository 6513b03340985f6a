"""A short run of the dominant kernels for the rocprofv3 --pmc passes (counter collection serialises every dispatch,
so the full bench is far too long under it): 20 loss+gradient evaluations of the headline workload (N_f = 10000,
N_u = 100, 8x20) in float32 (k_fused20m) and in float64 (k_fused20d), nothing else.

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o f -- python profiles/pmc_eval.py
    cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d DIR -o w -- python profiles/pmc_eval.py
    python profiles/summarize_pmc.py DIR/f_results.db DIR/w_results.db"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402

n_f = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, n_f, noise=0.0)
for dtype in ("f32", "f64"):
    eng = pinn_native.Engine(bench.LAYERS, r[11], r[10], pde="burgers", dtype=dtype)
    eng.set_collocation(r[9]); eng.set_data(r[7], r[8]); eng.set_pde_params(bench.NU)
    eng.set_weights(bench.canonical_weights())
    for _ in range(20):
        eng.loss_grad(want_grad=False)
    print(dtype, "path", eng.kernel_path(), "done", flush=True)
    eng.close()
