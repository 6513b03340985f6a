"""Adam-step time of the float32 width-20 kernels in the throughput regime (N_f = 40000 / 125000 / 10^6; more tiles than
CUs): k_fused20r (even layers stashed, odd recomputed, two workgroups per CU) vs k_fused20m, selected with
PINN_F32_RECOMPUTE=1 / 0.   python profiles/time_f32_regime.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import bench, burgersutil, pinn_native
for nf in (40000, 125000, 1000000):
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, nf, noise=0.0)
    eng = pinn_native.Engine(bench.LAYERS, r[11], r[10], pde="burgers", dtype="f32")
    eng.set_collocation(r[9]); eng.set_data(r[7], r[8]); eng.set_pde_params(bench.NU); eng.set_weights(bench.canonical_weights())
    eng.adam_init(1e-3, 0.9, 0.999, 1e-7); eng.adam_run(10, want_losses=False); eng.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); eng.adam_run(20, want_losses=False); eng.sync()
        best = min(best, (time.perf_counter() - t0) / 20)
    print("N_f=%%7d: %%8.1f us per Adam step, %%.3g points/s, %%.1f TFLOP/s = %%.1f %%%% of the FP32 peak" %% (
        nf, best * 1e6, nf / best, nf / best * 68640 / 1e12, nf / best * 68640 / 1e12 / 1.573), flush=True)
    eng.close()
''' % {"root": ROOT}
for name, flag in (("k_fused20r: recompute, two workgroups per CU", "1"), ("k_fused20m: full stash, one workgroup per CU", "0")):
    print("== " + name, flush=True)
    print(subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PINN_F32_RECOMPUTE=flag), capture_output=True, text=True).stdout, flush=True)
