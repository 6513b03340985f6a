"""Adam step and L-BFGS iteration of the headline workload (N_f = 10000, 8x20), both arithmetics, for each library given:
the same-box A/B used for every change to the optimiser tail.
    python profiles/time_tail.py [name=path/to/libpinn_hip_variant.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import bench, burgersutil, pinn_native
np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
for dt in ("f64", "f32"):
    eng = pinn_native.Engine(bench.LAYERS, r[11], r[10], pde="burgers", dtype=dt)
    eng.set_collocation(r[9]); eng.set_data(r[7], r[8]); eng.set_pde_params(bench.NU)
    eng.set_weights(bench.canonical_weights())
    eng.adam_init(1e-3, 0.9, 0.999, 1e-7); eng.adam_run(50, want_losses=False); eng.sync()
    a = []
    for rep in range(7):
        t0 = time.perf_counter(); eng.adam_run(400, want_losses=False); eng.sync()
        a.append((time.perf_counter() - t0) / 400)
    l = []
    for rep in range(7):
        eng.set_weights(bench.canonical_weights()); eng.sync()
        t0 = time.perf_counter(); bench.run_steps(eng, 0, 300); eng.sync()
        l.append((time.perf_counter() - t0) / 300)
    print("%%s: Adam step %%.2f us   L-BFGS iteration %%.2f us   (medians of 7)" %% (dt, sorted(a)[3] * 1e6, sorted(l)[3] * 1e6), flush=True)
    eng.close()
''' % {"root": ROOT}
variants = [("product", None)] + [tuple(a.split("=", 1)) for a in sys.argv[1:]]
for rnd in range(2):                       # two rounds, interleaved: drift of the box shows up as a difference between rounds
    for name, lib in variants:
        env = dict(os.environ)
        if lib:
            env["PINN_HIP_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        print("== %s (round %d)" % (name, rnd), flush=True)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(res.stdout + res.stderr[-2000:], flush=True)
