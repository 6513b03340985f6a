"""Per-step timing of the non-headline BASELINE configs on one MI355X (Adam steps, inputs resident):
cfg 3 identification (N_u = 10000), cfg 4 Schrodinger (N_f = 20000, 4x100), cfg 5 shard sizes of the
1D Burgers N_f = 1e6 run.   python profiles/time_configs.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
sys.path.insert(0, ROOT)
import bench
sys.path.insert(0, os.path.join(bench.PKG, "1dcomplex-schrodinger"))
import burgersutil, schrodingerutil, pinn_native
from oracle import init

def timeit(eng, n=50):
    """median of 7 blocks of n steps after 0.1 s of warm-up (one short block right after start-up sees the clock ramp:
    cfg 4 float64 read 459 us in a 4 ms block where bench.py's one-second leg reads 401)"""
    eng.loss_grad(); eng.adam_init(1e-3, 0.9, 0.999, 1e-7)
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        eng.adam_run(n, want_losses=False); eng.sync()
    blocks = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.adam_run(n, want_losses=False); eng.sync()
        blocks.append((time.perf_counter() - t0) / n)
    return sorted(blocks)[3]

# cfg 3: identification, N_u = 10000
np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 10000, noise=0.0)
X_u, u, ub, lb = r[7], r[8], r[9], r[10]
for dt in ("f32", "f64"):
    eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers_ide", dtype=dt)
    eng.set_data(X_u, u)
    w = np.concatenate([bench.canonical_weights(), [0.0, -6.0]]); eng.set_weights(w)
    s = timeit(eng)
    print("cfg3 identification %s path=%d: %.1f us/Adam step -> %.3g pts/s" % (dt, eng.kernel_path(), s * 1e6, 10000 / s)); eng.close()
# cfg 4: Schrodinger
np.random.seed(1234)
r = schrodingerutil.prep_data(os.path.join(bench.PKG, "1dcomplex-schrodinger", "data", "NLS.mat"), 50, 50, 20000, noise=0.0)
X_f, ub, lb, tb, x0, u0, v0, X0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17], r[18]
layers = [2, 100, 100, 100, 100, 2]
for dt in ("f32", "f64"):
    eng = pinn_native.Engine(layers, lb, ub, pde="schrodinger", dtype=dt)
    eng.set_collocation(X_f); eng.set_boundary(np.concatenate((0 * tb + lb[0], tb), 1), np.concatenate((0 * tb + ub[0], tb), 1))
    eng.set_data(X0, np.concatenate([u0, v0], 1)); eng.set_weights(init.glorot_flat(layers))
    s = timeit(eng, 25)
    print("cfg4 schrodinger %s path=%d: %.1f us/Adam step -> %.3g pts/s" % (dt, eng.kernel_path(), s * 1e6, 20000 / s)); eng.close()
# cfg 5 shape on one GPU: Burgers N_f = 125000 (the per-GPU shard of 1e6 over 8) and 1e6
for nf in (125000, 1000000):
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, nf, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    for dt in ("f32", "f64"):
        eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype=dt)
        eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU); eng.set_weights(bench.canonical_weights())
        s = timeit(eng, 20)
        print("cfg5 burgers %s N_f=%d path=%d: %.1f us/Adam step -> %.3g pts/s (%.1f TFLOP/s)" % (dt, nf, eng.kernel_path(), s * 1e6, nf / s, nf / s * 68640 / 1e12)); eng.close()
