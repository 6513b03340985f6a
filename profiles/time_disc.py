"""Per-step timing of the discrete-time (IRK) Burgers models on one MI355X, inputs resident:
inference  [1,50,50,50,501], q = 500, N_n = 250 (+2 wall points)     (1d-burgers/inf_disc_burgers.py defaults)
identification [1,50,50,50,81], q = 81, N_0 = 199, N_1 = 201        (1d-burgers/ide_disc_burgers.py defaults)
and the CPU oracle (numpy f64) on the same evaluation.   python profiles/time_disc.py [--no-cpu]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import burgersutil, pinn_native
from oracle import disc, init

MAT = os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat")
NU = 0.01 / np.pi
lb, ub = np.array([-1.0]), np.array([1.0])


def flops(layers, sets):
    """2 FLOP/MAC x (3 channels x network MACs x (fwd + rev + dW) + IRK fwd + rev) per evaluation."""
    mw = sum(a * b for a, b in zip(layers[:-1], layers[1:]))
    total = 0
    for x, _, M in sets:
        total += len(x) * 2 * 3 * 3 * mw
        if M is not None:
            total += len(x) * 2 * 2 * M.shape[0] * M.shape[1]
    return total


def timeit(eng, n=200):
    eng.loss_grad(); eng.adam_init(1e-3, 0.9, 0.999, 1e-8); eng.adam_run(10, want_losses=False); eng.sync()
    t0 = time.perf_counter(); eng.adam_run(n, want_losses=False); eng.sync()
    adam = (time.perf_counter() - t0) / n
    eng.lbfgs_begin(n + 10, 0.8, 50, float(np.finfo(float).eps)); eng.lbfgs_run(10); eng.sync()
    t0 = time.perf_counter(); eng.lbfgs_run(n); eng.sync()
    return adam, (time.perf_counter() - t0) / n


def cases():
    np.random.seed(1234)
    r = burgersutil.prep_data(MAT, N_n=250, q=500, lb=lb, ub=ub, noise=0.0, idx_t_0=10, idx_t_1=90)
    layers = [1, 50, 50, 50, 501]
    yield "inference q=500", layers, disc.inference_sets(r[4], r[5], r[6], r[2], r[9]), False
    np.random.seed(1234)
    r = burgersutil.prep_data(MAT, N_0=199, N_1=201, lb=lb, ub=ub, noise=0.0, idx_t_0=10, idx_t_1=90)
    layers = [1, 50, 50, 50, r[7]]
    yield "identification q=%d" % r[7], layers, disc.identification_sets(r[0], r[1], r[2], r[3], r[6], r[9], r[10]), True


for name, layers, sets, ide in cases():
    w = init.glorot_flat(layers)
    if ide:
        w = np.concatenate([w, [0.0, -6.0]])
    fl = flops(layers, sets)
    npts = sum(len(s[0]) for s in sets)
    for dt in ("f32", "f64"):
        eng = pinn_native.Engine(layers, lb, ub, pde="burgers_disc_ide" if ide else "burgers_disc", dtype=dt)
        for s, (x, t, M) in enumerate(sets):
            eng.disc_set_stage(s, x, t, M)
        eng.set_pde_params(NU); eng.set_weights(w)
        adam, lbfgs = timeit(eng)
        print("disc %s %s: %d points, P = %d, %.3f GFLOP/eval: %.1f us/Adam step (%.2f TFLOP/s), %.1f us/L-BFGS iteration"
              % (name, dt, npts, w.size, fl / 1e9, adam * 1e6, fl / adam / 1e12, lbfgs * 1e6))
        eng.close()
    if "--no-cpu" not in sys.argv:
        disc.disc_loss_grad(w, layers, lb, ub, sets, nu=NU, identify=ide)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 5.0:
            disc.disc_loss_grad(w, layers, lb, ub, sets, nu=NU, identify=ide); n += 1
        cpu = (time.perf_counter() - t0) / n
        print("disc %s cpu oracle (numpy f64, %d threads): %.2f ms/eval" % (name, os.cpu_count(), cpu * 1e3))
