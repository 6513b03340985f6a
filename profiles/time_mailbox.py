"""Cost of the mailbox all-reduce kernel itself: the N_f = 10000 bench workload on one GPU with (a) no communicator
(k_reduce_rows / k_reduce_adam), (b) a one-rank mailbox communicator (k_reduce_xgmi: same reduction + store to the
own mailbox + flag + wait + rank-ordered sum), (c) a one-rank RCCL communicator.  python profiles/time_mailbox.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import burgersutil, pinn_native

np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]


def run(mode):
    eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype="f32")
    eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU); eng.set_weights(bench.canonical_weights())
    if mode == "rccl":
        eng.comm_init(pinn_native.Engine.comm_unique_id(), 1, 0)
    if mode == "mailbox":
        h = eng.comm_xgmi_export(1, 0)
        assert eng.comm_xgmi_attach([h]) and eng.comm_xgmi_selftest()
        eng.comm_set_mode("mailbox")
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    eng.adam_run(20, want_losses=False); eng.lbfgs_begin(400, 0.8, 50, 2.2e-16); eng.lbfgs_run(20); eng.sync()
    eng.set_weights(bench.canonical_weights()); eng.adam_init(0.03, 0.9, 0.999, 1e-7); eng.sync()
    t0 = time.perf_counter(); eng.adam_run(200, want_losses=False); eng.sync(); ta = (time.perf_counter() - t0) / 200
    eng.lbfgs_begin(400, 0.8, 50, 2.2e-16); eng.sync()
    t0 = time.perf_counter(); eng.lbfgs_run(200); eng.sync(); tl = (time.perf_counter() - t0) / 200
    print("%-8s Adam step %.1f us, L-BFGS iteration %.1f us" % (mode, ta * 1e6, tl * 1e6))
    eng.close()


for m in ("none", "mailbox", "rccl"):
    run(m)
