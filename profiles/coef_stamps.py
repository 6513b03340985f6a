"""(profiling build) s_memtime phase timeline of k_lbc_coef:
    PINN_HIP_LIB=pinns-tf2.0_amd/pinn_native/libpinn_hip_stamps.so python profiles/coef_stamps.py"""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, burgersutil, pinn_native
np.random.seed(1234)
r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000, noise=0.0)
X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype="f32")
eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(bench.NU)
eng.set_weights(bench.canonical_weights())
eng.lbfgs_begin(200, 0.8, 50, 2.2e-16)
lib = pinn_native.load()
for n in (5, 60, 60):
    eng.lbfgs_run(n)
    buf = (ctypes.c_longlong * 16)()
    assert lib.pinn_debug_coef_stamps(buf) == 0
    st = np.array(buf[:7], dtype=np.int64)
    print("after %3d more iterations: phase ticks" % n, np.diff(st), "total", st[6] - st[0])
print("phases: loads | lds-stage+sync | post+candidate+sync | prep | backward loop | forward loop | tail")
