#!/usr/bin/env python3
"""Per-wave phase timeline of the fused loss+grad kernel (profiling build, -DPINN_STAMPS).

    PINN_HIP_LIB=pinns-tf2.0_amd/pinn_native/libpinn_hip_stamps.so python profiles/stamps.py [f32|f64] [N_f]

Prints, for the median workgroup, the s_memtime ticks spent per phase (max over its 4 waves)
and the spread of workgroup start/end times across the grid."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets sys.path for the package)
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    n_f = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, n_f, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype=dtype)
    eng.set_collocation(X_f)
    eng.set_data(X_u, u)
    eng.set_pde_params(bench.NU)
    eng.set_weights(bench.canonical_weights())
    for _ in range(3):
        eng.loss_grad()
    st = eng.debug_stamps().astype(np.float64)          # [n_waves, 32]
    H = len(bench.LAYERS) - 2
    path = eng.kernel_path()
    n_st = 2 * H + 3 if path in (2, 7) else 2 * H + 2
    if len(sys.argv) > 3:
        eng.set_kernel_path(int(sys.argv[3]))
        path = eng.kernel_path()
        n_st = 2 * H + 3 if path in (2, 7) else 2 * H + 2
        st = eng.debug_stamps().astype(np.float64)
    st = st[:, :n_st].reshape(-1, 4, n_st)              # [wg, wave, stamp]
    t0 = st[:, :, 0].min()
    names = (["stage weights + dense0"] + ["fwd dense %d" % d for d in range(1, H)] +
             ["output + seeds + bwd dense %d" % H] + ["bwd dense %d" % d for d in range(H - 1, 0, -1)] +
             (["bwd dense 0", "epilogue: wave sums + gradient row"] if path in (2, 7) else ["bwd dense 0 + stores"]))
    dur = np.diff(st, axis=2)                           # [wg, wave, phase]
    wg_total = st[:, :, -1].max(axis=1) - st[:, :, 0].min(axis=1)
    print("# %s N_f=%d workgroups=%d  (ticks = shader cycles via s_memtime)" % (dtype, n_f, st.shape[0]))
    print("# workgroup duration ticks: min %.0f median %.0f max %.0f" % (wg_total.min(), np.median(wg_total), wg_total.max()))
    print("# first start -> last end: %.0f ticks; start spread %.0f ticks" % (
        st[:, :, -1].max() - t0, st[:, :, 0].min(axis=1).max() - t0))
    print("%-34s %10s %10s %10s" % ("phase", "median", "min", "max"))
    per = dur.max(axis=1)                               # slowest wave per wg
    for i, nme in enumerate(names):
        print("%-34s %10.0f %10.0f %10.0f" % (nme, np.median(per[:, i]), per[:, i].min(), per[:, i].max()))
    print("%-34s %10.0f" % ("sum of medians", np.median(per, axis=0).sum()))
    if path == 7:
        full = eng.debug_stamps().astype(np.float64).reshape(-1, 4, 32)
        if np.all(full[:, :, 20:28] > 0):                  # -DPINN_STAMPS2: inside reverse layer 4 and forward layer 4
            def seg(a, b):
                return np.median((full[:, :, b] - full[:, :, a]).max(axis=1))
            r0, r1 = 2 * H + 1 - 5, 2 * H + 1 - 4           # end of reverse layer 5 = start of layer 4; its end
            print("# reverse layer 4: adjoints + rotations %.0f | phase sum (barrier) %.0f | reverse GEMV %.0f | dW in-groups %s | bias blocks %.0f" % (
                seg(r0, 20), seg(20, 21), seg(21, 22), " ".join("%.0f" % seg(22 + m, 23 + m) for m in range(5)), seg(27, r1)))
            print("# forward layer 4: GEMV %.0f | tanh + channels + stash %.0f" % (seg(4, 29), seg(29, 5)))
    if path == 7 and st.shape[0] != (n_f + 100 + 63) // 64:         # k_fused20dh: waves 0-2 = mains, wave 3 = helper
        print("# helper-wave variant (48-point tiles): mains (slowest of waves 0-2) | helper wave, medians over workgroups")
        pm, ph = dur[:, :3, :].max(axis=1), dur[:, 3, :]
        for i, nme in enumerate(names):
            print("%-34s %10.0f %10.0f" % (nme, np.median(pm[:, i]), np.median(ph[:, i])))
        full = eng.debug_stamps().astype(np.float64).reshape(-1, 4, 32)
        H_ = H
        def segm(a, b):
            return np.median((full[:, :3, b] - full[:, :3, a]).max(axis=1))
        def segh(a, b):
            return np.median(full[:, 3, b] - full[:, 3, a])
        prev = 2 * H_ + 1 - 5                            # end of layer 5 = start of layer 4
        print("# layer 4, mains: adjoints+channels %.0f | wait consumed %.0f | park+flag %.0f | read back %.0f | own dW %.0f | GEMV %.0f" % (
            segm(prev, 20), segm(20, 21), segm(21, 22), segm(22, 23), segm(23, 24), segm(24, 2 * H_ + 1 - 4)))
        print("# layer 4, helper: after layer 5 .. poll %.0f | wait ready(main 0) %.0f | main 0 %.0f | main 1 %.0f | main 2 %.0f | folds+stores %.0f" % (
            segh(prev, 20), segh(20, 21), segh(21, 22), segh(22, 23), segh(23, 24), segh(24, 2 * H_ + 1 - 4)))
        print("%-34s %10.0f %10.0f" % ("whole wave (first to last stamp)", np.median((st[:, :3, -1] - st[:, :3, 0]).max(axis=1)),
                                       np.median(st[:, 3, -1] - st[:, 3, 0])))
    if path == 2:
        full = eng.debug_stamps().astype(np.float64).reshape(-1, 4, 32)
        H_ = H
        def seg(a, b):
            return np.median((full[:, :, b] - full[:, :, a]).max(axis=1))
        print("# layer 4 detail (slowest wave per workgroup, median): ")
        print("#  fwd: reads issued..gemv start %.0f | gemv %.0f | Q write + own nonlinear %.0f | barrier wait %.0f | g4 tail+barrier %.0f" % (
            seg(1 + 3, 24), seg(24, 25), seg(25, 26), seg(26, 27), seg(27, 1 + 4)))
        print("#  prologue: entry..DMA issued %.0f | ..setup done %.0f | dense 0 %.0f | barrier %.0f | layer 1 reads+gemv %.0f | "
              "nonlinear+exchange %.0f | image wait + barrier %.0f" % (
                  seg(0, 19), seg(19, 1), seg(1, 22), seg(22, 23), seg(23, 20), seg(20, 21), seg(21, 2)))
        print("#  epilogue: wave sums + first/last layer rows %.0f | dW scatter %.0f" % (seg(2 * H_ + 1, 31), seg(31, 2 * H_ + 2)))
        print("#  bwd: phase A %.0f | barrier wait %.0f | phase B (gemv + dW) %.0f | Q exchange %.0f" % (
            seg(2 * H_ + 1 - 5, 28), seg(28, 29), seg(29, 30), seg(30, 2 * H_ + 1 - 4)))
    eng.close()


if __name__ == "__main__":
    main()
