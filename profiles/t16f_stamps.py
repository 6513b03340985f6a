"""(profiling build) s_memtime phase timeline of k_t16_fused, workgroup 0, its second 16-point group, every wave:
    python -c "import sys; sys.path.insert(0, 'pinns-tf2.0_amd'); import pinn_native; pinn_native.build(stamps=True)"
    PINN_HIP_LIB=pinns-tf2.0_amd/pinn_native/libpinn_hip_stamps.so python profiles/t16f_stamps.py
Prints, per wave, the shader cycles (s_memtime) spent in each phase of one group of
BASELINE configs[3] (Schrodinger 2-100x4-2, N_f = 20000, float64)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pinn_native  # noqa: E402

wl = bench.schrodinger_workload()
eng = bench.make_engine("f64", 0, wl, 1, 0)
lib = pinn_native.load()
for _ in range(3):
    eng.loss_grad()
buf = (ctypes.c_longlong * 512)()
assert lib.pinn_debug_t16f_stamps(buf) == 0, pinn_native.last_error()
st = np.array(buf[:], dtype=np.int64).reshape(8, 64)
names = {0: "start", 1: "dense 0"}
for l in (1, 2, 3):
    names[2 + 3 * (l - 1)] = "fwd%d barrier" % l
    names[3 + 3 * (l - 1)] = "fwd%d GEMM" % l
    names[4 + 3 * (l - 1)] = "fwd%d tanh epilogue" % l
names.update({14: "fwd end barrier", 15: "output layer", 16: "hand-over / barrier", 17: "seeds + barrier", 18: "dense H reverse"})
for i, d in enumerate((3, 2, 1)):
    b = 19 + 6 * i
    names[b] = "rev%d barrier" % d
    names[b + 1] = "rev%d dW tiles" % d
    names[b + 2] = "rev%d bias gradient" % d
    names[b + 3] = "rev%d adjoint GEMM" % d
    names[b + 4] = "rev%d barrier" % d
    names[b + 5] = "rev%d epilogue" % d
names.update({40: "barrier", 41: "dense 0 reverse", 42: "end barrier"})
idx = sorted(k for k in names if st[0, k] > 0)
print("%-22s" % "phase (cycles)" + "".join("  wave%d" % w for w in range(8)))
prev = idx[0]
tot = np.zeros(8, dtype=np.int64)
for k in idx[1:]:
    d = st[:, k] - st[:, prev]
    tot += d
    print("%-22s" % names[k] + "".join("%7d" % v for v in d))
    prev = k
print("%-22s" % "group total" + "".join("%7d" % v for v in tot))
k = st[:, 60:64]
print("whole kernel, workgroup 0 (cycles): zero the row %s | group loop %s | sums + scratch -> row %s"
      % (k[0, 1] - k[0, 0], k[0, 2] - k[0, 1], (k[:, 3] - k[:, 2]).max()))
print("group total = %d cycles; matrix instructions of the group per SIMD: ~1640 x 64 = 105 k cycles" % tot[0])
