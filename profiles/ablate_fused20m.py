"""Ablation of k_fused20m: Adam step time of the headline workload with one ingredient compiled out at a time
(-DPINN_ABL=n, see csrc/kernels_fused20m.h; the results of those builds are wrong by construction, only the time is read).

    # build (CPU): one shared object per variant
    for n in 0..7: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -DPINN_ABL=n -c csrc/engine.hip ...
    # run (GPU):
    python profiles/ablate_fused20m.py DIR_WITH_libpinn_hip_abl{n}.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pinns-tf2.0_amd", "pinn_native", "abl")
NAMES = {0: "product kernel", 1: "no dW matrix instructions", 2: "no group-4 chain / exchange / its barrier",
         3: "workgroup barriers -> LDS waits only", 4: "tanh -> one multiply", 5: "no AGPR stash traffic",
         6: "no adjoint arithmetic in phase A", 7: "no own-group GEMV matrix instructions"}
base = None
for n in range(8):
    lib = os.path.join(d, "libpinn_hip_abl%d.so" % n)
    if not os.path.exists(lib):
        continue
    env = dict(os.environ, PINN_HIP_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "step_time.py"), "f32", "300"], env=env,
                         capture_output=True, text=True).stdout
    us = [float(l.split(":")[2].split("us/step")[0]) for l in out.splitlines() if "us/step" in l]
    if not us:
        print("%d %-44s failed" % (n, NAMES[n])); continue
    t = min(us[1:]) if len(us) > 1 else us[0]
    base = t if n == 0 else base
    print("%d %-44s %6.2f us per Adam step  (%+.2f us)" % (n, NAMES[n], t, t - (base or t)), flush=True)
