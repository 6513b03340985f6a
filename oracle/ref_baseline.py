#!/usr/bin/env python3
"""Tier-A CPU baseline: the REFERENCE's own 1d-burgers/inf_cont_burgers.py (packed unmodified into
oracle/_ref/reference_sources.tar.gz by oracle/make_ref.py, unpacked into a scratch directory for this run) executed
over the torch-CPU stand-in for the `tensorflow` module (tests/ref_shims), timed.

TEST INFRASTRUCTURE: called by bench.py's cpu_baseline leg (as a subprocess) and by nothing in the product path.

    python3 oracle/ref_baseline.py [--tf-epochs 100] [--nt-epochs 200] [--threads N]

Prints ONE JSON line: collocation-points/s = N_f x (#loss+grad evaluations) / wall time of NeuralNetwork.fit
(utils/neuralnetwork.py:138-149: the Adam loop then custom_lbfgs.lbfgs; data prep, model construction and
plotting are outside the timed call), the thread count, and the final relative L2 error the script's own
error() reports (inf_cont_burgers.py:114-116).
"""
import argparse
import contextlib
import io
import json
import os
import runpy
import shutil
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIMS = os.path.join(ROOT, "tests", "ref_shims")
sys.path.insert(0, ROOT)
from oracle import make_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tf-epochs", type=int, default=100)
    ap.add_argument("--nt-epochs", type=int, default=200)
    ap.add_argument("--threads", type=int, default=0, help="torch intra-op threads (0 = torch's default)")
    args = ap.parse_args()
    if not make_ref.staged():
        print(json.dumps({"error": "oracle/_ref is not built (python3 oracle/make_ref.py in the build container)"}))
        return 1
    REFDIR = make_ref.unpack(tempfile.mkdtemp(prefix="pinn_ref_"))
    import torch
    if args.threads > 0:
        torch.set_num_threads(args.threads)
    os.chdir(REFDIR)                              # the reference's paths are cwd-relative
    sys.path.insert(0, SHIMS)
    sys.path.insert(1, os.path.join(REFDIR, "utils"))
    sys.path.insert(2, os.path.join(REFDIR, "1d-burgers"))
    hp = {"N_u": 100, "N_f": 10000, "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1],
          "tf_epochs": args.tf_epochs, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None,
          "nt_epochs": args.nt_epochs, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10}
    hp_path = os.path.join(REFDIR, "_hp.json")
    with open(hp_path, "w") as f:
        json.dump(hp, f)
    import neuralnetwork                           # the reference's module
    timing = {}
    fit = neuralnetwork.NeuralNetwork.fit

    def timed_fit(self, *a, **k):
        t0 = time.perf_counter()
        r = fit(self, *a, **k)
        timing["fit_s"] = time.perf_counter() - t0
        return r
    neuralnetwork.NeuralNetwork.fit = timed_fit
    sys.argv = ["1d-burgers/inf_cont_burgers.py", hp_path]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path("1d-burgers/inf_cont_burgers.py", run_name="__main__")
    evals = args.tf_epochs + args.nt_epochs        # one loss+grad per Adam epoch; L-BFGS: 1 initial + maxIter-1
    out = {"value": hp["N_f"] * evals / timing["fit_s"], "unit": "collocation-points/s", "evals": evals,
           "fit_seconds": timing["fit_s"], "threads": torch.get_num_threads(), "host_cores": os.cpu_count(),
           "final_l2_error": float(g["error"]()), "tf_epochs": args.tf_epochs, "nt_epochs": args.nt_epochs}
    os.chdir(ROOT)
    shutil.rmtree(REFDIR, ignore_errors=True)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
