#!/usr/bin/env python3
"""The reference's own CPU path, timed IN THE BUILD CONTAINER, recorded as a fixture.

TEST INFRASTRUCTURE.  The reference is Python: it can be imported here (from /root/reference) but it does not travel to
the GPU box in any form, so it cannot be timed there.  This script runs the reference's unmodified
1d-burgers/inf_cont_burgers.py from where it lies under /root/reference over the torch-CPU stand-in for the
`tensorflow` module (tests/ref_shims), times NeuralNetwork.fit, and -- with --write-fixture -- stores the measurement as
data in tests/golden/cpu_reference_timing.json together with the host it was taken on.  bench.py prints that record
beside its live `cpu_baseline` (kind "port": oracle/fit.py timed on the GPU box's own host cores); nothing under
/root/reference is read at bench or test time.

    python3 oracle/ref_baseline.py [--tf-epochs 100] [--nt-epochs 200] [--threads N] [--write-fixture]

Prints ONE JSON line: collocation-points/s = N_f x (#loss+grad evaluations) / wall time of NeuralNetwork.fit
(utils/neuralnetwork.py:138-149: the Adam loop, then custom_lbfgs.lbfgs, every 10th epoch logged, the error function
called once by log_train_end; data prep, model construction and plotting are outside the timed call), the thread
count, and the final relative L2 error the script's own error() reports (inf_cont_burgers.py:114-116).
"""
import argparse
import contextlib
import io
import json
import os
import platform
import runpy
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIMS = os.path.join(ROOT, "tests", "ref_shims")
REF = "/root/reference"
FIXTURE = os.path.join(ROOT, "tests", "golden", "cpu_reference_timing.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tf-epochs", type=int, default=100)
    ap.add_argument("--nt-epochs", type=int, default=200)
    ap.add_argument("--threads", type=int, default=0, help="torch intra-op threads (0 = torch's default)")
    ap.add_argument("--write-fixture", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print(json.dumps({"error": "%s is absent: the reference can only be timed in the build container" % REF}))
        return 1
    import torch
    if args.threads > 0:
        torch.set_num_threads(args.threads)
    os.chdir(REF)                                  # the reference's paths are cwd-relative; nothing is written there
    sys.dont_write_bytecode = True
    sys.path.insert(0, SHIMS)
    sys.path.insert(1, os.path.join(REF, "utils"))
    sys.path.insert(2, os.path.join(REF, "1d-burgers"))
    hp = {"N_u": 100, "N_f": 10000, "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1],
          "tf_epochs": args.tf_epochs, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None,
          "nt_epochs": args.nt_epochs, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10}
    hp_path = "/tmp/_ref_baseline_hp.json"
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
           "final_l2_error": float(g["error"]()), "tf_epochs": args.tf_epochs, "nt_epochs": args.nt_epochs,
           "progress_lines": len([l for l in buf.getvalue().splitlines() if l.startswith(("tf_epoch", "nt_epoch"))]),
           "host": {"machine": platform.machine(), "cpu": _cpu_model(), "where": "build container (no GPU)"},
           "what": "the reference's 1d-burgers/inf_cont_burgers.py (unmodified, run from /root/reference) over the "
                   "torch-CPU float64 stand-in for tensorflow: wall time of NeuralNetwork.fit on the default schedule",
           "torch": torch.__version__}
    os.chdir(ROOT)
    if args.write_fixture:
        with open(FIXTURE, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))
    return 0


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    sys.exit(main())
