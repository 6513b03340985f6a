"""One rank of `python bench.py --gpus N` with the engine scripted (no GPU): what bench.self_launch starts in
tests/test_data_parallel_gloo.py instead of bench.py itself.  The stub stands in for pinn_native.Engine only; the
launcher, the gloo rendezvous, the sharding, the timing blocks and the JSON line are bench.py's own.
BENCH_STUB_FAIL_RANK=r : that rank dies with exit code 7 before the rendezvous (the others would wait for ever)
BENCH_STUB_HANG=1      : every rank sleeps (the launcher's time-out has to end the run)
BENCH_STUB_STALL_RANK=r: that rank's first optimiser call never returns (the rank's own watchdog has to end it)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "pinns-tf2.0_amd")


class StubEngine(object):
    """what bench.py touches of pinn_native.Engine; 'training' = a deterministic walk of the weight vector"""
    made = []

    def __init__(self, layers, lb, ub, pde="burgers", dtype="f32", device=0):
        self.dtype, self.pde, self.w = dtype, pde, np.zeros(3021)
        self.n_params = sum(a * b + b for a, b in zip(layers[:-1], layers[1:])) + (2 if pde == "burgers_ide" else 0)
        self.n_f = self.n_u = self.n_b = 0
        self.comm = None
        StubEngine.made.append(self)

    def set_collocation(self, X, n_total=None): self.n_f, self.n_f_total = len(X), n_total
    def set_data(self, X, u, n_total=None): self.n_u = len(X)
    def set_boundary(self, A, B, n_total=None): self.n_b = len(A)
    def set_pde_params(self, *p): pass
    def set_kernel_path(self, p): pass
    def kernel_path(self): return 4 if self.pde == "schrodinger" else 2 if self.dtype == "f32" else 7
    def set_weights(self, w): self.w = np.array(w, dtype=np.float64)
    def get_weights(self): return self.w.copy()
    def adam_init(self, *a): pass
    def adam_run(self, n, want_losses=True): self.w = self.w + 1e-3 * n
    def lbfgs_begin(self, n, *a): self.left = n
    def lbfgs_run(self, n): self.w = self.w - 1e-4 * self.left; return np.zeros(0, np.int32), np.zeros(0), 1
    def sync(self): pass
    def loss_grad(self, want_grad=True): return 0.28, (np.zeros(self.n_params) if want_grad else None), np.array([0.004, 0.276, 0.0])
    def timing_enable(self, n, every=1): pass
    def timing_read(self): return {"fwd_ms": 0.03, "sweeps_ms": 0.03, "eval_ms": 0.04, "empty_bracket_ms": 0.005, "kernel_exact": True, "n": 32}
    def predict(self, X): return np.zeros((len(X), 1))
    def error_l2(self, X, ref, modulus=False): return 0.25
    def comm_benchmark(self, mode, iters=200): return 21.0
    def comm_init(self, uid, world, rank): self.comm = (bytes(uid), world, rank)
    def comm_set_mode(self, m): pass
    def close(self): pass

    @staticmethod
    def comm_unique_id(): return b"u" * 128


def main():
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("BENCH_STUB_FAIL_RANK") == str(rank):
        sys.stderr.write("stub rank %d: simulated failure before the rendezvous\n" % rank)
        sys.exit(7)
    if os.environ.get("BENCH_STUB_HANG"):
        time.sleep(3600)
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.pop("PINN_COMM", None)
    os.environ.setdefault("PINN_BENCH_MIN_TIMED_MS", "20")
    import pinn_native
    import bench
    if os.environ.get("BENCH_STUB_STALL_RANK") == str(rank):
        StubEngine.adam_run = lambda self, n, want_losses=True: time.sleep(3600)
    pinn_native.Engine = StubEngine
    pinn_native.device_info = lambda d=0: {"name": "stub", "compute_units": 256, "hbm_bytes": 0}
    sys.argv = ["bench.py"] + sys.argv[1:]
    bench.main()


if __name__ == "__main__":
    main()
