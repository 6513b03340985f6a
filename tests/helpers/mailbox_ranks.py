"""Worker of tests/test_gpu_comm.py::test_mailbox_allreduce_*: launched with torch.distributed.run, every rank on
device 0 (one GPU is enough: the mailboxes are hipIpc-shared between processes either way), no RCCL communicator
(RCCL refuses two ranks on one device).  Checks, in float64 and float32:
  * sharded loss/gradient through the mailbox all-reduce == single-process evaluation of the whole set
  * Adam (update fused behind the all-reduce) and L-BFGS trajectories == the single-process ones
  * replicas stay bit-identical across ranks."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "pinns-tf2.0_amd")
for p in (ROOT, PKG, os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers")):
    sys.path.insert(0, p)
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402
from pinn_native.parallel import attach_shards, init_engine_comm  # noqa: E402
from oracle import init  # noqa: E402

LAYERS = [2] + [20] * 8 + [1]
NU = 0.01 / np.pi


def main():
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 6000, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    w0 = init.glorot_flat(LAYERS)
    for dtype, tol in (("f64", 1e-12), ("f32", 2e-6)):
        eng = pinn_native.Engine(LAYERS, lb, ub, pde="burgers", dtype=dtype, device=0)
        attach_shards(eng, world, rank, X_f=X_f, X_u=X_u, u=u)
        eng.set_pde_params(NU)
        mode = init_engine_comm(eng, dist, world, rank, mailbox=True, rccl=False)
        assert mode == "mailbox" and eng.comm_mode() == "mailbox", mode
        # the families bench.py --gpus N launches: k_fused20d (path 7) in float64, k_fused20m (path 2) in float32
        assert eng.kernel_path() == (7 if dtype == "f64" else 2), eng.kernel_path()
        ref = pinn_native.Engine(LAYERS, lb, ub, pde="burgers", dtype=dtype, device=0)
        ref.set_collocation(X_f); ref.set_data(X_u, u); ref.set_pde_params(NU)
        for e in (eng, ref):
            e.set_weights(w0)
        loss, grad, _ = eng.loss_grad()
        lr, gr, _ = ref.loss_grad()
        assert abs(loss - lr) <= tol * abs(lr), (dtype, loss, lr)
        assert np.max(np.abs(grad - gr)) <= tol * 10 * np.max(np.abs(gr)), dtype
        for e in (eng, ref):
            e.adam_init(0.01, 0.9, 0.999, 1e-7)
        la, lb_ = eng.adam_run(25), ref.adam_run(25)
        assert np.max(np.abs(la - lb_) / lb_) <= (1e-9 if dtype == "f64" else 1e-3), (dtype, la[-1], lb_[-1])
        for e in (eng, ref):
            e.lbfgs_begin(15, 0.8, 50, float(np.finfo(float).eps))
        (_, l1, d1), (_, l2, d2) = eng.lbfgs_run(15), ref.lbfgs_run(15)
        assert d1 == d2 and len(l1) == len(l2)
        assert np.max(np.abs(l1 - l2) / l2) <= (1e-7 if dtype == "f64" else 5e-2), (dtype, l1, l2)
        eng.sync()
        w = eng.get_weights()
        ws = [None] * world
        dist.all_gather_object(ws, w.tobytes())
        assert all(b == ws[0] for b in ws), "replicas diverged"
        dist.barrier()
        eng.close(); ref.close()
    if rank == 0:
        print("MAILBOX_OK world=%d" % world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
