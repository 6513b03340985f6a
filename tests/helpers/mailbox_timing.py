"""Rough cost of the mailbox exchange with real peers: `world` processes share device 0, each with its own
N_f = 10000 Burgers shard; Adam / L-BFGS step time with the mailbox communicator vs no communicator at all
(same processes, same GPU sharing).  torch.distributed.run --nproc-per-node W tests/helpers/mailbox_timing.py"""
import os
import sys
import time

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import burgersutil  # noqa: E402
import pinn_native  # noqa: E402
from pinn_native.parallel import attach_shards, init_engine_comm  # noqa: E402


def main():
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    np.random.seed(1234)
    r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, 10000 * world, noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    for mode in ("none", "mailbox"):
        eng = pinn_native.Engine(bench.LAYERS, lb, ub, pde="burgers", dtype="f32", device=0)
        attach_shards(eng, world, rank, X_f=X_f, X_u=X_u, u=u)
        eng.set_pde_params(bench.NU); eng.set_weights(bench.canonical_weights())
        if mode == "mailbox":
            assert init_engine_comm(eng, dist, world, rank, mailbox=True, rccl=False) == "mailbox"
        eng.adam_init(0.03, 0.9, 0.999, 1e-7)
        eng.adam_run(20, want_losses=False); eng.lbfgs_begin(400, 0.8, 50, 2.2e-16); eng.lbfgs_run(20); eng.sync()
        eng.set_weights(bench.canonical_weights()); eng.adam_init(0.03, 0.9, 0.999, 1e-7); eng.sync()
        dist.barrier()
        t0 = time.perf_counter(); eng.adam_run(300, want_losses=False); eng.sync(); ta = (time.perf_counter() - t0) / 300
        dist.barrier()
        eng.lbfgs_begin(400, 0.8, 50, 2.2e-16); eng.sync()
        dist.barrier()
        t0 = time.perf_counter(); eng.lbfgs_run(300); eng.sync(); tl = (time.perf_counter() - t0) / 300
        dist.barrier()
        if rank == 0:
            print("world=%d %-8s Adam step %.1f us, L-BFGS iteration %.1f us" % (world, mode, ta * 1e6, tl * 1e6))
        eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
