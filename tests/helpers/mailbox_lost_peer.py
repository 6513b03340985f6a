"""Worker of tests/test_gpu_comm.py::test_mailbox_lost_peer_is_an_error_not_a_hang: two ranks on device 0 map each
other's mailboxes; rank 1 then never takes part in the exchange.  Rank 0's self-test must come back False after
its bounded wait (5 s) instead of hanging."""
import os
import sys
import time

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pinns-tf2.0_amd"))
import pinn_native  # noqa: E402


def main():
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    eng = pinn_native.Engine([2, 20, 20, 1], np.array([-1.0, 0.0]), np.array([1.0, 1.0]), pde="burgers", dtype="f32", device=0)
    handles = [None] * world
    dist.all_gather_object(handles, eng.comm_xgmi_export(world, rank))
    assert eng.comm_xgmi_attach(handles)
    dist.barrier()
    t0 = time.time()
    if rank == 0:
        ok = eng.comm_xgmi_selftest()          # the peer never answers
        dt = time.time() - t0
        assert ok is False and 3.0 < dt < 25.0, (ok, dt)
        print("LOST_PEER_OK %.1f s" % dt)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
