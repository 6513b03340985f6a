"""GPU: RCCL path of the engine at world_size 1 (the only size a 1-GPU box allows), and the
sharding contract with the real kernels: evaluating two shards one after the other on the same
GPU with global denominators sums to the full-batch gradient."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
NU = 0.01 / np.pi


def _engine(burgers_sets, dtype):
    from pinn_native import Engine
    r = burgers_sets(64, 2048)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    eng = Engine([2] + [20] * 8 + [1], lb, ub, pde="burgers", dtype=dtype)
    eng.set_pde_params(NU)
    return eng, X_f, X_u, u


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rccl_world1_allreduce_is_identity(burgers_sets, dtype):
    from pinn_native import Engine
    g = np.load(golden("burgers_eval_small.npz"))
    eng, X_f, X_u, u = _engine(burgers_sets, dtype)
    eng.set_collocation(X_f)
    eng.set_data(X_u, u)
    eng.set_weights(g["w0"])
    l0, g0, _ = eng.loss_grad()
    eng.comm_init(Engine.comm_unique_id(), 1, 0)          # ncclCommInitRank, 1 rank
    l1, g1, _ = eng.loss_grad()                           # now goes through ncclAllReduce
    assert l0 == l1 and np.array_equal(g0, g1)
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    losses = eng.adam_run(3)
    assert np.all(np.isfinite(losses))
    eng.close()


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-12), ("f32", 5e-6)])
@pytest.mark.parametrize("path", [0, 1])
def test_shards_with_global_denominators_add_up(burgers_sets, dtype, tol, path):
    from pinn_native.parallel import attach_shards
    g = np.load(golden("burgers_eval_small.npz"))
    eng, X_f, X_u, u = _engine(burgers_sets, dtype)
    eng.set_kernel_path(path)
    eng.set_weights(g["w0"])
    eng.set_collocation(X_f)
    eng.set_data(X_u, u)
    l_full, g_full, _ = eng.loss_grad()
    tot_l, tot_g = 0.0, 0.0
    for rank in range(3):                                 # 3 ragged shards
        attach_shards(eng, 3, rank, X_f=X_f, X_u=X_u, u=u)
        lo, gr, _ = eng.loss_grad()
        tot_l, tot_g = tot_l + lo, tot_g + gr
    assert abs(tot_l - l_full) / l_full < tol
    assert np.max(np.abs(tot_g - g_full)) / np.max(np.abs(g_full)) < tol
    eng.close()


@pytest.mark.parametrize("world", [2, 4])
def test_mailbox_allreduce_between_processes(world):
    """csrc/kernels_xgmi.h end to end: `world` processes (all on device 0 -- the mailboxes are hipIpc-mapped across
    processes exactly as across GPUs), sharded Burgers set, no RCCL: loss/gradient, fused Adam, L-BFGS and replica
    bit-identity (tests/helpers/mailbox_ranks.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() + world) % 400
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "helpers", "mailbox_ranks.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "MAILBOX_OK world=%d" % world in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


def test_auto_policy_probes_both_implementations(burgers_sets, monkeypatch):
    """init_engine_comm, policy auto, on a one-rank world: RCCL communicator + mailboxes, self-test, then the exchange
    of both implementations is timed and the faster one kept; either way the evaluation is unchanged."""
    from pinn_native.parallel import init_engine_comm

    class OneRankWorld(object):          # the three torch.distributed calls the helper uses, for world size 1
        @staticmethod
        def broadcast_object_list(box, src=0):
            pass

        @staticmethod
        def all_gather_object(out, obj):
            out[0] = obj

    monkeypatch.setenv("PINN_COMM", "auto")              # opt-in; the default policy is RCCL
    eng, X_f, X_u, u = _engine(burgers_sets, "f32")
    eng.set_collocation(X_f); eng.set_data(X_u, u)
    g = np.load(golden("burgers_eval_small.npz"))
    eng.set_weights(g["w0"])
    ref = eng.loss_grad()
    mode = init_engine_comm(eng, OneRankWorld, 1, 0)
    assert mode in ("rccl", "mailbox") and eng.comm_mode() == mode
    probe = eng.comm_probe_us
    assert probe and probe["rccl"] > 0 and probe["mailbox"] > 0
    assert (mode == "mailbox") == (probe["mailbox"] <= probe["rccl"])
    got = eng.loss_grad()
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    eng.close()


def test_mailbox_lost_peer_is_an_error_not_a_hang():
    """a rank that stops taking part costs its peers one bounded wait, then a negative verdict / error code"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(here, "helpers", "mailbox_lost_peer.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "LOST_PEER_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
