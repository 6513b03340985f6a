"""Data-parallel plumbing: one process per GPU, collocation/data/boundary sets sharded in
contiguous blocks, gradient all-reduce by RCCL inside the engine.

`torch.distributed` (gloo) is used only for rendezvous: shipping the 128-byte RCCL unique id
from rank 0 and for host-side barriers / timing reductions.  The reference has no
distributed code at all; this is new.
"""
import os


def shard_bounds(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank` when n items are split over `world` ranks;
    the first n % world ranks get one extra item."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


class DataParallel(object):
    """What the drop-in classes (utils/neuralnetwork.py) need from the process group: who am I, which block of a point
    set is mine, and a digest comparison across the replicas.  torch.distributed (gloo) is rendezvous only; the
    gradient exchange itself happens inside the engine (RCCL / mailboxes, see init_engine_comm)."""

    def __init__(self, dist, world, rank, local_rank):
        self.dist, self.world, self.rank, self.local_rank = dist, int(world), int(rank), int(local_rank)

    @property
    def is_root(self):
        return self.rank == 0

    def shard(self, n):
        return shard_bounds(n, self.world, self.rank)

    def replicas_identical(self, w):
        """every rank holds bit-identical weights (they are never broadcast: they stay equal because every rank applies
        the same update to the same all-reduced gradient).  Collective."""
        import hashlib
        digests = [None] * self.world
        self.dist.all_gather_object(digests, hashlib.sha256(w.tobytes()).hexdigest())
        return all(d == digests[0] for d in digests)

    def gather_rows(self, block):
        """the ranks' row blocks [n_r, k], concatenated in rank order, on every rank.  Collective."""
        import numpy as np
        blocks = [None] * self.world
        self.dist.all_gather_object(blocks, np.ascontiguousarray(block))
        return np.concatenate(blocks, axis=0)

    def barrier(self):
        self.dist.barrier()


def is_root():
    """rank 0 of a torchrun launch, or a plain single-process run: the one process that prints, plots and saves"""
    world, rank, _ = env_world()
    return world == 1 or rank == 0


def from_env(enabled=True):
    """-> DataParallel when this process is one rank of `python -m torch.distributed.run --nproc-per-node N script.py`
    (WORLD_SIZE > 1), else None.  Joins the gloo process group on first use (MASTER_ADDR / MASTER_PORT from the
    launcher; 127.0.0.1 when unset)."""
    world, rank, local = env_world()
    if world <= 1 or not enabled:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return DataParallel(dist, world, rank, local)


def attach_shards(engine, world, rank, X_f=None, X_u=None, u=None, X_lb=None, X_ub=None):
    """Give `engine` this rank's block of every point set; mean() denominators stay global so
    that the per-rank partial sums add up to the single-process loss and gradient."""
    if X_f is not None:
        lo, hi = shard_bounds(len(X_f), world, rank)
        engine.set_collocation(X_f[lo:hi], n_total=len(X_f))
    if X_u is not None:
        lo, hi = shard_bounds(len(X_u), world, rank)
        engine.set_data(X_u[lo:hi], u[lo:hi], n_total=len(X_u))
    if X_lb is not None:
        lo, hi = shard_bounds(len(X_lb), world, rank)
        engine.set_boundary(X_lb[lo:hi], X_ub[lo:hi], n_total=len(X_lb))


def _comm_init_bounded(engine, unique_id, world, rank):
    """engine.comm_init (ncclCommInitRank, a collective) under a deadline -> None on success, else a one-line error.
    A rank whose peers never arrive (one of them failed before entering the call) would otherwise sit in it until the
    launcher's watchdog: after PINN_COMM_INIT_TIMEOUT_S (default 180) the rank reports a time-out into the gather that
    follows, like any other failure.  The call itself cannot be cancelled; it is left behind in a daemon thread and the
    engine never selects the RCCL mode afterwards."""
    import threading
    from . import PinnNativeError
    limit = float(os.environ.get("PINN_COMM_INIT_TIMEOUT_S", "180"))
    box = {}

    def run():
        try:
            engine.comm_init(unique_id, world, rank)
            box["err"] = None
        except PinnNativeError as e:
            box["err"] = str(e).splitlines()[0][:200]
        except Exception as e:                                   # anything else is this rank's failure as well
            box["err"] = "%s: %s" % (type(e).__name__, str(e)[:160])
    t = threading.Thread(target=run, name="pinn_comm_init", daemon=True)
    t.start()
    t.join(limit)
    if t.is_alive():
        return "ncclCommInitRank did not return within %.0f s on rank %d (a peer never joined?)" % (limit, rank)
    return box["err"]


def init_engine_comm(engine, dist, world, rank, mailbox=None, rccl=True, probe=False):
    """Create the communicator of `engine`.

    RCCL: rank 0 draws the unique id, torch.distributed (any backend; gloo in this repo) broadcasts it, every
    rank joins.  That is the default and what north_star names.  With PINN_COMM=auto (opt-in: the mailboxes have
    only ever run between processes sharing ONE device, never across an xGMI link) or mailbox=True, the
    peer-mapped mailbox all-reduce of csrc/kernels_xgmi.h is set up on top: handles are all-gathered, every rank
    attaches and self-tests, both implementations are timed on the node, and the mailboxes are switched on only
    if *every* rank reports success and they are not slower -- otherwise all ranks stay on RCCL.
    probe=True (bench.py) additionally times one exchange on the node under the RCCL default -> engine.comm_probe_us;
    the ranks first agree that every one of them holds a communicator, so that no rank enters the timed collective
    alone.  Returns the mode in use ("rccl" or "mailbox")."""
    from . import Engine, PinnNativeError
    policy = os.environ.get("PINN_COMM", "rccl").lower()     # rccl (default) | auto | mailbox-only (no RCCL communicator)
    if policy == "mailbox-only":
        rccl, mailbox = False, True
    engine.comm_fallback = None
    if rccl:
        box = [Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        mine = _comm_init_bounded(engine, box[0], world, rank)
        errors = [None] * world
        dist.all_gather_object(errors, mine)
        if any(errors):
            # ncclCommInitRank has never met N > 1 ranks on this project's hardware.  If it fails on EVERY rank
            # the run is not lost: the peer-mapped mailboxes (self-tested below before they are used) take the exchange and the
            # bench line says so.  A failure on some ranks only cannot be repaired here (the others hold a communicator).
            if not all(errors):
                raise RuntimeError("RCCL communicator: ranks disagree (%s)" % errors)
            if world > 1 and rank == 0:
                import sys
                print("pinn_native: RCCL communicator failed on every rank (%s); trying the mailbox all-reduce" % errors[0],
                      file=sys.stderr)
            engine.comm_fallback = "rccl failed: " + errors[0]
            rccl, mailbox = False, True
    if mailbox is None:
        mailbox = policy != "rccl"
    engine.comm_probe_us = None
    if not mailbox:
        if probe:
            # what one exchange of the [P+4] vector costs on this node (k_reduce_rows + ncclAllReduce, wall time per
            # iteration, MAX over ranks): the number the 8-GPU budget of DESIGN.md 6 needs.  The probe is a collective:
            # it runs only after every rank has said it is about to enter it.
            ready = [None] * world
            dist.all_gather_object(ready, engine.comm_mode() == "rccl" if hasattr(engine, "comm_mode") else True)
            if all(ready):
                try:
                    mine = engine.comm_benchmark("rccl")
                except PinnNativeError:
                    mine = None
                probes = [None] * world
                dist.all_gather_object(probes, mine)
                if all(p is not None for p in probes):
                    engine.comm_probe_us = {"rccl": max(probes)}
        return "rccl"
    try:
        mine = engine.comm_xgmi_export(world, rank)
    except PinnNativeError:
        mine = None
    def agree(flag):
        votes = [None] * world
        dist.all_gather_object(votes, bool(flag))
        return all(votes), votes

    handles = [None] * world
    dist.all_gather_object(handles, mine)
    ok = False
    if all(h is not None for h in handles):
        try:
            ok = engine.comm_xgmi_attach(handles)
        except PinnNativeError:
            ok = False
    mapped, verdicts = agree(ok)
    if mapped:                       # every rank can reach every mailbox: exchange test vectors (bounded waits)
        try:
            ok = engine.comm_xgmi_selftest()
        except PinnNativeError:
            ok = False
        mapped, verdicts = agree(ok)
    engine.comm_probe_us = None
    if mapped and rccl and policy == "auto":
        # both implementations work here: keep the one that is faster ON THIS NODE (collective timing of the
        # exchange alone; every rank takes the same decision from the gathered numbers)
        try:
            mine = (engine.comm_benchmark("rccl"), engine.comm_benchmark("mailbox"))
        except PinnNativeError:
            mine = None
        probes = [None] * world
        dist.all_gather_object(probes, mine)
        if all(p is not None for p in probes):
            t_rccl, t_box = max(p[0] for p in probes), max(p[1] for p in probes)
            engine.comm_probe_us = {"rccl": t_rccl, "mailbox": t_box}
            mapped = t_box <= t_rccl
        else:
            mapped = False
    if mapped:
        engine.comm_set_mode("mailbox")
        return "mailbox"
    if rccl:
        engine.comm_set_mode("rccl")
        return "rccl"
    if engine.comm_fallback:
        raise RuntimeError("no gradient exchange available: %s, and the mailbox all-reduce that was tried in its place is "
                           "unavailable too (per-rank verdicts %s)" % (engine.comm_fallback, verdicts))
    raise RuntimeError("mailbox all-reduce unavailable and no RCCL communicator requested: %s" % verdicts)
