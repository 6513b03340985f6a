"""GPU tests (pytest -m gpu) of the ticketed optimiser loops (include/pinn_hip.h: pinn_adam_enqueue / _collect, pinn_lbfgs_enqueue /
_collect, ABI v6) and of NeuralNetwork.fit logging one chunk behind the GPU (utils/neuralnetwork.py _pipelined).

What the reference does at this point: it formats a progress line every log_frequency epochs from a loss it has just
synchronised on (utils/neuralnetwork.py:105-109, utils/logger.py:45-51).  The pipelined loops must print the same lines and
end at bit-identical weights -- they launch the same kernels on the same chunk sizes -- while never idling the device."""
import contextlib
import io
import re

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

NU = 0.01 / np.pi
LAYERS = [2] + [20] * 8 + [1]


def _engine(burgers_sets, dtype="f64", N_u=100, N_f=10000):
    from pinn_native import Engine
    r = burgers_sets(N_u, N_f)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    eng = Engine(LAYERS, lb, ub, pde="burgers", dtype=dtype)
    eng.set_collocation(X_f); eng.set_data(X_u, u); eng.set_pde_params(NU)
    eng.set_weights(np.load(golden("burgers_eval.npz"))["w0"])
    eng.adam_init(0.03, 0.9, 0.999, 1e-7)
    return eng


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_enqueued_chunks_equal_the_synchronous_loops_bit_for_bit(burgers_sets, dtype):
    a, b = _engine(burgers_sets, dtype), _engine(burgers_sets, dtype)
    # Adam: 1 + 10 + 10 + 4 steps, synchronous vs two chunks in flight
    want = np.concatenate([a.adam_run(n) for n in (1, 10, 10, 4)])
    t1, t2 = b.adam_enqueue(1), b.adam_enqueue(10)
    got = [b.adam_collect(t1)]
    t3 = b.adam_enqueue(10)
    got.append(b.adam_collect(t2))
    t4 = b.adam_enqueue(4)
    got += [b.adam_collect(t3), b.adam_collect(t4)]
    assert np.array_equal(np.concatenate(got), want)
    assert np.array_equal(a.get_weights(), b.get_weights())
    # L-BFGS: 37 iterations in chunks of 10, synchronous vs one chunk ahead (done is seen one chunk late)
    eps = float(np.finfo(float).eps)
    a.lbfgs_begin(37, 0.8, 50, eps)
    its_a, ls_a, done = [], [], 0
    while not done:
        i, l, done = a.lbfgs_run(10)
        its_a.append(i); ls_a.append(l)
    b.lbfgs_begin(37, 0.8, 50, eps)
    its_b, ls_b, done, queue = [], [], 0, []
    while not done:
        queue.append(b.lbfgs_enqueue(10))
        if len(queue) > 1:
            i, l, done = b.lbfgs_collect(queue.pop(0))
            its_b.append(i); ls_b.append(l)
    for t in queue:
        i, l, d = b.lbfgs_collect(t)
        its_b.append(i); ls_b.append(l)
        assert d == 1
    assert np.array_equal(np.concatenate(its_a), np.concatenate(its_b)) and np.concatenate(its_a)[-1] == 36
    assert np.array_equal(np.concatenate(ls_a), np.concatenate(ls_b))
    assert np.array_equal(a.get_weights(), b.get_weights()) and np.array_equal(a.lbfgs_x(), b.lbfgs_x())
    a.close(); b.close()


def test_ticket_rules(burgers_sets):
    from pinn_native import PinnNativeError
    eng = _engine(burgers_sets, N_u=64, N_f=2048)
    t = [eng.adam_enqueue(2) for _ in range(4)]
    with pytest.raises(PinnNativeError, match="in flight already"):          # at most four chunks
        eng.adam_enqueue(2)
    with pytest.raises(PinnNativeError, match="not the oldest"):             # collected in the order they were issued
        eng.adam_collect(t[1])
    with pytest.raises(PinnNativeError, match="in flight"):                  # the synchronous call refuses to overtake them
        eng.adam_run(1)
    assert all(len(eng.adam_collect(k)) == 2 for k in t)
    assert len(eng.adam_run(3)) == 3
    # a chunk larger than the loss ring while another one is in flight: the ring grows, the earlier ticket stays collectable
    ref = _engine(burgers_sets, N_u=64, N_f=2048)
    ref.adam_run(3 + 8)
    want = np.concatenate([ref.adam_run(1), ref.adam_run(50)])
    t1, t2 = eng.adam_enqueue(1), eng.adam_enqueue(50)
    assert np.array_equal(np.concatenate([eng.adam_collect(t1), eng.adam_collect(t2)]), want)
    ref.close()
    # a restart drops what is in flight: lbfgs_begin with a chunk outstanding, then a clean run
    eps = float(np.finfo(float).eps)
    eng.lbfgs_begin(12, 0.8, 50, eps)
    eng.lbfgs_enqueue(5)
    eng.lbfgs_begin(12, 0.8, 50, eps)
    its, losses, done = eng.lbfgs_run(12)
    assert done == 1 and list(its) == list(range(1, 12)) and np.all(np.isfinite(losses))
    eng.close()


LINE = re.compile(r"^(tf_epoch|nt_epoch) =\s+(\d+)\s+elapsed = \S+ \(\+\S+\)  loss = (\S+)  ")


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_fit_logs_one_chunk_behind_and_ends_at_the_same_weights(burgers_sets, monkeypatch, dtype):
    """the drop-in class on the reference's default schedule: async_log (default) vs hp["async_log"] = False"""
    import importlib
    import sys
    import neuralnetwork
    monkeypatch.setattr(sys, "argv", ["inf_cont_burgers.py"])       # the script reads an hp file from argv[1]
    mod = importlib.import_module("inf_cont_burgers")
    r = burgers_sets(100, 10000)
    X_star, u_star, X_u, u, X_f, ub, lb = r[5], r[6], r[7], r[8], r[9], r[10], r[11]
    out = {}
    for mode in (True, False):
        hp = dict(mod.hp, dtype=dtype, async_log=mode, nt_guard=0.0)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            neuralnetwork.set_seed(1234)
            logger = mod.Logger(hp)
            pinn = mod.BurgersInformedNN(hp, logger, X_f, ub, lb, nu=NU)
            assert pinn._pipelined() is mode
            logger.set_error_fn(lambda: pinn.error_l2(X_star, u_star))
            pinn.fit(X_u, u)
        rows = [(m.group(1), int(m.group(2)), m.group(3)) for m in map(LINE.match, buf.getvalue().splitlines()) if m]
        out[mode] = (rows, pinn.get_weights(), buf.getvalue().split("error = ")[1].split()[0])
        pinn._engine.close()
    assert out[True][0] == out[False][0] and len(out[True][0]) == 10 + 19          # every printed digit of every line
    assert np.array_equal(out[True][1], out[False][1]) and out[True][2] == out[False][2]


def test_weight_snapshots_are_device_side_copies(burgers_sets):
    eng = _engine(burgers_sets, N_u=64, N_f=2048)
    w0 = eng.get_weights()
    eng.weights_snapshot(2)
    eng.adam_run(5)
    w5 = eng.get_weights()
    eng.weights_snapshot(0)
    assert not np.array_equal(w0, w5)
    eng.weights_restore(2)
    assert np.array_equal(eng.get_weights(), w0)
    l_a = eng.loss_grad()[0]                                # the compute-dtype mirror followed the restore
    eng.set_weights(w0)
    assert eng.loss_grad()[0] == l_a
    eng.weights_restore(0)
    assert np.array_equal(eng.get_weights(), w5)
    from pinn_native import PinnNativeError
    with pytest.raises(PinnNativeError, match="no snapshot in slot 3"):
        eng.weights_restore(3)
    with pytest.raises(PinnNativeError, match="outside 0..3"):
        eng.weights_snapshot(4)
    eng.close()


@pytest.mark.parametrize("guard", [1e3, 0.5])
def test_guarded_fit_one_chunk_behind_equals_the_synchronous_guard(burgers_sets, monkeypatch, guard):
    """float32 with the restart guard (its default 1e3 -- no restart on this schedule -- and an absurd 0.5 that refuses every
    chunk whose loss has not halved, i.e. spends all five restarts at once, each with a halved step): the pipelined loop (device-side snapshots) takes the decisions of the synchronous one
    (host copies): same restarts, same lines, bit-identical final weights"""
    import importlib
    import sys
    import neuralnetwork
    monkeypatch.setattr(sys, "argv", ["inf_cont_burgers.py"])
    mod = importlib.import_module("inf_cont_burgers")
    r = burgers_sets(100, 10000)
    X_star, u_star, X_u, u, X_f, ub, lb = r[5], r[6], r[7], r[8], r[9], r[10], r[11]
    out = {}
    for mode in (True, False):
        hp = dict(mod.hp, dtype="f32", async_log=mode, nt_guard=guard, nt_epochs=120)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            neuralnetwork.set_seed(1234)
            logger = mod.Logger(hp)
            pinn = mod.BurgersInformedNN(hp, logger, X_f, ub, lb, nu=NU)
            logger.set_error_fn(lambda: pinn.error_l2(X_star, u_star))
            pinn.fit(X_u, u)
        rows = [(m.group(1), int(m.group(2)), m.group(3)) for m in map(LINE.match, buf.getvalue().splitlines()) if m]
        out[mode] = (rows, pinn.get_weights(), list(pinn.nt_restarts))
        pinn._engine.close()
    assert out[True][2] == out[False][2]
    assert (len(out[True][2]) > 0) == (guard < 2)                   # the hair trigger fires, the default does not
    assert out[True][0] == out[False][0]
    assert np.array_equal(out[True][1], out[False][1])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_chunk_sizes_and_depths_equal_the_synchronous_loops(burgers_sets, seed):
    """chunk sizes 1..60 (past the first ring size of 16 steps), one to four chunks in flight, Adam then L-BFGS: every loss,
    every logged iteration and the final weights equal the synchronous calls on the same chunk sizes"""
    rs = np.random.RandomState(seed)
    sizes = [int(v) for v in rs.randint(1, 61, size=9)]
    a, b = _engine(burgers_sets, N_u=64, N_f=2048), _engine(burgers_sets, N_u=64, N_f=2048)
    want = np.concatenate([a.adam_run(n) for n in sizes])
    got, queue = [], []
    for n in sizes:
        queue.append(b.adam_enqueue(n))
        while len(queue) > int(rs.randint(0, 4)):                  # keep 0..3 chunks in flight behind this one
            got.append(b.adam_collect(queue.pop(0)))
    got += [b.adam_collect(t) for t in queue]
    assert np.array_equal(np.concatenate(got), want) and np.array_equal(a.get_weights(), b.get_weights())
    eps, total = float(np.finfo(float).eps), sum(sizes) // 2
    a.lbfgs_begin(total, 0.8, 50, eps); b.lbfgs_begin(total, 0.8, 50, eps)
    ia, la, done, k = [], [], 0, 0
    while not done:
        i, l, done = a.lbfgs_run(sizes[k % len(sizes)]); k += 1
        ia.append(i); la.append(l)
    ib, lb_, done, k, queue = [], [], 0, 0, []
    while not done:
        queue.append(b.lbfgs_enqueue(sizes[k % len(sizes)])); k += 1
        if len(queue) > int(rs.randint(0, 3)):
            i, l, done = b.lbfgs_collect(queue.pop(0))
            ib.append(i); lb_.append(l)
    for t in queue:
        i, l, _ = b.lbfgs_collect(t)
        ib.append(i); lb_.append(l)
    assert np.array_equal(np.concatenate(ia), np.concatenate(ib)) and np.array_equal(np.concatenate(la), np.concatenate(lb_))
    assert np.array_equal(a.get_weights(), b.get_weights()) and np.array_equal(a.lbfgs_x(), b.lbfgs_x())
    a.close(); b.close()


def test_schrodinger_per_evaluation_lines_one_chunk_behind(schrodinger_sets, monkeypatch):
    """the Schrodinger class prints mse_0 / mse_b / mse_f on EVERY evaluation (1dcomplex-schrodinger/inf_cont_schrodinger.py:128):
    its chunks travel as pinn_adam_enqueue_terms; stdout (progress lines and the per-evaluation lines, in order) and the final
    weights equal the synchronous run's"""
    import os
    import runpy
    import sys
    import neuralnetwork
    from conftest import PKG
    from logger import Logger
    monkeypatch.setattr(sys, "argv", ["inf_cont_schrodinger.py"])
    g = runpy.run_path(os.path.join(PKG, "1dcomplex-schrodinger", "inf_cont_schrodinger.py"), run_name="schro")
    r = schrodinger_sets(50, 50, 20000)
    X_f, ub, lb, tb, x0, u0, v0 = r[11], r[12], r[13], r[14], r[15], r[16], r[17]
    out = {}
    for mode in (True, False):
        hp = dict(g["hp"], tf_epochs=23, log_frequency=5, async_log=mode)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            neuralnetwork.set_seed(1234)
            pinn = g["SchrodingerInformedNN"](hp, Logger(hp), X_f, tb, ub, lb)
            assert pinn._pipelined() is mode
            pinn.logger.set_error_fn(lambda: 0.5)
            pinn.fit(x0, np.concatenate([u0, v0], 1))
        strip = re.compile(r"elapsed = \S+ \(\+\S+\)")
        lines = [strip.sub("", l) for l in buf.getvalue().splitlines() if l.startswith(("tf_epoch", "mse_0"))]
        out[mode] = (lines, pinn.get_weights())
        pinn._engine.close()
    assert out[True][0] == out[False][0]
    assert sum(l.startswith("mse_0") for l in out[True][0]) == 23 and sum(l.startswith("tf_epoch") for l in out[True][0]) == 5
    assert np.array_equal(out[True][1], out[False][1])
