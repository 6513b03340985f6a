#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own
Python sources (read-only, /root/reference) over the test shims in tests/ref_shims.

TEST INFRASTRUCTURE.  Runs only in the build container (the GPU box has no
/root/reference); the produced .npz/.json files are committed and are what the
CPU and GPU test-suites compare against.

    python3 tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json

What each fixture pins, and from which reference code:
  burgers_data.json      burgersutil.prep_data (1d-burgers/burgersutil.py:27-131), both branches
  schrodinger_data.json  schrodingerutil.prep_data (1dcomplex-schrodinger/schrodingerutil.py:21-61)
  burgers_eval.npz       BurgersInformedNN.loss/f_model + NeuralNetwork.get_loss_and_flat_grad
                         (1d-burgers/inf_cont_burgers.py:59-98, utils/neuralnetwork.py:91-103)
  burgers_eval_1e6.npz   BASELINE configs[4] (N_f = 10^6): the same get_loss_and_flat_grad closure evaluated on the eight
                         125 000-point blocks an 8-rank launch shards the set into (one reference model per block; the
                         loss is a mean, so the N_f = 10^6 value is the block average) at the canonical and a perturbed
                         weight vector, + the base class's plain-MSE closure (utils/neuralnetwork.py:51-52) for the data term
  burgers_adam.npz       NeuralNetwork.tf_optimization_step (utils/neuralnetwork.py:112-116)
  burgers_lbfgs.npz      custom_lbfgs.lbfgs driven by the PINN closure (utils/custom_lbfgs.py:39-236)
  lbfgs_kat.json         custom_lbfgs.lbfgs on a 6-D analytic function
  burgers_default_run.json  stdout of the unmodified script with default hp
  burgers_default_trace.npz the same run, every epoch's / iteration's loss at full precision (Logger wrapped from outside)
  schrodinger_eval.npz   SchrodingerInformedNN.loss at the canonical init (compat x0 broadcast)
  logger_bytes.json      utils/logger.py output format
  burgers_ide_eval.npz   identification variant (1d-burgers/ide_cont_burgers.py:47-172): the file does not
  burgers_ide_run.json   parse as shipped (IndentationError line 31); repair_ide_cont.py re-indents it --
                         whitespace only, checked -- and the repaired module runs unmodified over the shims
  burgers_disc_eval*.npz      discrete-time inference: BurgersInformedNN.U_0_model / loss /
                         get_loss_and_flat_grad / predict of 1d-burgers/inf_disc_burgers.py:57-129 and the
                         prep_data branch burgersutil.py:43-61, script run unmodified
  burgers_disc_ide_eval*.npz  discrete-time identification: 1d-burgers/ide_disc_burgers.py:52-196 and
                         burgersutil.py:78-98, script run unmodified except for three environment
                         adapters (see gen_burgers_disc): the Butcher table file of the absent PINNs
                         submodule, np.asscalar (removed from numpy), Logger(frequency=...) (older
                         Logger signature than utils/logger.py in the same tree)
"""
import contextlib
import hashlib
import io
import json
import os
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SHIMS = os.path.join(os.path.dirname(HERE), "ref_shims")
REF = "/root/reference"


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def run_reference_script(relpath, hp):
    """Execute a reference script as __main__ with an hp JSON, return its globals + stdout."""
    import tensorflow as tf
    tf._reset_for_new_process_like_state()
    hp_path = "/tmp/_golden_hp.json"
    with open(hp_path, "w") as f:
        json.dump(hp, f)
    old_argv = sys.argv
    sys.argv = [relpath, hp_path]
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            g = runpy.run_path(relpath, run_name="__main__")
    finally:
        sys.argv = old_argv
    return g, buf.getvalue()


def burgers_hp(**kw):
    hp = {"N_u": 100, "N_f": 10000,
          "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1],
          "tf_epochs": 0, "tf_lr": 0.03, "tf_b1": 0.9, "tf_eps": None,
          "nt_epochs": 0, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10}
    hp.update(kw)
    return hp


def gen_burgers_data():
    import burgersutil
    out = {}
    np.random.seed(1234)
    r = burgersutil.prep_data("1d-burgers/data/burgers_shock.mat", 100, 10000, noise=0.0)
    x, t, X, T, Exact_u, X_star, u_star, X_u, u, X_f, ub, lb = r
    out["inf"] = {
        "lb": lb.tolist(), "ub": ub.tolist(),
        "shapes": {"X_star": X_star.shape, "u_star": u_star.shape, "X_u": X_u.shape,
                   "u": u.shape, "X_f": X_f.shape},
        "X_f_first": X_f[0].tolist(), "X_f_last": X_f[-1].tolist(),
        "X_u_first": X_u[0].tolist(), "u_first": float(u[0, 0]),
        "sha": {"X_f": sha16(X_f), "X_u": sha16(X_u), "u": sha16(u),
                "X_star": sha16(X_star), "u_star": sha16(u_star)},
    }
    np.random.seed(1234)
    r = burgersutil.prep_data("1d-burgers/data/burgers_shock.mat", 64, 2048, noise=0.0)
    out["inf_small"] = {"N_u": 64, "N_f": 2048,
                        "sha": {"X_f": sha16(r[9]), "X_u": sha16(r[7]), "u": sha16(r[8])},
                        "X_f_first": r[9][0].tolist()}
    np.random.seed(1234)
    r = burgersutil.prep_data("1d-burgers/data/burgers_shock.mat", 10000, noise=0.0)
    out["ide"] = {"n_returns": len(r), "X_u_shape": r[7].shape, "X_u_first": r[7][0].tolist(),
                  "lb": r[10].tolist(), "ub": r[9].tolist(),
                  "sha": {"X_u": sha16(r[7]), "u": sha16(r[8])}}
    with open(os.path.join(HERE, "burgers_data.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("burgers_data.json", out["inf"]["sha"])


def gen_schrodinger_data():
    import schrodingerutil
    np.random.seed(1234)
    r = schrodingerutil.prep_data("1dcomplex-schrodinger/data/NLS.mat", 50, 50, 20000, noise=0.0)
    (x, t, X, T, Exact_u, Exact_v, Exact_h, X_star, u_star, v_star, h_star, X_f,
     ub, lb, tb, x0, u0, v0, X0, H0) = r
    out = {"lb": lb.tolist(), "ub": ub.tolist(),
           "shapes": {"X_star": X_star.shape, "X_f": X_f.shape, "x0": x0.shape, "tb": tb.shape,
                      "X0": X0.shape, "H0": H0.shape},
           "X_f_first": X_f[0].tolist(), "x0_first": float(x0[0, 0]), "tb_first": float(tb[0, 0]),
           "sha": {"X_f": sha16(X_f), "x0": sha16(x0), "tb": sha16(tb), "u0": sha16(u0),
                   "v0": sha16(v0), "X_star": sha16(X_star), "h_star": sha16(h_star),
                   "u_star": sha16(u_star), "v_star": sha16(v_star)}}
    with open(os.path.join(HERE, "schrodinger_data.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("schrodinger_data.json", out["sha"])


def gen_burgers_eval_adam_lbfgs():
    import tensorflow as tf
    from custom_lbfgs import lbfgs, Struct
    import custom_lbfgs
    for tag, hp in (("", burgers_hp()), ("_small", burgers_hp(N_u=64, N_f=2048))):
        g, _ = run_reference_script("1d-burgers/inf_cont_burgers.py", hp)
        pinn, X_u, u = g["pinn"], g["X_u_train"], g["u_train"]
        X_star, u_star, X_f = g["X_star"], g["u_star"], g["X_f"]
        w0 = pinn.get_weights().numpy()
        Xt, ut = pinn.tensor(X_u), pinn.tensor(u)
        closure = pinn.get_loss_and_flat_grad(Xt, ut)
        loss, grad = closure(tf.convert_to_tensor(w0))
        f0 = pinn.f_model().numpy()
        u_pred, _ = pinn.predict(X_star)
        mse_u = float(np.mean((u - pinn.model(Xt).numpy()) ** 2))
        err0 = float(np.linalg.norm(u_star - u_pred, 2) / np.linalg.norm(u_star, 2))
        np.savez_compressed(
            os.path.join(HERE, "burgers_eval%s.npz" % tag),
            hp=json.dumps(hp), w0=w0, loss=float(loss), grad=grad.numpy(),
            f_first=f0[:64, 0], f_sha=sha16(f0), u_pred_first=u_pred[:64, 0],
            u_pred_stride=u_pred[::257, 0], mse_u=mse_u, mse_f=float(loss) - mse_u,
            err0=err0, nu=float(pinn.nu), sha_w0=sha16(w0),
            sha_X_f=sha16(X_f), sha_X_u=sha16(X_u))
        print("burgers_eval%s: loss=%.17g |g|1=%.17g err0=%.17g" % (
            tag, float(loss), float(np.abs(grad.numpy()).sum()), err0))

        # ---- Adam trajectory (30 steps from w0) ----
        pinn.set_weights(tf.convert_to_tensor(w0))
        losses, snaps = [], {}
        for it in range(30):
            lv = pinn.tf_optimization_step(Xt, ut)
            losses.append(float(lv))
            if it + 1 in (1, 5, 30):
                snaps["w_after_%d" % (it + 1)] = pinn.get_weights().numpy()
        np.savez_compressed(os.path.join(HERE, "burgers_adam%s.npz" % tag),
                            hp=json.dumps(hp), losses=np.array(losses), **snaps)
        print("burgers_adam%s: loss[0,1,29] = %.10e %.10e %.10e" % (
            tag, losses[0], losses[1], losses[29]))

        # ---- L-BFGS trajectory (25 iterations from w0, reference driver) ----
        pinn.set_weights(tf.convert_to_tensor(w0))
        cfg = Struct()
        cfg.learningRate = hp["nt_lr"]
        cfg.maxIter = 25
        cfg.nCorrection = 6          # small history so the shift branch is exercised
        cfg.tolFun = 1.0 * np.finfo(float).eps
        logs = []
        ret = lbfgs(closure, pinn.get_weights(), cfg, Struct(), True,
                    lambda it, f, is_iter: logs.append((int(it), float(f))))
        x_ret, f_hist, n_eval = ret
        np.savez_compressed(
            os.path.join(HERE, "burgers_lbfgs%s.npz" % tag), hp=json.dumps(hp),
            max_iter=25, n_corr=6, lr=hp["nt_lr"],
            log_iters=np.array([l[0] for l in logs]), log_losses=np.array([l[1] for l in logs]),
            f_hist=np.array([float(f) for f in f_hist]), n_eval=int(n_eval),
            x_returned=x_ret.numpy(), w_model=pinn.get_weights().numpy(),
            final_loss_global=float(custom_lbfgs.final_loss))
        print("burgers_lbfgs%s: f_hist[-1]=%.10e n_eval=%d logs=%d" % (
            tag, float(f_hist[-1]), n_eval, len(logs)))


def perturbed_weights(w0):
    """a deterministic second evaluation point: every entry moved by up to 5 % of the largest kernel entry (biases leave 0)"""
    k = np.arange(w0.size, dtype=np.float64)
    return w0 + 0.05 * np.max(np.abs(w0)) * np.sin(1.0 + 0.37 * k)


def gen_burgers_eval_1e6():
    """BASELINE configs[4]: N_f = 10^6 (1d-burgers/inf_cont_burgers.py:59-90 on burgersutil.py:122's Latin hypercube).
    The reference evaluates mean(f^2) over all points in one tape; that graph does not fit this container's memory
    over the torch stand-in, and the loss is a mean: L = mse_u + (1/N) sum_c sum_{i in c} f_i^2 = sum_c (n_c/N) L_c with
    L_c = the reference's own loss of a model built on block c.  The blocks are the contiguous eight an 8-rank launch
    shards the set into (pinn_native.parallel.shard_bounds), so the fixture pins every rank's shard AND their sum.
    The classes come from the unmodified script (run once at its default size); the set is prep_data's with the
    script's seed.  Recorded per block: the closure's (loss, flat grad) at w0 (canonical init) and w1 (perturbed)."""
    import tensorflow as tf
    import burgersutil
    g, _ = run_reference_script("1d-burgers/inf_cont_burgers.py", burgers_hp())
    Cls, Base, Logger = g["BurgersInformedNN"], g["NeuralNetwork"], g["Logger"]
    hp = burgers_hp(N_f=1000000)
    np.random.seed(1234)                                   # the script's own seed (inf_cont_burgers.py:13)
    r = burgersutil.prep_data("1d-burgers/data/burgers_shock.mat", hp["N_u"], hp["N_f"], noise=0.0)
    X_u, u, X_f, ub, lb = r[7], r[8], r[9], r[10], r[11]
    w0 = g["pinn"].get_weights().numpy()                   # the canonical init (first model of the process, seed 1234)
    ws = {"w0": w0, "w1": perturbed_weights(w0)}
    N, R = X_f.shape[0], 8
    bounds = [(N * k // R, N * (k + 1) // R) for k in range(R)]
    out = {"bounds": np.array(bounds), "sha_X_f": sha16(X_f), "sha_X_u": sha16(X_u), "sha_w0": sha16(w0), "w1": ws["w1"],
           "X_f_first": X_f[:4], "X_f_last": X_f[-4:], "hp": json.dumps(hp), "nu": 0.01 / np.pi}
    with contextlib.redirect_stdout(io.StringIO()):
        logger = Logger(hp)
        for name, w in ws.items():
            losses, grads = [], []
            for lo, hi in bounds:
                pinn = Cls(hp, logger, X_f[lo:hi], ub, lb, nu=0.01 / np.pi)
                closure = pinn.get_loss_and_flat_grad(pinn.tensor(X_u), pinn.tensor(u))
                lv, gv = closure(tf.convert_to_tensor(w))
                losses.append(float(lv))
                grads.append(gv.numpy())
                del pinn, closure
            base = Base(hp, logger, ub, lb)                # the base class: loss = plain data misfit (:51-52)
            lv, gv = base.get_loss_and_flat_grad(base.tensor(X_u), base.tensor(u))(tf.convert_to_tensor(w))
            out["block_loss_" + name], out["block_grad_" + name] = np.array(losses), np.array(grads)
            out["mse_u_" + name], out["grad_mse_u_" + name] = float(lv), gv.numpy()
            wt = np.array([(hi - lo) / N for lo, hi in bounds])
            out["loss_" + name] = float(np.dot(wt, losses))
            out["grad_" + name] = (wt[:, None] * np.array(grads)).sum(0)
    np.savez_compressed(os.path.join(HERE, "burgers_eval_1e6.npz"), **out)
    print("burgers_eval_1e6: loss(w0)=%.17g mse_u=%.17g loss(w1)=%.17g |g(w0)|1=%.10e" % (
        out["loss_w0"], out["mse_u_w0"], out["loss_w1"], float(np.abs(out["grad_w0"]).sum())))


def gen_lbfgs_kat():
    import tensorflow as tf
    import custom_lbfgs
    from custom_lbfgs import lbfgs, Struct
    A = np.diag(np.arange(1.0, 7.0)) + 0.1 * np.ones((6, 6))
    b = np.arange(1.0, 7.0)
    args = []

    def opfunc(x):
        xv = x.numpy()
        args.append(xv.copy())
        f = 0.5 * xv @ A @ xv - b @ xv + 0.25 * np.sum(xv ** 4)
        gr = A @ xv - b + xv ** 3
        return tf.convert_to_tensor(f), tf.convert_to_tensor(gr)

    cfg = Struct()
    cfg.learningRate = 0.8
    cfg.maxIter = 8
    cfg.nCorrection = 3
    cfg.tolFun = 1.0 * np.finfo(float).eps
    logs = []
    x, f_hist, n_eval = lbfgs(opfunc, tf.convert_to_tensor(np.zeros(6)), cfg, Struct(), True,
                              lambda it, f, is_iter: logs.append((int(it), float(f))))
    cfg0 = Struct()
    cfg0.maxIter = 0
    none_ret = lbfgs(opfunc, tf.convert_to_tensor(np.zeros(6)), cfg0, Struct(), True, None)
    out = {"f_hist": [float(f) for f in f_hist], "logs": logs, "n_eval": int(n_eval),
           "x_returned": x.numpy().tolist(), "last_opfunc_arg": args[-1].tolist(),
           "n_opfunc_calls": len(args), "final_loss_global": float(custom_lbfgs.final_loss),
           "maxiter0_returns_none": none_ret is None, "struct_default": Struct().anything}
    with open(os.path.join(HERE, "lbfgs_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("lbfgs_kat:", out["f_hist"][-1], out["n_eval"])


def gen_default_run():
    hp = burgers_hp(tf_epochs=100, nt_epochs=200)
    g, out = run_reference_script("1d-burgers/inf_cont_burgers.py", hp)
    lines = [l for l in out.splitlines() if l.startswith(("tf_epoch", "nt_epoch", "Training finished"))]
    rec = {"hp": hp, "lines": lines, "final_error": float(g["error"]()),
           "w_final_sha": sha16(g["pinn"].get_weights().numpy())}
    np.save(os.path.join(HERE, "burgers_default_run_w_final.npy"), g["pinn"].get_weights().numpy())
    with open(os.path.join(HERE, "burgers_default_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("default run: final error %.6e" % rec["final_error"])


def gen_default_trace():
    """the default run again with the reference's Logger.log_train_epoch wrapped (from outside; the script and utils/
    are untouched): EVERY epoch's / iteration's loss at full precision instead of the 4 printed digits of every 10th --
    what `first L-BFGS iteration whose loss departs by more than 1e-6` is measured against"""
    sys.path.insert(1, os.path.join(REF, "utils"))
    import logger as ref_logger
    rows = []
    orig = ref_logger.Logger.log_train_epoch

    def spy(self, epoch, loss, custom="", is_iter=False):
        rows.append((int(epoch), float(loss), bool(is_iter)))
        return orig(self, epoch, loss, custom, is_iter)
    ref_logger.Logger.log_train_epoch = spy
    try:
        hp = burgers_hp(tf_epochs=100, nt_epochs=200)
        g, out = run_reference_script("1d-burgers/inf_cont_burgers.py", hp)
    finally:
        ref_logger.Logger.log_train_epoch = orig
    adam = [r for r in rows if not r[2]]
    lb = [r for r in rows if r[2]]
    assert [r[0] for r in adam] == list(range(100)), len(adam)
    np.savez_compressed(os.path.join(HERE, "burgers_default_trace.npz"),
                        adam_losses=np.array([r[1] for r in adam]), lbfgs_iters=np.array([r[0] for r in lb]),
                        lbfgs_losses=np.array([r[1] for r in lb]), final_error=float(g["error"]()),
                        hp=json.dumps(hp))
    print("default trace: %d Adam + %d L-BFGS losses, final error %.6e" % (len(adam), len(lb), float(g["error"]())))


def gen_schrodinger_eval():
    import tensorflow as tf
    hp = {"N_0": 50, "N_b": 50, "N_f": 20000, "layers": [2, 100, 100, 100, 100, 2],
          "tf_epochs": 0, "tf_lr": 0.05, "tf_b1": 0.99, "tf_eps": 1e-1,
          "nt_epochs": 0, "nt_lr": 1.2, "nt_ncorr": 50, "log_frequency": 10}
    for tag, hp_i in (("", hp), ("_small", dict(hp, N_f=1024, layers=[2, 24, 24, 24, 2]))):
        sys.path.insert(0, "1dcomplex-schrodinger")
        g, _ = run_reference_script("1dcomplex-schrodinger/inf_cont_schrodinger.py", hp_i)
        pinn = g["pinn"]
        x0, u0, v0, X0 = g["x0"], g["u0"], g["v0"], g["X0"]
        uv0 = np.concatenate([u0, v0], axis=1)
        w0 = pinn.get_weights().numpy()
        res = {}
        # compat: what the reference script actually does -> fit(x0 [N0,1], ...)
        # intent: X0 = (x0, 0) which prep_data builds and the script never uses
        for mode, Xin in (("compat", x0), ("intent", X0)):
            with contextlib.redirect_stdout(io.StringIO()):
                closure = pinn.get_loss_and_flat_grad(pinn.tensor(Xin), pinn.tensor(uv0))
                loss, grad = closure(tf.convert_to_tensor(w0))
            res["loss_" + mode] = float(loss)
            res["grad_" + mode] = grad.numpy()
        with contextlib.redirect_stdout(io.StringIO()):
            f_u, f_v = pinn.f_model()
        u_pred, v_pred = pinn.predict(g["X_star"])
        # a few Adam steps (compat input, as the script does)
        with contextlib.redirect_stdout(io.StringIO()):
            Xt, ut = pinn.tensor(x0), pinn.tensor(uv0)
            losses = [float(pinn.tf_optimization_step(Xt, ut)) for _ in range(5)]
        np.savez_compressed(
            os.path.join(HERE, "schrodinger_eval%s.npz" % tag), hp=json.dumps(hp_i), w0=w0,
            f_u_first=f_u.numpy()[:64, 0], f_v_first=f_v.numpy()[:64, 0],
            u_pred_stride=u_pred[::517, 0], v_pred_stride=v_pred[::517, 0],
            adam_losses_compat=np.array(losses), w_after_5=pinn.get_weights().numpy(), **res)
        print("schrodinger_eval%s: loss compat=%.17g intent=%.17g" % (
            tag, res["loss_compat"], res["loss_intent"]))


def gen_schrodinger_run():
    """stdout of the unmodified script for 10 Adam epochs (lr .05, b1 .99, eps .1 -- inf_cont_schrodinger.py:23-41):
    the progress lines and the per-evaluation `mse_0 / mse_b / mse_f` print of loss() (:128), plus the final error
    on |h| (:155-158).  The driver call is the reference's own fit(x0 [N0,1], ...) (:164), i.e. the x0-broadcast
    reading."""
    hp = {"N_0": 50, "N_b": 50, "N_f": 20000, "layers": [2, 100, 100, 100, 100, 2],
          "tf_epochs": 10, "tf_lr": 0.05, "tf_b1": 0.99, "tf_eps": 1e-1,
          "nt_epochs": 0, "nt_lr": 1.2, "nt_ncorr": 50, "log_frequency": 1}
    sys.path.insert(0, "1dcomplex-schrodinger")
    g, out = run_reference_script("1dcomplex-schrodinger/inf_cont_schrodinger.py", hp)
    mse = []
    for l in out.splitlines():
        if l.startswith("mse_0"):
            t = l.split()
            mse.append([float(t[1]), float(t[3]), float(t[5])])
    rec = {"hp": hp, "lines": [l for l in out.splitlines() if l.startswith(("tf_epoch", "Training finished"))],
           "mse_0_b_f": mse, "final_error": float(g["error"]()),
           "w_final_sha": sha16(g["pinn"].get_weights().numpy())}
    np.save(os.path.join(HERE, "schrodinger_run_w_final.npy"), g["pinn"].get_weights().numpy().astype(np.float32))
    with open(os.path.join(HERE, "schrodinger_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("schrodinger run: %d mse lines, final error %.6e" % (len(mse), rec["final_error"]))


def gen_full_runs():
    """The reference's DEFAULT schedules end to end for the two non-headline BASELINE configurations:
      schrodinger_default_run.json  configs[3]: inf_cont_schrodinger.py defaults (:23-41: N_f 20000, 4x100, Adam x 200 at
                                    lr .05 / beta_1 .99 / eps .1, no L-BFGS) -- a benign optimiser regime, so the whole
                                    log and the final error on |h| are reproducible quantities
      burgers_ide_cfg3_run.json     configs[2]: ide_cont_burgers.py (whitespace-repaired) with N_u = 10000, 100 Adam +
                                    500 L-BFGS, both models"""
    hp = {"N_0": 50, "N_b": 50, "N_f": 20000, "layers": [2, 100, 100, 100, 100, 2],
          "tf_epochs": 200, "tf_lr": 0.05, "tf_b1": 0.99, "tf_eps": 1e-1,
          "nt_epochs": 0, "nt_lr": 1.2, "nt_ncorr": 50, "log_frequency": 10}
    sys.path.insert(0, "1dcomplex-schrodinger")
    g, out = run_reference_script("1dcomplex-schrodinger/inf_cont_schrodinger.py", hp)
    rec = {"hp": hp, "lines": [l for l in out.splitlines() if l.startswith(("tf_epoch", "Training finished"))],
           "final_error": float(g["error"]())}
    u_pred, v_pred = g["pinn"].predict(g["X_star"])
    rec["h_pred_stride"] = np.sqrt(u_pred ** 2 + v_pred ** 2)[::517, 0].tolist()
    with open(os.path.join(HERE, "schrodinger_default_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("schrodinger default run: final error %.6e" % rec["final_error"], flush=True)
    hp = ide_hp(N_u=10000, tf_epochs=100, nt_epochs=500)
    g, out = _run_repaired_ide(hp)
    keep = ("tf_epoch", "nt_epoch", "Training finished", "l1", "l2")
    rec = {"hp": hp, "lines": [l for l in out.splitlines() if l.startswith(keep)],
           "lambda_1": float(g["lambda_1_pred"]), "lambda_2": float(g["lambda_2_pred"]),
           "lambda_1_noise": float(g["lambda_1_pred_noise"]), "lambda_2_noise": float(g["lambda_2_pred_noise"])}
    with open(os.path.join(HERE, "burgers_ide_cfg3_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("ide cfg3 run: l1 %.6e l2 %.6e | noise l1 %.6e l2 %.6e" % (
        rec["lambda_1"], rec["lambda_2"], rec["lambda_1_noise"], rec["lambda_2_noise"]), flush=True)


def gen_logger_bytes():
    from logger import Logger
    hp = {"log_frequency": 10, "N_f": 3}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        lg = Logger(hp)
        lg.set_error_fn(lambda: 0.0123456)
        lg.start_time = lg.prev_time = 0.0  # deterministic-ish; tests mask the clock fields
        lg.log_train_start(None)
        lg.log_train_opt("Adam")
        lg.log_train_epoch(0, 0.26786867)
        lg.log_train_epoch(7, 0.5)
        lg.log_train_epoch(20, 0.058455, "l1 = 0.5", True)
        lg.log_train_end(300)
    with open(os.path.join(HERE, "logger_bytes.json"), "w") as f:
        json.dump({"hp": hp, "stdout": buf.getvalue()}, f, indent=1)
    print("logger_bytes ok")


def _run_repaired_ide(hp):
    """ide_cont_burgers.py does not parse as shipped (IndentationError line 31); repair_ide_cont.py re-indents it
    (whitespace only, proven by its check()) into a scratch file which is then run unmodified over the shims."""
    import repair_ide_cont
    dst = repair_ide_cont.write("/tmp/_golden_ide/1d-burgers/ide_cont_burgers.py")
    return run_reference_script(dst, hp)


def ide_hp(**kw):
    hp = {"N_u": 2000, "layers": [2, 20, 20, 20, 20, 20, 20, 20, 20, 1], "tf_epochs": 0, "tf_lr": 0.001,
          "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 0, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10}
    hp.update(kw)
    return hp


def gen_burgers_ide_eval():
    """Identification pinned to the reference's own code: BurgersInformedNN.f_model / loss /
    wrap_training_variables / get_weights / set_weights / get_params / fit / predict of
    1d-burgers/ide_cont_burgers.py:47-172 and the prep_data branch burgersutil.py:63-75,99-102, run over the shims
    after the whitespace-only repair.  The model evaluated is the script's second one (its "noise" rerun, :200-205:
    a fresh sample from the continuing numpy stream, weights from the continuing initialiser stream)."""
    import tensorflow as tf
    from custom_lbfgs import lbfgs, Struct
    for tag, N_u in (("", 10000), ("_small", 1500)):
        hp = ide_hp(N_u=N_u)
        g, _ = _run_repaired_ide(hp)
        pinn, X_u, u, X_star = g["pinn"], g["X_u_train"], g["u_train"], g["X_star"]
        w_init = pinn.get_weights().numpy()
        w0 = w_init.copy()
        w0[-2:] = [0.3, -5.0]                  # off the (0, -6) initial values so every lambda term is live
        Xt, ut = pinn.tensor(X_u), pinn.tensor(u)
        closure = pinn.get_loss_and_flat_grad(Xt, ut)
        loss, grad = closure(tf.convert_to_tensor(w0))
        f = pinn.f_model(pinn.X_u).numpy()
        u_pred, f_star = pinn.predict(X_star)
        l1, l2 = pinn.get_params(numpy=True)
        # ---- Adam trajectory (10 steps from w0, lr 1e-3 as the script) ----
        pinn.set_weights(tf.convert_to_tensor(w0))
        losses, snaps = [], {}
        for it in range(10):
            losses.append(float(pinn.tf_optimization_step(Xt, ut)))
            if it + 1 in (1, 10):
                snaps["adam_w_after_%d" % (it + 1)] = pinn.get_weights().numpy()
        # ---- L-BFGS trajectory (25 iterations from w0, reference driver) ----
        pinn.set_weights(tf.convert_to_tensor(w0))
        cfg = Struct()
        cfg.learningRate = hp["nt_lr"]
        cfg.maxIter = 25
        cfg.nCorrection = 6
        cfg.tolFun = 1.0 * np.finfo(float).eps
        logs = []
        x_ret, f_hist, n_eval = lbfgs(closure, pinn.get_weights(), cfg, Struct(), True,
                                      lambda it, fv, is_iter: logs.append((int(it), float(fv))))
        np.savez_compressed(
            os.path.join(HERE, "burgers_ide_eval%s.npz" % tag), hp=json.dumps(hp),
            N_u=N_u, w_init=w_init, w0=w0, loss=float(loss), grad=grad.numpy(), f_first=f[:64, 0], f_sha=sha16(f),
            X_u=X_u, u=u, sha_X_u=sha16(X_u), params=np.array([l1, l2]),
            u_pred_stride=u_pred[::257, 0], f_star_stride=f_star[::257, 0],
            adam_losses=np.array(losses), lbfgs_max_iter=25, lbfgs_n_corr=6,
            lbfgs_log_iters=np.array([l[0] for l in logs]), lbfgs_log_losses=np.array([l[1] for l in logs]),
            lbfgs_f_hist=np.array([float(v) for v in f_hist]), lbfgs_n_eval=int(n_eval),
            lbfgs_x_returned=x_ret.numpy(), lbfgs_w_model=pinn.get_weights().numpy(),
            source="reference ide_cont_burgers.py (whitespace-only repair, tests/golden/repair_ide_cont.py) over the shims",
            **snaps)
        print("burgers_ide_eval%s: loss=%.17g dl1=%.6e dl2=%.6e adam[9]=%.10e lbfgs[-1]=%.10e" % (
            tag, float(loss), grad.numpy()[-2], grad.numpy()[-1], losses[-1], float(f_hist[-1])))


def gen_burgers_ide_run():
    """stdout of the repaired script: 100 Adam + 100 L-BFGS (the default 500 cut to 100 to keep the fixture
    inside the range where two float64 implementations still agree), both models, and the printed lambdas."""
    hp = ide_hp(tf_epochs=100, nt_epochs=100)
    g, out = _run_repaired_ide(hp)
    keep = ("tf_epoch", "nt_epoch", "Training finished", "l1", "l2", "--")
    rec = {"hp": hp, "lines": [l for l in out.splitlines() if l.startswith(keep)],
           "lambda_1": float(g["lambda_1_pred"]), "lambda_2": float(g["lambda_2_pred"]),
           "lambda_1_noise": float(g["lambda_1_pred_noise"]), "lambda_2_noise": float(g["lambda_2_pred_noise"]),
           "u_pred_stride": np.asarray(g["u_pred"])[::257, 0].tolist(),
           "f_pred_stride": np.asarray(g["f_pred"])[::257, 0].tolist()}
    with open(os.path.join(HERE, "burgers_ide_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("ide run: l1 %.6e l2 %.6e | noise l1 %.6e l2 %.6e" % (
        rec["lambda_1"], rec["lambda_2"], rec["lambda_1_noise"], rec["lambda_2_noise"]))


def _install_disc_adapters():
    """Environment adapters that let the two discrete-time scripts run unmodified:
    * burgersutil.py:58,91 read PINNs/Utilities/IRK_weights/Butcher_IRK<q>.txt from a git submodule that is not
      vendored -> np.loadtxt serves the Gauss-Legendre tableau in that file's layout [A | b | c] (the table is
      third-party *data*; oracle.disc.gauss_legendre_butcher restates its published construction);
    * burgersutil.py:89 calls np.asscalar (removed in numpy 1.23);
    * ide_disc_burgers.py:225 calls Logger(frequency=10), a signature utils/logger.py no longer has."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import disc
    if getattr(np.loadtxt, "_disc_adapter", False):
        return
    orig = np.loadtxt

    def loadtxt(path, *a, **k):
        b = os.path.basename(str(path))
        if b.startswith("Butcher_IRK"):
            q = int(b[len("Butcher_IRK"):-4])
            A, bb, c = disc.gauss_legendre_butcher(q)
            v = np.concatenate([A.ravel(), bb, c])
            return v[:, None] if k.get("ndmin") == 2 else v
        return orig(path, *a, **k)
    loadtxt._disc_adapter = True
    np.loadtxt = loadtxt
    np.asscalar = lambda a: a.item()
    import logger as ref_logger
    base = ref_logger.Logger

    class Logger(base):
        def __init__(self, hp=None, frequency=None):
            base.__init__(self, {"log_frequency": frequency} if hp is None else hp)
    ref_logger.Logger = Logger


def gen_burgers_disc():
    import tensorflow as tf
    _install_disc_adapters()
    full = {"N_n": 250, "q": 500, "layers": [1, 50, 50, 50, 501], "tf_epochs": 0, "tf_lr": 0.001,
            "tf_b1": 0.9, "tf_eps": 1e-08, "nt_epochs": 0, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10}
    small = dict(full, N_n=64, q=8, layers=[1, 20, 20, 9])
    for tag, hp in (("", full), ("_small", small)):
        g, _ = run_reference_script("1d-burgers/inf_disc_burgers.py", hp)
        pinn = g["pinn"]
        x_0, u_0, x_1, dt = g["x_0"], g["u_0"], g["x_1"], g["dt"]
        W_irk, x_star, u_star = g["IRK_weights"], g["x_star"], g["u_star"]
        w0 = pinn.get_weights().numpy()
        with contextlib.redirect_stdout(io.StringIO()):
            xt, ut = pinn.tensor(x_0), pinn.tensor(u_0)
            # NeuralNetwork.grad (inf_disc_burgers.py:98-101) is the Adam path and records the loss inside the
            # tape.  The L-BFGS closure (:104-116) computes the loss *outside* its tape: under TensorFlow that
            # yields None gradients (the script cannot reach L-BFGS), under the torch shim a partial gradient.
            # Both are recorded; the engine implements the evident intent = the true gradient for both optimisers.
            loss, grads = pinn.grad(xt, ut)
            grad = tf.concat([tf.reshape(gg, [-1]) for gg in grads], 0)
            closure = pinn.get_loss_and_flat_grad(xt, ut)
            loss_c, grad_c = closure(tf.convert_to_tensor(w0))
            U0 = pinn.U_0_model(xt).numpy()
            pred = np.asarray(pinn.predict(x_star))
            err0 = float(np.linalg.norm(pred - u_star, 2) / np.linalg.norm(u_star, 2))
            losses, snaps = [], {}
            for it in range(10):
                losses.append(float(pinn.tf_optimization_step(xt, ut)))
                if it + 1 in (1, 10):
                    snaps["w_after_%d" % (it + 1)] = pinn.get_weights().numpy()
        np.savez_compressed(
            os.path.join(HERE, "burgers_disc_eval%s.npz" % tag), hp=json.dumps(hp), w0=w0,
            loss=float(loss), grad=grad.numpy(), x_0=x_0, u_0=u_0, x_1=x_1, dt=np.asarray(dt, float),
            irk_dtype=str(W_irk.dtype), irk_shape=np.array(W_irk.shape), irk_sha=sha16(W_irk),
            irk_corner=np.asarray(W_irk[:3, :3], float), U0_first=U0[:8, :8], U0_sha=sha16(U0),
            pred_first=pred[:64], err0=err0, adam_losses=np.array(losses),
            closure_loss=float(loss_c), closure_grad_maxdiff=float(np.abs(grad_c.numpy() - grad.numpy()).max()),
            **snaps)
        print("burgers_disc_eval%s: loss=%.17g |g|1=%.10e err0=%.10e adam[9]=%.10e" % (
            tag, float(loss), float(np.abs(grad.numpy()).sum()), err0, losses[-1]))

    hp_ide = {"N_0": 199, "N_1": 201, "layers": [1, 50, 50, 50, 0], "tf_epochs": 0, "tf_lr": 0.001,
              "tf_b1": 0.9, "tf_eps": None, "nt_epochs": 0, "nt_lr": 0.8, "nt_ncorr": 50}
    for tag, hp in (("", hp_ide), ("_small", dict(hp_ide, N_0=48, N_1=40, layers=[1, 20, 20, 0]))):
        g, _ = run_reference_script("1d-burgers/ide_disc_burgers.py", hp)
        pinn = g["pinn"]                      # the second ("noisy", noise=0.01) model of the script
        x_0, u_0, x_1, u_1 = g["x_0"], g["u_0"], g["x_1"], g["u_1"]
        dt, q, al, be = g["dt"], g["q"], g["IRK_alpha"], g["IRK_beta"]
        w0 = pinn.get_weights().numpy().copy()
        w0[-2:] = [0.7, -5.0]                 # off the (0, -6) initial values so every lambda term is live
        with contextlib.redirect_stdout(io.StringIO()):
            pinn.set_weights(tf.convert_to_tensor(w0))
            T = lambda a: tf.convert_to_tensor(a, dtype=pinn.dtype)
            x0t, u0t, x1t, u1t = T(x_0), T(u_0), T(x_1), T(u_1)
            loss, grads = pinn.grad(x0t, u0t, x1t, u1t)
            flat = np.concatenate([np.asarray(gg.numpy()).ravel() for gg in grads])
            U0p, U1p = pinn.predict(g["x_star"])
            losses = []
            for it in range(10):
                lv, gr = pinn.grad(x0t, u0t, x1t, u1t)
                pinn.tf_optimizer.apply_gradients(zip(gr, pinn.wrap_training_variables()))
                losses.append(float(lv))
            w10 = pinn.get_weights().numpy()
        np.savez_compressed(
            os.path.join(HERE, "burgers_disc_ide_eval%s.npz" % tag), hp=json.dumps(hp), w0=w0,
            loss=float(loss), grad=flat, x_0=x_0, u_0=u_0, x_1=x_1, u_1=u_1, dt=float(dt), q=int(q),
            irk_sha=sha16(np.concatenate([al, be])), U0_first=U0p.numpy()[:8, :8], U1_first=U1p.numpy()[:8, :8],
            adam_losses=np.array(losses), w_after_10=w10, layers=np.array(hp["layers"][:-1] + [int(q)]))
        print("burgers_disc_ide_eval%s: q=%d loss=%.17g dl1=%.6e dl2=%.6e adam[9]=%.10e" % (
            tag, q, float(loss), flat[-2], flat[-1], losses[-1]))


def gen_disc_runs():
    """stdout of the two discrete-time scripts (unmodified, over the shims).  inf_disc: Adam only -- its L-BFGS
    closure has no gradient under TensorFlow (see gen_burgers_disc); ide_disc: Adam + L-BFGS, both models."""
    _install_disc_adapters()
    hp = {"N_n": 250, "q": 500, "layers": [1, 50, 50, 50, 501], "tf_epochs": 200, "tf_lr": 0.001,
          "tf_b1": 0.9, "tf_eps": 1e-08, "nt_epochs": 0, "nt_lr": 0.8, "nt_ncorr": 50, "log_frequency": 10}
    g, out = run_reference_script("1d-burgers/inf_disc_burgers.py", hp)
    keep = ("tf_epoch", "nt_epoch", "Training finished", "l1", "l2", "noisy")
    rec = {"hp": hp, "lines": [l for l in out.splitlines() if l.startswith(keep)],
           "final_error": float(g["error"]())}
    with open(os.path.join(HERE, "burgers_disc_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("disc run: final error %.6e" % rec["final_error"])
    hp = {"N_0": 199, "N_1": 201, "layers": [1, 50, 50, 50, 0], "tf_epochs": 100, "tf_lr": 0.001, "tf_b1": 0.9,
          "tf_eps": None, "nt_epochs": 60, "nt_lr": 0.8, "nt_ncorr": 50}
    g, out = run_reference_script("1d-burgers/ide_disc_burgers.py", hp)
    rec = {"hp": hp, "lines": [l for l in out.splitlines() if l.startswith(keep)],
           "lambda_1": float(g["lambda_1_pred"]), "lambda_2": float(g["lambda_2_pred"]),
           "lambda_1_noisy": float(g["lambda_1_pred_noisy"]), "lambda_2_noisy": float(g["lambda_2_pred_noisy"])}
    with open(os.path.join(HERE, "burgers_disc_ide_run.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("disc ide run: l1 %.6e l2 %.6e" % (rec["lambda_1"], rec["lambda_2"]))


def main():
    os.chdir(REF)
    sys.path.insert(0, SHIMS)
    sys.path.append(HERE)
    sys.path.insert(1, os.path.join(REF, "utils"))
    sys.path.insert(2, os.path.join(REF, "1d-burgers"))
    sys.path.insert(3, os.path.join(REF, "1dcomplex-schrodinger"))
    which = sys.argv[1:] or ["data", "kat", "logger", "burgers", "ide", "schrodinger", "default", "disc"]
    if "data" in which:
        gen_burgers_data()
        gen_schrodinger_data()
    if "kat" in which:
        gen_lbfgs_kat()
    if "logger" in which:
        gen_logger_bytes()
    if "burgers" in which:
        gen_burgers_eval_adam_lbfgs()
    if "burgers_1e6" in which or not sys.argv[1:]:
        gen_burgers_eval_1e6()
    if "ide" in which:
        gen_burgers_ide_eval()
        gen_burgers_ide_run()
    if "schrodinger" in which:
        gen_schrodinger_eval()
    if "schrodinger_run" in which or not sys.argv[1:]:
        gen_schrodinger_run()
    if "default" in which:
        gen_default_run()
    if "default_trace" in which or not sys.argv[1:]:
        gen_default_trace()
    if "full_runs" in which:
        gen_full_runs()
    if "disc" in which:
        gen_burgers_disc()
    if "disc_runs" in which or not sys.argv[1:]:
        gen_disc_runs()


if __name__ == "__main__":
    main()
