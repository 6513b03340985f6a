"""NeuralNetwork base class with the reference's public surface
(utils/neuralnetwork.py:7-159 of PINNs-TF2.0) on top of the MI355X HIP engine.

What is kept: constructor `NeuralNetwork(hp, logger, ub, lb)`, the attributes
`nt_config, tf_epochs, tf_optimizer, dtype, model, sizes_w, sizes_b, logger`, and the methods
`loss, grad, wrap_training_variables, get_params, get_weights, set_weights,
get_loss_and_flat_grad, tf_optimization, tf_optimization_step, nt_optimization,
nt_optimization_steps, fit, predict, summary, tensor` with the same argument meaning and
return shapes (numpy arrays where the reference returns eager tensors).

What changes: there is no TensorFlow, so a subclass cannot spell its PDE with GradientTapes.
It names one of the engine's residual kinds instead (`pde="burgers" | "burgers_ide" |
"schrodinger" | "burgers_disc" | "burgers_disc_ide"`) and the engine evaluates forward, u_t/u_x/u_xx, residual, loss and the flat
gradient on the GPU (csrc/).  Extra, optional hp keys: "dtype" ("f64" default = the reference's
arithmetic, neuralnetwork.py:24-26 | "f32" = the throughput mode north_star sanctions) for the
kernel arithmetic, "device" (HIP ordinal), "nt_guard" (see nt_optimization).  Host interchange stays float64.

Data parallel (north_star; the reference has no distributed code): launched as
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 1d-burgers/inf_cont_burgers.py [hp.json]
every process is one rank of ONE model: device = LOCAL_RANK, the collocation / data / boundary sets handed to the
class are split in contiguous blocks over the ranks (mean() denominators stay global), the engines exchange the
[P+4] float64 gradient vector once per evaluation (RCCL all-reduce over xGMI), every rank applies the same optimiser
update to its replica, and rank 0 alone prints, plots and saves.  hp["data_parallel"] = false keeps N independent
full-batch replicas.  Everything a script calls keeps its single-process meaning: predict / f_model / error_l2 return
full arrays on every rank.

There is no CPU path: constructing a NeuralNetwork without the HIP library or a GPU raises.
"""
import hashlib

import numpy as np

from custom_lbfgs import lbfgs, Struct

import os
import sys
sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pinn_native import Engine, parallel  # noqa: E402


_INIT_STREAM = {"seed": 1234, "rs": None}


def set_seed(seed):
    """Counterpart of the scripts' `tf.random.set_seed(1234)`: (re)starts the stream the glorot initialisers of
    all models built afterwards draw from.  As with TensorFlow's global seed, a second model built in the same
    process continues the stream instead of repeating the first model's weights (the identification scripts
    build two)."""
    _INIT_STREAM["seed"] = int(seed)
    _INIT_STREAM["rs"] = np.random.RandomState(int(seed))


def _init_stream():
    if _INIT_STREAM["rs"] is None:
        set_seed(_INIT_STREAM["seed"])
    return _INIT_STREAM["rs"]


class _AdamConfig(object):
    """Holds what tf.keras.optimizers.Adam held (neuralnetwork.py:19-22)."""

    def __init__(self, learning_rate, beta_1, epsilon, beta_2=0.999):
        self.learning_rate = learning_rate
        self.beta_1 = beta_1
        self.beta_2 = beta_2
        self.epsilon = 1e-7 if epsilon is None else epsilon     # Keras default


class _ModelView(object):
    """Stand-in for the Keras Sequential the reference exposes as `self.model`: callable for
    a plain forward pass, `.summary()`, `.layers` describing the Dense stack."""

    def __init__(self, owner, layers):
        self._owner = owner
        self.layers = [("lambda", "2(X-lb)/(ub-lb)-1")] + [
            ("dense", fi, fo, "tanh" if i < len(layers) - 2 else "linear")
            for i, (fi, fo) in enumerate(zip(layers[:-1], layers[1:]))]

    def __call__(self, X):
        return self._owner._engine.predict(_as_points(X, self._owner))

    def summary(self):
        rows = ["%-10s %-14s %8s" % ("layer", "shape", "params")]
        total = 0
        for l in self.layers[1:]:
            n = l[1] * l[2] + l[2]
            total += n
            rows.append("%-10s %-14s %8d" % ("dense/" + l[3], "[%d,%d]" % (l[1], l[2]), n))
        rows.append("total trainable scalars: %d" % total)
        return "\n".join(rows)


def _as_points(X, owner):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    if owner.layers[0] == 1:            # discrete-time models: the network input is x alone
        return X.reshape(-1, 1)
    if X.shape[1] == 1:
        # Schrodinger driver quirk (inf_cont_schrodinger.py:164): x0 of shape [N,1] is handed
        # to a 2-input network.  Default = what the reference computes: the Lambda layer broadcasts
        # [N,1] against lb/ub [2] (neuralnetwork.py:29-30), i.e. the network sees (x0, x0).
        # hp["compat_x0_broadcast"] = false switches to the evident intent (x0, t=0) = prep_data's X0.
        second = X if owner._compat_x0 else np.zeros_like(X)
        X = np.concatenate([X, second], axis=1)
    return X


class NeuralNetwork(object):
    pde = "burgers"
    MAX_RESTARTS = 5            # nt_guard: discards per nt_optimization call

    def __init__(self, hp, logger, ub, lb, pde=None):
        layers = hp["layers"]
        if pde is not None:
            self.pde = pde

        # L-BFGS configuration, same fields as the reference (neuralnetwork.py:13-17)
        self.nt_config = Struct()
        self.nt_config.learningRate = hp["nt_lr"]
        self.nt_config.maxIter = hp["nt_epochs"]
        self.nt_config.nCorrection = hp["nt_ncorr"]
        self.nt_config.tolFun = 1.0 * np.finfo(float).eps
        self.tf_epochs = hp["tf_epochs"]
        self.tf_optimizer = _AdamConfig(hp["tf_lr"], hp["tf_b1"], hp["tf_eps"])

        self.dtype = "float64"                       # host interchange dtype
        self.compute_dtype = hp.get("dtype", "f64")  # kernel arithmetic; the reference is float64 (:24-26)
        self._compat_x0 = bool(hp.get("compat_x0_broadcast", True))
        self.layers = [int(v) for v in layers]
        self.ub = np.asarray(ub, dtype=np.float64)
        self.lb = np.asarray(lb, dtype=np.float64)

        # one rank of a torchrun launch = one shard of the point sets on GPU LOCAL_RANK (module docstring); the
        # discrete-time models hold <= 256 points and stay replicated
        self._dp = None if self.pde.startswith("burgers_disc") else parallel.from_env(bool(hp.get("data_parallel", True)))
        self.is_root = parallel.is_root()
        world, _, local_rank = parallel.env_world()
        device = hp.get("device", os.environ.get("PINN_DEVICE", local_rank if world > 1 else 0))
        self._engine = Engine(self.layers, self.lb, self.ub, pde=self.pde,
                              dtype=self.compute_dtype, device=int(device))
        self.comm_mode = "none"
        if self._dp:
            self.comm_mode = parallel.init_engine_comm(self._engine, self._dp.dist, self._dp.world, self._dp.rank)
        self._X_f = None
        self._n_f_total = 0
        self.model = _ModelView(self, self.layers)

        # flat-layout bookkeeping, same rule as the reference (all hidden widths = layers[1])
        self.sizes_w = []
        self.sizes_b = []
        for i, width in enumerate(self.layers):
            if i != 1:
                self.sizes_w.append(int(width * self.layers[1]))
                self.sizes_b.append(int(width if i != 0 else self.layers[1]))

        self._engine.set_weights(self._initial_weights(hp))
        if hp.get("init_weights"):                   # resume from a checkpoint written by save_weights
            self.load_weights(hp["init_weights"])
        self._engine.adam_init(self.tf_optimizer.learning_rate, self.tf_optimizer.beta_1,
                               self.tf_optimizer.beta_2, self.tf_optimizer.epsilon)
        self._bound = None
        self.logger = logger
        # optional: re-draw the collocation set on the device every k Adam epochs (not in the reference, which
        # samples once on the host).  Seeds are resample_seed + epoch.
        self._resample_every = int(hp.get("resample_every", 0))
        self._resample_seed = int(hp.get("resample_seed", 1234))
        # L-BFGS restart guard (nt_optimization): off in the reference's arithmetic, on in float32
        guard = hp.get("nt_guard", 1e3 if self.compute_dtype in ("f32", "float32") else 0.0)
        self._nt_guard = float(guard or 0.0)
        self._async_log = bool(hp.get("async_log", True))        # log lines one chunk behind the GPU (see _pipelined)
        self.nt_restarts = []

    # ---- point sets (sharded over the ranks of a data-parallel launch) ---------------------------
    def _set_collocation(self, X_f):
        """collocation points of the residual term (1d-burgers/inf_cont_burgers.py:55-56): this rank's block goes to
        the GPU, the mean keeps its global denominator"""
        X_f = np.ascontiguousarray(np.asarray(X_f, dtype=np.float64).reshape(-1, 2))
        self._X_f, self._n_f_total = X_f, X_f.shape[0]
        if self._dp:
            lo, hi = self._dp.shard(X_f.shape[0])
            self._engine.set_collocation(X_f[lo:hi], n_total=X_f.shape[0])
        else:
            self._engine.set_collocation(X_f)

    def _set_boundary(self, X_lb, X_ub):
        """periodic-boundary pairs (1dcomplex-schrodinger/inf_cont_schrodinger.py:50-53); a pair stays on one rank"""
        if self._dp:
            lo, hi = self._dp.shard(len(X_lb))
            self._engine.set_boundary(X_lb[lo:hi], X_ub[lo:hi], n_total=len(X_lb))
        else:
            self._engine.set_boundary(X_lb, X_ub)

    def _residual_collocation(self):
        """f at ALL collocation points [N_f, n_out], on every rank (a shard holds only its block: the replicated
        weights are evaluated at the full set instead)"""
        if self._dp:
            if self._X_f is None:
                # after a device-side redraw (hp["resample_every"]) every rank holds only its block of the new design:
                # the blocks are gathered once (rank order = design order) and kept until the next redraw.  Collective.
                self._X_f = self._dp.gather_rows(self._engine.get_collocation())
            return self._engine.residual_at(self._X_f)
        return self._engine.residual()

    # ---- initialisation ------------------------------------------------------------------------
    def _n_net(self):
        return sum(fi * fo + fo for fi, fo in zip(self.layers[:-1], self.layers[1:]))

    def _extra_params(self):
        """Trainable scalars appended after the network weights (identification: lambdas)."""
        return np.zeros(0)

    def _initial_weights(self, hp):
        """glorot_normal kernels + zero biases (neuralnetwork.py:31-37): truncated normal within
        two sigma, sigma = sqrt(2/(fan_in+fan_out))/0.87962566103423978, drawn per Dense layer
        from the process-wide stream started by set_seed (default 1234; the first model of a process gets the
        engine's canonical initial vector), or from a private RandomState(hp["seed"]) when that key is given."""
        from scipy.stats import truncnorm
        rs = np.random.RandomState(int(hp["seed"])) if "seed" in hp else _init_stream()
        # hp["init_scale"] (diagnostic): every initial kernel multiplied by it -- how the ulp-perturbation ensembles of
        # tests/golden/make_band.py are reproduced on the GPU (tests/test_gpu_end_to_end.py)
        scale = float(hp.get("init_scale", 1.0))
        chunks = []
        for fi, fo in zip(self.layers[:-1], self.layers[1:]):
            sigma = np.sqrt(2.0 / (fi + fo)) / 0.87962566103423978
            chunks.append((truncnorm.rvs(-2, 2, size=(fi, fo), random_state=rs) * sigma).ravel() * scale
                          if scale != 1.0 else (truncnorm.rvs(-2, 2, size=(fi, fo), random_state=rs) * sigma).ravel())
            chunks.append(np.zeros(fo))
        chunks.append(self._extra_params())
        return np.concatenate(chunks)

    # ---- loss / gradient -----------------------------------------------------------------------
    def loss(self, u, u_pred):
        """Plain data misfit, as the base class of the reference (neuralnetwork.py:51-52)."""
        return float(np.mean(np.square(np.asarray(u) - np.asarray(u_pred))))

    @staticmethod
    def _digest(X, u):
        h = hashlib.blake2b(digest_size=16)          # full-buffer digest: a few ms at 1e6 points
        h.update(np.ascontiguousarray(X).tobytes())
        h.update(np.ascontiguousarray(u).tobytes())
        return (X.shape, u.shape, h.digest())

    def _bind(self, X, u, key=None):
        """make (X, u) the engine's data set unless it already is; `key` = a digest computed earlier for exactly
        these (private, unchanged) arrays, so that a closure does not re-hash its data on every evaluation"""
        X = _as_points(X, self)
        u = np.asarray(u, dtype=np.float64).reshape(X.shape[0], -1)
        if key is None:
            key = self._digest(X, u)
        if key != self._bound:
            if self._dp:                     # this rank's block of the data set; mean() over the global count
                lo, hi = self._dp.shard(X.shape[0])
                self._engine.set_data(X[lo:hi], u[lo:hi], n_total=X.shape[0])
            else:
                self._engine.set_data(X, u)
            self._bound = key
            self._X_bound = X
        return key

    def _split(self, flat):
        out, off = [], 0
        for fi, fo in zip(self.layers[:-1], self.layers[1:]):
            out.append(flat[off:off + fi * fo].reshape(fi, fo).copy())
            off += fi * fo
            out.append(flat[off:off + fo].copy())
            off += fo
        for v in flat[off:]:
            out.append(np.array([v]))
        return out

    def grad(self, X, u):
        self._bind(X, u)
        loss_value, flat, _ = self._engine.loss_grad()
        return loss_value, self._split(flat)

    def wrap_training_variables(self):
        return self._split(self._engine.get_weights())

    def get_params(self, numpy=False):
        return []

    def _log_custom(self):
        """Text appended to a logged progress line (the identification scripts print their lambdas)."""
        return ""

    def get_weights(self, convert_to_tensor=True):
        w = self._engine.get_weights()
        return w if convert_to_tensor else list(w)

    def set_weights(self, w):
        self._engine.set_weights(np.asarray(w, dtype=np.float64).ravel())

    # ---- checkpointing (the reference has none; the flat vector IS the natural format) --
    def save_weights(self, path):
        """np.save of the float64 flat weight vector (reference layout, get_weights())."""
        np.save(path, self.get_weights())
        return path

    def load_weights(self, path):
        w = np.load(path)
        if w.shape != (self._engine.n_params,):
            raise ValueError("checkpoint holds %s values, the model has %d" % (w.shape, self._engine.n_params))
        self.set_weights(w)

    def get_loss_and_flat_grad(self, X, u):
        # the closure owns private copies of its data (like the reference's immutable tensors), hashed once here
        Xc = np.array(_as_points(X, self), dtype=np.float64)
        uc = np.array(u, dtype=np.float64).reshape(Xc.shape[0], -1)
        key = self._bind(Xc, uc)

        def loss_and_flat_grad(w):
            self._bind(Xc, uc, key)      # the closure evaluates ITS data, whatever was bound in between
            self.set_weights(w)
            loss_value, flat, _ = self._engine.loss_grad()
            return loss_value, flat

        return loss_and_flat_grad

    # ---- Adam ------------------------------------------------------------------------------------
    def tf_optimization(self, X_u, u):
        self.logger.log_train_opt("Adam")
        self._bind(X_u, u)
        freq = max(int(self.logger.frequency), 1)
        epoch = 0
        n_design = self._n_f_total or self._engine.n_f
        every = self._resample_every if n_design > 0 else 0
        if self._pipelined() and not every:
            return self._tf_optimization_pipelined(freq)
        while epoch < self.tf_epochs:
            if every and epoch > 0 and epoch % every == 0:
                # one design for the whole job; a rank draws its own block of it (counter-based: no communication)
                lo, hi = self._dp.shard(n_design) if self._dp else (0, n_design)
                self._engine.lhs_collocation(n_design, self._resample_seed + epoch, first=lo, count=hi - lo)
                self._X_f = None
            # run up to and including the next epoch that is logged, then sync once
            stop = min(self.tf_epochs, (epoch + freq - 1) // freq * freq + 1)
            if every:
                stop = min(stop, (epoch // every + 1) * every)
            losses = self._adam_chunk(stop - epoch)
            for k, loss_value in enumerate(losses):
                last = epoch + k == stop - 1     # the weights on the device are those after this epoch
                self.logger.log_train_epoch(epoch + k, loss_value, self._log_custom() if last else "")
            epoch = stop

    def _pipelined(self):
        """log lines one chunk behind the GPU (Engine.adam_enqueue / lbfgs_enqueue): the kernels of the next chunk are in
        the stream while this process formats and prints the previous chunk's lines, so a log line costs the device
        nothing.  Same chunks, same kernels, same numbers as the synchronous loops.  Off (hp["async_log"] = false, or
        automatically) when a line needs more than the chunk's losses: subclasses that append text read from the device
        (_log_custom: the discrete-time identification script's lambdas) or print per evaluation without
        the enqueue / collect pair (_adam_chunk alone), periodic resampling.  The restart guard goes along: its way back is a device-side snapshot behind every chunk."""
        cls = type(self)
        chunk_ok = cls._adam_chunk is NeuralNetwork._adam_chunk or cls._adam_collect is not NeuralNetwork._adam_collect
        return (self._async_log and cls._log_custom is NeuralNetwork._log_custom and chunk_ok
                and hasattr(self._engine, "adam_enqueue"))

    def _log_boundaries(self, total, freq):
        """chunk ends: every chunk runs up to and including the next epoch that is logged (0, freq, 2 freq, ...)"""
        stops, epoch = [], 0
        while epoch < total:
            epoch = min(total, (epoch + freq - 1) // freq * freq + 1)
            stops.append(epoch)
        return stops

    def _tf_optimization_pipelined(self, freq):
        eng, start, queue = self._engine, 0, []
        for stop in self._log_boundaries(self.tf_epochs, freq) + [None]:
            if stop is not None:
                queue.append((start, self._adam_enqueue(stop - start)))
                start = stop
            while queue and (stop is None or len(queue) > 1):       # keep ONE chunk ahead of the one being logged
                first, ticket = queue.pop(0)
                for k, loss_value in enumerate(self._adam_collect(ticket)):
                    self.logger.log_train_epoch(first + k, loss_value, "")

    def _adam_chunk(self, n):
        """n device-resident Adam steps; the loss before each update.  Subclasses that print per-evaluation
        diagnostics (the Schrodinger loss parts) override this -- and _adam_enqueue / _adam_collect beside it if they want
        their lines one chunk behind the GPU as well."""
        return self._engine.adam_run(n)

    def _adam_enqueue(self, n):
        return self._engine.adam_enqueue(n)

    def _adam_collect(self, ticket):
        return self._engine.adam_collect(ticket)

    def tf_optimization_step(self, X_u, u):
        self._bind(X_u, u)
        return float(self._engine.adam_run(1)[0])

    # ---- L-BFGS ----------------------------------------------------------------------------------
    def nt_optimization(self, X_u, u):
        """Device-resident L-BFGS with the semantics of custom_lbfgs.lbfgs as the reference drives
        it (neuralnetwork.py:118-136): same config Struct, same log callbacks, and the model ends
        at the last evaluated iterate."""
        self.logger.log_train_opt("LBFGS")
        self._bind(X_u, u)
        cfg = self.nt_config
        if cfg.maxIter == 0:
            return

        def begin(iters_left, lr_scale=1.0):
            self._engine.lbfgs_begin(iters_left, (cfg.learningRate or 1) * lr_scale, cfg.nCorrection or 100,
                                     cfg.tolFun or 1e-5, cfg.tolX or 1e-19, cfg.maxEval or 0.0)

        begin(cfg.maxIter)
        freq = max(int(self.logger.frequency), 1)
        # Restart guard, hp["nt_guard"] = G (default 1e3 in float32, 0 = off in float64 = the reference's behaviour).
        # The reference's L-BFGS has no line search (utils/custom_lbfgs.py:159-163: t = learningRate): a curvature
        # pair with y.s barely above its 1e-10 test makes the next step arbitrarily long, and the run is lost -- in
        # ANY arithmetic (profiles/r04_diag_f32_k-10_shadow.txt: the float64 update from float64 pairs takes the
        # same step).  With G > 0 a chunk of iterations whose loss becomes non-finite or exceeds G x the lowest loss
        # accepted so far is discarded: the weights go back to the last accepted chunk boundary and L-BFGS starts
        # again there with an empty history for the iterations that are left.  A run that never explodes is
        # untouched (same kernels, same iterates); at most MAX_RESTARTS discards per call.
        # A restart from the SAME boundary as the one before it would be a bit-for-bit replay (same weights, empty
        # history, reproducible kernels -> the same explosion): each repeat at one boundary halves the step length
        # (learningRate x 0.5^repeats) from there on -- for ALL remaining iterations of the call, not only for the chunk
        # that failed (the factor is printed with the restart and recorded in nt_restarts).  A chunk flagged bad is never
        # made the restart point, also once the restarts are spent.
        # Chunks of log_frequency iterations.  Pipelined (see _pipelined): the host stays ONE chunk behind the GPU -- chunk
        # k + 1 is in the stream while chunk k is judged and logged; `done` is seen one chunk late, and a chunk enqueued after
        # the run has ended changes nothing on the device.  With the guard on, the point to go back to is a device-side
        # snapshot taken in stream order behind every chunk (Engine.weights_snapshot: no host round trip), and a chunk that is
        # judged bad takes the chunk that ran ahead of it down with it.  Synchronous otherwise (the scripted engines of the
        # CPU tests, subclasses whose lines read device state): same decisions, the restart point a host copy.
        eng = self._engine
        pipe = self._pipelined() and (self._nt_guard <= 0 or hasattr(eng, "weights_snapshot"))
        guard, base, restarts, done = self._nt_guard, 0, 0, 0
        last_restart_at, repeats = None, 0
        best = np.inf                                       # lowest loss of an accepted chunk
        keep, keep_it = None, 0                             # restart point: a snapshot slot (pipelined) or a host copy
        if guard > 0:
            if pipe:
                eng.weights_snapshot(0)
                keep = 0
            else:
                keep = eng.get_weights()
        queue = []                                          # pipelined: (ticket, snapshot slot behind that chunk) in flight

        def free_slot():
            used = {keep} | {s for _, s in queue}
            return next(s for s in range(eng.N_SNAPSHOTS) if s not in used)

        while not done:
            if pipe:
                while len(queue) < 2:
                    slot = free_slot() if guard > 0 else -1
                    ticket = eng.lbfgs_enqueue(freq)
                    if guard > 0:
                        eng.weights_snapshot(slot)
                    queue.append((ticket, slot))
                ticket, after = queue.pop(0)
                iters, losses, done = eng.lbfgs_collect(ticket)
            else:
                iters, losses, done = eng.lbfgs_run(freq)
            if guard > 0 and len(losses):
                if not np.isfinite(best):
                    if not np.isfinite(losses[0]):          # already lost before L-BFGS started: nothing to go back to
                        guard = 0.0
                    best = float(losses[0])
                bad = (~np.isfinite(losses)) | (losses > guard * best) if guard > 0 else np.zeros(len(losses), bool)
                if bad.any() and restarts < self.MAX_RESTARTS and keep_it < cfg.maxIter:
                    k = int(np.argmax(bad))
                    restarts += 1
                    repeats = repeats + 1 if last_restart_at == keep_it else 0
                    last_restart_at = keep_it
                    # (iteration whose loss was refused, iteration restarted from, learningRate factor in force from there
                    #  to the end of this call -- the reduced step of a repeated restart PERSISTS: going back to the full
                    #  step would need another lbfgs_begin, i.e. another loss of the curvature history)
                    self.nt_restarts.append((base + int(iters[k]), keep_it, 0.5 ** repeats))
                    if self.is_root:
                        print("nt_guard: loss %.3e at L-BFGS iteration %d (lowest accepted %.3e): discarded, restarting "
                              "from iteration %d%s" % (float(losses[k]), base + int(iters[k]), best, keep_it,
                                                       " with learningRate x %g for the rest of the run" % 0.5 ** repeats
                                                       if repeats else ""),
                              file=sys.stderr)
                    if pipe:
                        eng.weights_restore(keep)
                        queue = []                          # (lbfgs_begin drops the chunk that ran ahead)
                    else:
                        eng.set_weights(keep)
                    base, done = keep_it, 0
                    begin(cfg.maxIter - base, 0.5 ** repeats)
                    continue
            for k, (it, loss_value) in enumerate(zip(iters, losses)):
                custom = self._log_custom() if k == len(iters) - 1 else ""
                self.logger.log_train_epoch(base + int(it), loss_value, custom, True)
            if guard > 0 and len(iters) and not done and not bad.any():
                best = min(best, float(np.min(losses)))
                keep, keep_it = (after if pipe else eng.get_weights()), base + int(iters[-1])
        for ticket, _ in queue:                             # the chunk(s) that ran ahead of the end: their entries, if any
            iters, losses, _ = eng.lbfgs_collect(ticket)
            for it, loss_value in zip(iters, losses):
                self.logger.log_train_epoch(base + int(it), loss_value, "", True)

    def nt_optimization_steps(self, loss_and_flat_grad):
        """Host-driven variant with the reference's signature: any closure w -> (loss, grad)."""
        return lbfgs(loss_and_flat_grad, self.get_weights(), self.nt_config, Struct(), True,
                     lambda epoch, loss, is_iter:
                     self.logger.log_train_epoch(epoch, loss, "", is_iter))

    # ---- driver ----------------------------------------------------------------------------------
    def fit(self, X_u, u):
        if self._nt_guard > 0 and self.is_root and self.nt_config.maxIter:
            # one line, on stderr (stdout stays byte-compatible with the reference's log): this run is allowed to depart
            # from custom_lbfgs.py:159-163 (no line search, a lost run stays lost) at a restart
            print("note: L-BFGS restart guard on (nt_guard = %g, %s): a chunk whose loss explodes is discarded and L-BFGS "
                  "restarts from the last accepted iterate -- the reference has no such guard; hp[\"nt_guard\"] = 0 switches "
                  "it off" % (self._nt_guard, "the float32 default" if self.compute_dtype in ("f32", "float32")
                             else "set by hp"), file=sys.stderr)
        self.logger.log_train_start(self)
        X_u = self.tensor(X_u)
        u = self.tensor(u)
        self.tf_optimization(X_u, u)
        self.nt_optimization(X_u, u)
        self.logger.log_train_end(self.tf_epochs + self.nt_config.maxIter)
        if self._dp and not self._dp.replicas_identical(self._engine.get_weights()):
            raise RuntimeError("data-parallel replicas hold different weights after training (rank %d of %d)"
                               % (self._dp.rank, self._dp.world))
        bad = self._engine.status()[1]
        if bad:
            print("warning: the loss became non-finite at evaluation %d" % bad, file=sys.stderr)

    def predict(self, X_star):
        return self._engine.predict(_as_points(X_star, self))

    def error_l2(self, X_star, reference, modulus=False):
        """The scripts' error metric ||reference - model(X_star)||_2 / ||reference||_2
        (1d-burgers/inf_cont_burgers.py:114-116 through utils/logger.py:56-60), reduced on the device: the grid and
        the reference field are uploaded once, 24 bytes come back.  modulus=True compares |h| of a two-output
        model with reference [n] (1dcomplex-schrodinger/inf_cont_schrodinger.py:155-158)."""
        return self._engine.error_l2(_as_points(X_star, self), reference, modulus=modulus)

    def status(self):
        """(evaluations so far, number of the first evaluation whose loss was NaN/Inf or 0): the reference has no
        such guard, a NaN just propagates (utils/custom_lbfgs.py:154)"""
        return self._engine.status()

    def summary(self):
        return self.model.summary()

    def tensor(self, X):
        return np.asarray(X, dtype=np.float64)
