"""Training log in the reference's stdout format (utils/logger.py:7-60 of PINNs-TF2.0).

Same constructor and methods -- Logger(hp), set_error_fn, log_train_start, log_train_epoch,
log_train_opt, log_train_end -- and byte-compatible progress lines, e.g.

    tf_epoch =     10  elapsed = 00:02 (+00.5)  loss = 3.6782e-01  
    Training finished (epoch 300): duration = 00:19  error = 2.6564e-01  

Only the three TensorFlow banner lines differ: they report the HIP engine and device.
In a data-parallel launch (torchrun, WORLD_SIZE > 1) rank 0 alone prints; the other ranks' Logger is silent
(and never calls the error function).
The loss handed to log_train_epoch is a host float that the engine copies back only at
log points, so logging does not force a device sync per iteration as the reference does.
"""
import json
import os
import time
from datetime import datetime


def _is_root():
    return int(os.environ.get("WORLD_SIZE", "1")) <= 1 or int(os.environ.get("RANK", "0")) == 0


def _engine_banner():
    try:
        import pinn_native
        lib = pinn_native.load()
        n = pinn_native.device_count()
        dev = pinn_native.device_info(0)["name"] if n > 0 else "none"
        return "pinn_hip ABI v%d (HIP/gfx950, no TensorFlow)" % lib.pinn_abi_version(), dev, n > 0
    except Exception as exc:  # banner only; the engine itself fails loudly later
        return "pinn_hip unavailable (%s)" % exc, "none", False


class _Clock(object):
    """Two stopwatches: since construction ("MM:SS") and since the previous logged line ("SS.f")."""

    def __init__(self):
        self.t0 = self.t_last = time.time()

    @staticmethod
    def _fmt(seconds, pattern):
        return datetime.fromtimestamp(seconds).strftime(pattern)

    def total(self):
        return self._fmt(time.time() - self.t0, "%M:%S")

    def lap(self):
        now = time.time()
        text = self._fmt(now - self.t_last, "%S.%f")[:-5]
        self.t_last = now
        return text


_EPOCH_LINE = "{tag} = {epoch:6d}  elapsed = {total} (+{lap})  loss = {loss:.4e}  {custom}"
_END_LINE = "Training finished (epoch {epoch}): duration = {total}  error = {err:.4e}  {custom}"


class Logger(object):
    def __init__(self, hp):
        self.quiet = not _is_root()
        if not self.quiet:
            banner = ["Hyperparameters:", json.dumps(hp, indent=2), ""]
            engine, device, on_gpu = _engine_banner()
            banner += ["Engine: %s" % engine, "Device: %s" % device, "GPU-accerelated: %s" % on_gpu]
            print("\n".join(banner))
        self._clock = _Clock()
        self.frequency = hp["log_frequency"]
        self.error_fn = None
        self.model = None

    # the reference exposes its two time stamps as attributes; keep them readable / settable
    start_time = property(lambda self: self._clock.t0, lambda self, v: setattr(self._clock, "t0", v))
    prev_time = property(lambda self: self._clock.t_last, lambda self, v: setattr(self._clock, "t_last", v))

    def get_epoch_duration(self):
        return self._clock.lap()

    def get_elapsed(self):
        return self._clock.total()

    def set_error_fn(self, error_fn):
        self.error_fn = error_fn

    def get_error_u(self):
        return self.error_fn()

    def log_train_start(self, model, model_description=False):
        self.model = model
        if self.quiet:
            return
        print("\nTraining started\n================")
        if model_description:
            print(model.summary())

    def log_train_opt(self, name):
        if self.quiet:
            return
        print("-- Starting %s optimization --" % name)

    def log_train_epoch(self, epoch, loss, custom="", is_iter=False):
        if self.quiet or epoch % self.frequency:
            return
        print(_EPOCH_LINE.format(tag="nt_epoch" if is_iter else "tf_epoch", epoch=int(epoch),
                                 total=self._clock.total(), lap=self._clock.lap(), loss=float(loss),
                                 custom=custom))

    def log_train_end(self, epoch, custom=""):
        if self.quiet:
            return
        print("==================")
        print(_END_LINE.format(epoch=epoch, total=self._clock.total(), err=float(self.get_error_u()),
                               custom=custom))
