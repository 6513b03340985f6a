"""Training log in the reference's stdout format (utils/logger.py:7-60 of PINNs-TF2.0).

Same constructor and methods -- Logger(hp), set_error_fn, log_train_start, log_train_epoch,
log_train_opt, log_train_end -- and byte-compatible progress lines, e.g.

    tf_epoch =     10  elapsed = 00:02 (+00.5)  loss = 3.6782e-01  
    Training finished (epoch 300): duration = 00:19  error = 2.6564e-01  

Only the three TensorFlow banner lines differ: they report the HIP engine and device.
The loss handed to log_train_epoch is a host float that the engine copies back only at
log points, so logging does not force a device sync per iteration as the reference does.
"""
import json
import time
from datetime import datetime


def _engine_banner():
    try:
        import pinn_native
        lib = pinn_native.load()
        n = pinn_native.device_count()
        dev = pinn_native.device_info(0)["name"] if n > 0 else "none"
        return "pinn_hip ABI v%d (HIP/gfx950, no TensorFlow)" % lib.pinn_abi_version(), dev, n > 0
    except Exception as exc:  # banner only; the engine itself fails loudly later
        return "pinn_hip unavailable (%s)" % exc, "none", False


class Logger(object):
    def __init__(self, hp):
        print("Hyperparameters:")
        print(json.dumps(hp, indent=2))
        print()

        engine, device, on_gpu = _engine_banner()
        print("Engine: {}".format(engine))
        print("Device: {}".format(device))
        print("GPU-accerelated: {}".format(on_gpu))

        self.start_time = time.time()
        self.prev_time = self.start_time
        self.frequency = hp["log_frequency"]

    def get_epoch_duration(self):
        now = time.time()
        stamp = datetime.fromtimestamp(now - self.prev_time).strftime("%S.%f")[:-5]
        self.prev_time = now
        return stamp

    def get_elapsed(self):
        return datetime.fromtimestamp(time.time() - self.start_time).strftime("%M:%S")

    def get_error_u(self):
        return self.error_fn()

    def set_error_fn(self, error_fn):
        self.error_fn = error_fn

    def log_train_start(self, model, model_description=False):
        print("\nTraining started")
        print("================")
        self.model = model
        if model_description:
            print(model.summary())

    def log_train_epoch(self, epoch, loss, custom="", is_iter=False):
        if epoch % self.frequency != 0:
            return
        tag = "nt_epoch" if is_iter else "tf_epoch"
        print("%s = %6d  elapsed = %s (+%s)  loss = %.4e  %s" % (
            tag, epoch, self.get_elapsed(), self.get_epoch_duration(), float(loss), custom))

    def log_train_opt(self, name):
        print("-- Starting %s optimization --" % name)

    def log_train_end(self, epoch, custom=""):
        print("==================")
        print("Training finished (epoch %s): duration = %s  error = %.4e  %s" % (
            epoch, self.get_elapsed(), float(self.get_error_u()), custom))
