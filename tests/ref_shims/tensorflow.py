"""Test-only stand-in for the `tensorflow` module (torch CPU float64 underneath).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  TensorFlow 2.0.0-rc0 (requirements.txt:5
of the reference) is not installable here, but the reference's own Python
sources only touch a small slice of its API.  Putting this directory first on
PYTHONPATH lets `/root/reference/utils/*.py`, `1d-burgers/inf_cont_burgers.py`
and `1dcomplex-schrodinger/inf_cont_schrodinger.py` run unmodified, which is how
`tests/golden/make_golden.py` produces the golden vectors (SURVEY.md §8c,
Appendix B).  Only used inside the build container; nothing on the GPU box
imports it.

What is restated here (un-pinned upstream behaviour): matmul/tanh/autodiff
(torch f64), the Adam update formula of TF-2.0 `ResourceApplyAdam`, the
glorot_normal initialiser (truncated normal, numpy RandomState stream).
"""
import math
import sys

import numpy as np
import torch
from scipy.stats import truncnorm

torch.set_default_dtype(torch.float64)
__version__ = "2.0.0-rc0 (torch-f64 test shim)"

float64 = "float64"
float32 = "float32"

_TAPES = []          # stack of active tapes
_SEED = [None]


def _raw(x):
    if isinstance(x, T):
        return x.t
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], T):
        return torch.stack([e.t for e in x])
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


class T(object):
    """Plain wrapper (not a Tensor subclass) so tensor-ndarray mixing works."""
    __array_priority__ = 1000

    def __init__(self, t):
        self.t = t

    # arithmetic
    def __add__(self, o): return T(self.t + _raw(o))
    def __radd__(self, o): return T(_raw(o) + self.t)
    def __sub__(self, o): return T(self.t - _raw(o))
    def __rsub__(self, o): return T(_raw(o) - self.t)
    def __mul__(self, o): return T(self.t * _raw(o))
    def __rmul__(self, o): return T(_raw(o) * self.t)
    def __truediv__(self, o): return T(self.t / _raw(o))
    def __rtruediv__(self, o): return T(_raw(o) / self.t)
    def __pow__(self, o): return T(self.t ** o)
    def __neg__(self): return T(-self.t)

    def __iadd__(self, o):
        # TF eager tensors are immutable: `x += y` rebinds
        return T(self.t + _raw(o))

    # comparisons -> python bool (scalars only, as the reference uses them)
    def __gt__(self, o): return bool(self.t > _raw(o))
    def __ge__(self, o): return bool(self.t >= _raw(o))
    def __lt__(self, o): return bool(self.t < _raw(o))
    def __le__(self, o): return bool(self.t <= _raw(o))
    def __bool__(self): return bool(self.t)
    def __float__(self): return float(self.t)
    def __format__(self, spec): return format(float(self.t), spec)
    def __len__(self): return self.t.shape[0]
    def __getitem__(self, k): return T(self.t[k])
    def __iter__(self):
        for i in range(self.t.shape[0]):
            yield T(self.t[i])

    @property
    def shape(self): return tuple(self.t.shape)
    @property
    def dtype(self): return float64

    def numpy(self): return self.t.detach().numpy().copy()
    def __array__(self, dtype=None, copy=None):
        a = self.t.detach().numpy()
        return a.astype(dtype) if dtype is not None else a

    def assign(self, v):
        with torch.no_grad():
            self.t.copy_(_raw(v).reshape(self.t.shape))
        return self

    def __repr__(self): return "T(%r)" % (self.t,)


def convert_to_tensor(x, dtype=None):
    if isinstance(x, T):
        return x
    return T(_raw(x).clone())


def Variable(v, dtype=None):
    t = _raw(v).clone()
    t.requires_grad_(True)
    return T(t)


def reduce_mean(x): return T(torch.mean(_raw(x)))
def reduce_sum(x): return T(torch.sum(_raw(x)))
def square(x): return T(_raw(x) ** 2)
def abs(x): return T(torch.abs(_raw(x)))  # noqa: A001
def exp(x): return T(torch.exp(_raw(x)))
def stack(xs, axis=0): return T(torch.stack([_raw(x) for x in xs], dim=axis))
def concat(xs, axis=0): return T(torch.cat([_raw(x) for x in xs], dim=axis))
def reshape(x, shape): return T(_raw(x).reshape(list(shape)))
def matmul(a, b): return T(_raw(a) @ _raw(b))
def ones(shape, dtype=None): return T(torch.ones(list(shape)))
def executing_eagerly(): return True


def print(*a, **k):  # noqa: A001
    import builtins
    builtins.print(*a, **k)


class _NN(object):
    @staticmethod
    def tanh(x): return T(torch.tanh(_raw(x)))


nn = _NN()


class _Test(object):
    @staticmethod
    def is_gpu_available(): return False


test = _Test()


class _Random(object):
    @staticmethod
    def set_seed(s): _SEED[0] = int(s)


random = _Random()


class GradientTape(object):
    def __init__(self, persistent=False):
        self.persistent = persistent

    def __enter__(self):
        _TAPES.append(self)
        return self

    def __exit__(self, *exc):
        _TAPES.remove(self)
        return False

    def watch(self, x):
        if not x.t.requires_grad:
            x.t.requires_grad_(True)

    def gradient(self, target, sources, output_gradients=None):
        single = not isinstance(sources, (list, tuple))
        srcs = [sources] if single else list(sources)
        tgt = target.t
        go = torch.ones_like(tgt) if output_gradients is None else _raw(output_gradients)
        active = len(_TAPES) > 0
        gs = torch.autograd.grad(
            tgt, [s.t for s in srcs], grad_outputs=go,
            create_graph=active, retain_graph=(self.persistent or active),
            allow_unused=True)
        if not active and not self.persistent:
            # TF eager values carry no graph; drop it so f_hist in lbfgs does
            # not keep every iteration alive.
            target.t = target.t.detach()
        out = [None if g is None else T(g) for g in gs]
        return out[0] if single else out


# ----------------------------------------------------------------------------
# keras
class _InputLayer(object):
    def __init__(self, input_shape=None):
        self.input_shape = input_shape


class _Lambda(object):
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)

    def get_weights(self): return []


_GLOROT_RS = [None]
# sensitivity studies (tests/golden/make_band.py): every initial kernel is multiplied by this factor, e.g.
# 1 + k * 2**-52 = a k-ulp relative perturbation of the initial weights.  1.0 (exact) by default.
_INIT_SCALE = [1.0]


def _glorot_rs():
    if _GLOROT_RS[0] is None:
        seed = _SEED[0] if _SEED[0] is not None else 0
        _GLOROT_RS[0] = np.random.RandomState(seed)
    return _GLOROT_RS[0]


class _Dense(object):
    def __init__(self, units, activation=None, kernel_initializer=None):
        self.units = units
        self.activation = activation
        self.W = None
        self.b = None

    def build(self, fan_in):
        rs = _glorot_rs()
        std = math.sqrt(2.0 / (fan_in + self.units)) / 0.87962566103423978
        w = truncnorm.rvs(-2, 2, size=(fan_in, self.units), random_state=rs) * std
        if _INIT_SCALE[0] != 1.0:
            w = w * _INIT_SCALE[0]
        self.W = T(torch.tensor(w, requires_grad=True))
        self.b = T(torch.zeros(self.units, requires_grad=True))

    def __call__(self, x):
        z = T(_raw(x) @ self.W.t + self.b.t)
        return self.activation(z) if self.activation is not None else z

    def get_weights(self):
        return [self.W.numpy(), self.b.numpy()]

    def set_weights(self, wb):
        with torch.no_grad():
            self.W.t.copy_(_raw(wb[0]).reshape(self.W.t.shape))
            self.b.t.copy_(_raw(wb[1]).reshape(self.b.t.shape))


class _Sequential(object):
    def __init__(self):
        self.layers = []
        self._in = None

    def add(self, layer):
        if isinstance(layer, _InputLayer):
            self._in = layer.input_shape[0]
            return
        if isinstance(layer, _Dense):
            layer.build(self._in)
            self._in = layer.units
        self.layers.append(layer)

    def __call__(self, x):
        h = x if isinstance(x, T) else T(_raw(x))
        for layer in self.layers:
            h = layer(h)
        return h

    @property
    def trainable_variables(self):
        out = []
        for layer in self.layers:
            if isinstance(layer, _Dense):
                out.extend([layer.W, layer.b])
        return out

    def summary(self):
        return "Sequential(%d layers)" % len(self.layers)


class _Adam(object):
    """TF-2.0 ResourceApplyAdam semantics (epsilon outside bias correction)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=None):
        self.lr = learning_rate
        self.b1 = beta_1
        self.b2 = beta_2
        self.eps = 1e-7 if epsilon is None else epsilon
        self.it = 0
        self.state = {}

    def apply_gradients(self, grads_and_vars):
        self.it += 1
        t = self.it
        alpha = self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is None:
                    continue
                key = id(v.t)
                if key not in self.state:
                    self.state[key] = (torch.zeros_like(v.t), torch.zeros_like(v.t))
                m, s = self.state[key]
                gt = g.t.detach()
                m += (1.0 - self.b1) * (gt - m)
                s += (1.0 - self.b2) * (gt * gt - s)
                v.t -= alpha * m / (torch.sqrt(s) + self.eps)


class _Layers(object):
    InputLayer = _InputLayer
    Lambda = _Lambda
    Dense = _Dense


class _Optimizers(object):
    Adam = _Adam


class _Backend(object):
    @staticmethod
    def set_floatx(x): pass


class _Keras(object):
    Sequential = _Sequential
    layers = _Layers
    optimizers = _Optimizers
    backend = _Backend


keras = _Keras()


def _reset_for_new_process_like_state():
    """Called by golden generators between independent reference runs."""
    _GLOROT_RS[0] = None
    _SEED[0] = None
    del _TAPES[:]
