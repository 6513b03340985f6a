"""Oracle (test infrastructure): canonical initial weights.

TensorFlow's seeded glorot_normal stream (utils/neuralnetwork.py:31-37 with
tf.random.set_seed(1234), inf_cont_burgers.py:10) cannot be reproduced without TF, so
parity is defined with a shared initial vector: per Dense layer, in order, from ONE
numpy RandomState(seed): truncnorm(-2,2) * sqrt(2/(fan_in+fan_out))/0.87962566103423978,
zero bias (same recipe as tests/ref_shims/tensorflow.py).
"""
import numpy as np
from scipy.stats import truncnorm


def glorot_flat(layers, seed=1234):
    rs = np.random.RandomState(seed)
    parts = []
    for fi, fo in zip(layers[:-1], layers[1:]):
        std = np.sqrt(2.0 / (fi + fo)) / 0.87962566103423978
        parts.append((truncnorm.rvs(-2, 2, size=(fi, fo), random_state=rs) * std).ravel())
        parts.append(np.zeros(fo))
    return np.concatenate(parts)
