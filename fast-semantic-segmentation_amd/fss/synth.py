"""Synthetic weights for tests and benchmarks (no checkpoints ship offline).

Values are drawn per variable from a generator seeded by (seed, variable name),
so any subset of variables gets the same values regardless of creation order.
Convolution kernels use He-normal initialisation; batch-norm statistics are
close to identity, with a small gamma on the convolution that closes a residual
branch so activations keep O(1) magnitude through the 16 bottleneck units and
the relu6 layers of the head are neither dead nor saturated.
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode('utf-8'))])


def make_weights(variables, seed=1234, input_std=70.0):
    """variables: {name: shape} (e.g. `Graph.variables`) -> {name: float32 array}.

    `input_std` is the typical magnitude of the preprocessed image (about 70 for
    mean-subtracted 0..255 pixels, about 0.55 for [-1, 1] inputs); kernels that
    read the image (cin == 3) are scaled by its inverse, which is what trained
    batch-norm statistics would do."""
    out = {}
    for name, shape in variables.items():
        rng = _rng(seed, name)
        leaf = name.rsplit('/', 1)[-1]
        closes_residual = ('/conv3/' in name or '/project/' in name)
        if leaf == 'weights':
            kh, kw, cin, _ = shape
            v = rng.normal(0.0, np.sqrt(2.0 / (kh * kw * cin)), shape)
            if cin == 3:
                v = v / input_std
            if name.endswith('Predictions/Conv/weights'):
                # zero-mean over input channels: no class wins on the features' mean alone
                v = v - v.mean(axis=2, keepdims=True)
        elif leaf == 'depthwise_weights':
            kh, kw, _, _ = shape
            v = rng.normal(0.0, np.sqrt(2.0 / (kh * kw)), shape)
        elif leaf == 'gamma':
            v = rng.uniform(0.2, 0.4, shape) if closes_residual else rng.uniform(0.7, 1.1, shape)
        elif leaf == 'beta':
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == 'moving_mean':
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == 'moving_variance':
            v = rng.uniform(0.8, 1.6, shape)
        elif leaf == 'biases':
            v = rng.normal(0.0, 0.1, shape)
        else:
            raise ValueError('no synthetic rule for variable %s' % name)
        out[name] = v.astype(np.float32)
    return out


def make_weights_survey(variables, seed=1234):
    """The plain recipe: every convolution kernel He-normal N(0, 2 / fan_in) with no special
    cases, batch norm gamma ~ U(0.5, 1.5), beta ~ N(0, 0.1), moving_mean ~ N(0, 0.1),
    moving_variance ~ U(0.5, 1.5), biases ~ N(0, 0.1).  Unlike make_weights nothing keeps the
    activations O(1): the residual trunk grows from unit to unit and the relu6 layers of the
    head largely saturate, so label maps have large uniform regions."""
    out = {}
    for name, shape in variables.items():
        rng = _rng(seed, name)
        leaf = name.rsplit('/', 1)[-1]
        if leaf == 'weights':
            kh, kw, cin, _ = shape
            v = rng.normal(0.0, np.sqrt(2.0 / (kh * kw * cin)), shape)
        elif leaf == 'depthwise_weights':
            kh, kw, _, _ = shape
            v = rng.normal(0.0, np.sqrt(2.0 / (kh * kw)), shape)
        elif leaf == 'gamma':
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf in ('beta', 'moving_mean', 'biases'):
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == 'moving_variance':
            v = rng.uniform(0.5, 1.5, shape)
        else:
            raise ValueError('no synthetic rule for variable %s' % name)
        out[name] = v.astype(np.float32)
    return out
