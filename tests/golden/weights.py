"""Deterministic weights for the model-level golden fixtures.

The fixtures store inputs' seeds and expected OUTPUTS; the (large) weights are a
pure function of (state_dict key order, shapes, seed) through NumPy's frozen
legacy `RandomState`, so both the generator (tests/golden/make_golden.py, run
against the imported reference modules) and the tests (run against this repo's
modules / the oracle) rebuild the identical tensors.
"""
import collections

import numpy as np


def fill_state_dict(shapes, seed):
    """shapes: ordered {key: shape} as produced by `module.state_dict()`.
    Returns {key: np.ndarray} (float32, or int64 for num_batches_tracked)."""
    rs = np.random.RandomState(seed)
    out = collections.OrderedDict()
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if key.endswith('num_batches_tracked'):
            out[key] = np.zeros(shape, np.int64)
        elif key.endswith('running_var'):
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif key.endswith('running_mean'):
            out[key] = (0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif len(shape) >= 2:  # conv / linear weight: xavier-uniform bound
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            bound = np.sqrt(6.0 / (fan_in + fan_out))
            out[key] = rs.uniform(-bound, bound, shape).astype(np.float32)
        elif key.endswith('weight'):  # BN gamma
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        else:  # biases (BN beta, conv bias)
            out[key] = (0.1 * rs.standard_normal(shape)).astype(np.float32)
    return out


def load_into(module, seed):
    """Fill `module` (any nn.Module) in place; returns the ordered shape dict."""
    import torch
    sd = module.state_dict()
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in sd.items())
    new = fill_state_dict(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in new.items()})
    return shapes
