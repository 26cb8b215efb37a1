"""Parameter initialisers for the yolo3_darknet53 parameter table (host side, numpy).

``uniform``   what ``net.initialize()`` does in the reference with gluon's default initializer
              (train_yolov3.py:428) [UPSTREAM-RECALLED]: conv weights U(-0.07, 0.07), biases 0,
              gamma 1, beta 0, running_mean 0, running_var 1.
``synthetic`` a variance-preserving random network for benchmarks and parity tests (there are no
              pretrained weights offline): He-style conv weights for LeakyReLU(0.1), the 3x3 conv
              closing every residual block scaled by 0.25 and the conv after each stage scaled back by
              (1 + 0.45*0.25)^n_blocks, so activations stay O(1) through all 75 layers and the heads
              produce a realistic spread of scores; BN statistics near identity.
"""
import zlib

import numpy as np

RES_GAIN = 0.25
# first conv after a run of n residual blocks: name prefix -> n
_AFTER_BLOCKS = {"stages.0.3": 1, "stages.0.6": 2, "stages.1.0": 8, "stages.2.0": 8,
                 "yolo_blocks.0.body.0": 4}


def _rng(seed, name):
    """One generator per (seed, tensor name): a tensor's values do not depend on the order of the table
    (the library's parameter table and the oracle's list the heads in different orders)."""
    if seed is None:
        return np.random.default_rng()
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def uniform_params(table, seed=None, scale=0.07):
    out = {}
    for name, shape in table:
        rng = _rng(seed, name)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "weight":
            out[name] = rng.uniform(-scale, scale, shape).astype(np.float32)
        elif leaf in ("gamma", "running_var"):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def synthetic_params(table, seed=233, obj_bias=0.0):
    """table: [(structural name, reference shape)].  obj_bias is added to the objectness rows of the
    three prediction biases (obj_bias = -5 gives a 'trained-like' sparse candidate set)."""
    out = {}
    for name, shape in table:
        rng = _rng(seed, name)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "weight":
            fan = shape[1] * shape[2] * shape[3]
            g = np.sqrt(2.0 / (1.01 * fan))
            if name.startswith("stages") and ".body.1.0." in name:
                g *= RES_GAIN
            for pre, n in _AFTER_BLOCKS.items():
                if name.startswith(pre + ".0."):
                    g /= (1.0 + 0.45 * RES_GAIN) ** n
            if "prediction" in name:
                g = np.sqrt(1.0 / fan)
            out[name] = (rng.standard_normal(shape) * g).astype(np.float32)
        elif leaf == "gamma":
            out[name] = rng.uniform(0.9, 1.1, shape).astype(np.float32)
        elif leaf in ("beta", "running_mean"):
            out[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        elif leaf == "running_var":
            out[name] = rng.uniform(0.9, 1.1, shape).astype(np.float32)
        elif leaf == "bias":
            b = (rng.standard_normal(shape) * 0.1).astype(np.float32)
            b.reshape(3, -1)[:, 4] += np.float32(obj_bias)
            out[name] = b
        else:
            raise ValueError("unknown parameter leaf in %r" % name)
    return out
