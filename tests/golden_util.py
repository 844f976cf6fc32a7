"""Shared by oracle/gen_golden*.py (build container) and the parity tests (anywhere).

`fill_params` overwrites every entry of a module's state_dict with values drawn from a
generator seeded by the crc32 of the entry's NAME, so the reference module (in the build
container) and the bridgeqa_amd module (in the tests) hold identical weights iff their
state-dict key sets and shapes are identical -- no weight blobs need to be committed.
"""
import math
import zlib

import numpy as np
import torch

MAX_ELEMS = 1 << 16


def fill_params(module, prefix=""):
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            g = torch.Generator().manual_seed(zlib.crc32((prefix + name).encode()) & 0x7FFFFFFF)
            if not t.is_floating_point():
                if name.endswith("position_ids"):
                    continue
                t.zero_()
                continue
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                v = torch.rand(t.shape, generator=g) + 0.5
            elif leaf == "running_mean":
                v = torch.randn(t.shape, generator=g) * 0.1
            elif t.dim() <= 1 and leaf == "weight":  # norm scales
                v = torch.rand(t.shape, generator=g) + 0.5
            elif t.dim() <= 1:  # biases
                v = torch.randn(t.shape, generator=g) * 0.1
            else:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (1.0 / math.sqrt(max(fan_in, 1)))
            t.copy_(v.to(t.dtype))
    return sorted((prefix + k, tuple(v.shape)) for k, v in sd.items())


def subsample(a):
    """Deterministic thinning of big arrays so fixtures stay small."""
    a = np.asarray(a)
    if a.size <= MAX_ELEMS:
        return a
    stride = -(-a.size // MAX_ELEMS)
    return a.reshape(-1)[::stride]
