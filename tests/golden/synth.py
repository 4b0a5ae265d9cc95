"""Deterministic synthetic weights for the golden fixtures (shared by make_golden.py and the tests).

Weights are NOT the reference initialisers: every tensor is filled from a seeded CPU generator
(torch's CPU Philox/MT stream is reproducible for a given torch build, and the GPU box runs the same
image) with non-trivial values so that biases, LayerNorm/BatchNorm affine terms and running statistics
all matter.  The same state is loaded INTO the reference model by make_golden.py (load_state_dict),
so the fixtures are genuine reference outputs under these weights.
"""
import zlib

import torch


def synth_state(spec, seed=0):
    """spec: dict name -> shape.  Returns dict name -> tensor (float32, or int64 for num_batches_tracked)."""
    out = {}
    for name in spec:
        shape = tuple(spec[name])
        g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(3, dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            out[name] = 0.1 * torch.randn(shape, generator=g)
        elif ("layer_norm" in name or ".bn" in name or "downsample.1" in name) and name.endswith(".weight") and len(shape) == 1:
            out[name] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        elif name == "embedding.0.weight":
            out[name] = 0.5 * torch.randn(shape, generator=g)      # row 0 (<pad>) deliberately non-zero, iq.py:72-73
        elif len(shape) == 4:                                     # conv: He-style so activations stay O(1)
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif len(shape) == 2:
            out[name] = torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5
        else:
            out[name] = torch.randn(shape, generator=g)
    return out
