"""Deterministic synthetic weights.

The reference ships no checkpoints and its zero-initialised output layers make a
random-init ADM network return exactly 0 (src/nn_util.py:68-74, src/unet_adm.py:210,294,617), so
parity tests, the smoke test and the benchmark all need a construction-order-independent rule
that fills ANY module's ``state_dict`` identically here, in the oracle and in the reference:

    for every key:  g = Generator().manual_seed(crc32(key) ^ seed)
        ndim >= 2 tensors : N(0, gain^2 / fan_in),  fan_in = prod(shape[1:])
        *.running_var     : U(0.5, 1.5)
        *.running_mean    : N(0, 0.1^2)
        norm weights (1-D '...weight' whose sibling bias exists and no >=2-D weight) : 1 + N(0, 0.1^2)
        other 1-D tensors (biases) : N(0, 0.05^2)
        integer tensors (num_batches_tracked) : 0
        0-D / buffers named resample_filter : left untouched (they are architecture constants)
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Mapping, Optional

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)


def fill_state_dict(template: Mapping[str, torch.Tensor], seed: int = 0, gain: float = 1.0,
                    overrides: Optional[Dict[str, float]] = None) -> "OrderedDict[str, torch.Tensor]":
    """Return a new state_dict with the same keys/shapes/dtypes as ``template``, filled by the rule above.

    ``overrides`` maps a key suffix to an extra multiplier (e.g. {'final_mlp.weight': 0.1} keeps the
    NLC residual small, as a trained sigma net's is).
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    keys = set(template.keys())
    for key, ref in template.items():
        shape = tuple(ref.shape)
        g = _gen(key, seed)
        if not ref.dtype.is_floating_point:
            val = torch.zeros(shape, dtype=ref.dtype)
        elif key.endswith("resample_filter") or ref.dim() == 0:
            val = ref.detach().clone()
        elif key.endswith("running_var"):
            val = torch.rand(shape, generator=g) + 0.5
        elif key.endswith("running_mean"):
            val = torch.randn(shape, generator=g) * 0.1
        elif ref.dim() >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            val = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        elif key.endswith("weight"):
            val = 1.0 + 0.1 * torch.randn(shape, generator=g)   # 1-D weight = a normalisation scale
        else:
            val = 0.05 * torch.randn(shape, generator=g)
        if overrides:
            for suf, mul in overrides.items():
                if key.endswith(suf):
                    val = val * mul
        out[key] = val.to(ref.dtype)
    return out


def checksum(sd: Mapping[str, torch.Tensor]):
    """(sum, sum of squares) over every floating tensor, in float64 - recorded in the golden fixtures."""
    s = q = 0.0
    for v in sd.values():
        if v.dtype.is_floating_point:
            d = v.double()
            s += d.sum().item()
            q += (d * d).sum().item()
    return s, q
