"""Drop-in for the sampling-path helpers of the reference's src/utils.py (vector_norm, normalize) on GPU tensors."""
import math

import torch

from diffusion_nlc_amd import ops


def vector_norm(x, keepdim=True):
    """src/utils.py:7-9 via nlc_row_sumsq."""
    n = ops.row_sumsq(x.contiguous().float()).sqrt()
    return n.view(-1, *([1] * (x.dim() - 1))) if keepdim else n


def normalize(x, inp_dim, eps=1e-12, norm_detach=False):
    """src/utils.py:11-16 via nlc_row_sumsq + nlc_scale_rows."""
    denom = torch.clamp(ops.row_sumsq(x.contiguous().float()).sqrt(), min=eps)
    return ops.scale_rows(x.contiguous().float(), math.sqrt(inp_dim) / denom)


def get_model_size(model):
    """Parameter + buffer size in MiB (what the entry points print)."""
    size = sum(p.nelement() * p.element_size() for p in model.parameters())
    size += sum(b.nelement() * b.element_size() for b in model.buffers())
    return size / 1024 ** 2
