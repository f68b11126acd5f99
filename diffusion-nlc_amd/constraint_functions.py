"""Inpainting constraint for restoration sampling (BASELINE config 4).

Mirrors the parts of the reference that the inpainting path touches:
``functions/svd_operators.py:324-359`` (``Inpainting.A`` / ``A_pinv``), ``src/constraint_functions.py:216-241``
(the ``svd_constraint`` inpainting branch) and ``image_sample.py:282-343,371-383`` (``Constraint_Function`` with the
``affine_svd`` projection  x0 <- x0 - A^+(A x0 - y)).  For inpainting that projection is "copy the known
pixels": the per-step work is fused into ``nlc_sched_step`` (mask / known), A and A^+ themselves are the two
gather kernels of ``csrc/constraint.hip``.  The other degradations of the reference (SR, deblurring,
colorisation, compressed sensing, DDRM) are out of scope (not in BASELINE's configs).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import _ext, ops
from ._ext import NlcError, check


def _s():
    return torch.cuda.current_stream().cuda_stream


class Inpainting:
    """functions/svd_operators.py:324-359.  ``missing_indices`` index the pixel-interleaved (H*W, C) flattening."""

    def __init__(self, channels, img_dim, missing_indices, device):
        self.channels, self.img_dim = channels, img_dim
        self.device = torch.device(device)
        n = channels * img_dim ** 2
        missing = torch.as_tensor(missing_indices).long().cpu()
        keep = torch.ones(n, dtype=torch.bool)
        keep[missing] = False
        self.missing_indices = missing.to(self.device)
        kept = torch.nonzero(keep).reshape(-1)
        inv = torch.full((n,), -1, dtype=torch.int32)
        inv[kept] = torch.arange(kept.numel(), dtype=torch.int32)
        self.kept_indices = kept.to(self.device).contiguous()
        self._inv = inv.to(self.device).contiguous()
        self._singulars = torch.ones(kept.numel(), device=self.device)
        # [C][HW] f32 mask of KNOWN entries, the layout nlc_sched_step wants
        self.mask_chw = keep.view(img_dim * img_dim, channels).t().contiguous().float().to(self.device)

    def singulars(self):
        return self._singulars

    @ops.on_device
    def A(self, vec):
        x = vec.reshape(vec.shape[0], self.channels, -1).to(self.device, torch.float32).contiguous()
        B, C, HW = x.shape
        nk = self.kept_indices.numel()
        out = torch.empty(B, nk, device=self.device, dtype=torch.float32)
        check(_ext.load().nlc_inpaint_A(x.data_ptr(), self.kept_indices.data_ptr(), out.data_ptr(), B, C, HW, nk, _s()),
              "nlc_inpaint_A")
        return out

    @ops.on_device
    def A_pinv(self, vec):
        y = vec.reshape(vec.shape[0], -1).to(self.device, torch.float32).contiguous()
        B, nk = y.shape
        if nk != self.kept_indices.numel():
            raise ValueError("A_pinv: measurement length mismatch")
        C, HW = self.channels, self.img_dim ** 2
        out = torch.empty(B, C * HW, device=self.device, dtype=torch.float32)
        check(_ext.load().nlc_inpaint_Apinv(y.data_ptr(), self._inv.data_ptr(), out.data_ptr(), B, C, HW, nk, _s()),
              "nlc_inpaint_Apinv")
        return out

    At = A_pinv          # singular values are all 1: A^T = A^+


def svd_constraint(fn, fn_scale=4, device="cuda:0", base_mask_dir="store/inp_masks", image_size=256, channels=3):
    """src/constraint_functions.py:206-241, inpainting branch only."""
    if "inpainting" not in fn:
        raise NotImplementedError(f"constraint '{fn}': only the inpainting operators are on the HIP path (SURVEY.md §8 f-1)")
    if fn == "inpainting_random":
        missing_r = torch.randperm(image_size ** 2)[: image_size ** 2 // 2].long() * 3
    elif fn in ("inpainting_ddnm", "inpainting_half"):
        name = "mask.npy" if fn == "inpainting_ddnm" else "mask_half.npy"
        mask = torch.from_numpy(np.load(os.path.join(base_mask_dir, name))).reshape(-1)
        missing_r = torch.nonzero(mask == 0).long().reshape(-1) * 3
    else:
        missing_r = torch.load(os.path.join(base_mask_dir, "mask_random.pt")).long().cpu()
    missing = torch.cat([missing_r, missing_r + 1, missing_r + 2], dim=0)
    return Inpainting(channels, image_size, missing, device)


class AffineInpaint:
    """``partial(affine_svd, A=A, Ap=Ap, y=y)`` (image_sample.py:376-381) as an object the sampling loop can fuse:
    x0 - A^+(A x0 - y) == where(known, A^+ y, x0)."""

    def __init__(self, op: Inpainting, y: torch.Tensor, shape):
        self.op = op
        self.y = y
        self.mask_chw = op.mask_chw
        self.known = op.A_pinv(y).view(*shape).contiguous()

    def __call__(self, x0_t):
        A, Ap = self.op.A, self.op.A_pinv
        flat = x0_t.reshape(x0_t.size(0), -1)
        return x0_t - Ap(A(flat) - self.y.reshape(self.y.size(0), -1)).reshape(*x0_t.size())


class Constraint_Function:
    """image_sample.py:282-343 for deg='inpainting*' / proj='svd'."""

    def __init__(self, deg, A_funcs: Inpainting, channels=3, image_size=256, lr=1.0):
        if "inp" not in deg:
            raise NotImplementedError(deg)
        self.deg, self.op = deg, A_funcs
        self.A, self.Ap = A_funcs.A, A_funcs.A_pinv
        self.proj, self.channels, self.image_size, self.lr = "svd", channels, image_size, lr

    def transform(self, x):
        return self.A(x)

    def inv_transform(self, y):
        Apy = self.Ap(y).view(y.shape[0], self.channels, self.image_size, self.image_size)
        if self.deg == "inpainting":                 # image_sample.py:321-322
            Apy = Apy + self.Ap(self.A(torch.ones_like(Apy))).reshape(*Apy.shape) - 1
        return Apy

    def constraint_fn(self, x0_t, y, lambda_t=None):
        return AffineInpaint(self.op, y, x0_t.shape)(x0_t)

    def bind(self, y, shape) -> AffineInpaint:
        """The per-batch projection (what evaluate_constraint builds with functools.partial, image_sample.py:648)."""
        return AffineInpaint(self.op, y, shape)

    def loss(self, x, y):
        y_hat = self.transform(x)
        x_hat = self.inv_transform(y)
        f = torch.linalg.vector_norm(y_hat - y, ord=1, dim=tuple(range(1, y.dim()))).cpu()
        b = torch.linalg.vector_norm(x_hat - x, ord=1, dim=tuple(range(1, x.dim()))).cpu()
        return f, b
