"""Oracle: DDPM/DDIM "simple" UNet + its sigma net (src/unet_simple.py), functional PyTorch-CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class SimpleConfig:
    """The config.model / config.data fields Model.__init__ reads (src/unet_simple.py:193-214)."""
    ch: int
    out_ch: int = 3
    ch_mult: Tuple[int, ...] = (1, 2, 2)
    num_res_blocks: int = 1
    attn_resolutions: Tuple[int, ...] = (16,)
    in_channels: int = 3
    resolution: int = 32
    resamp_with_conv: bool = True
    feat_layer: int = 1
    sigma_block: int = 2


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """get_timestep_embedding (src/unet_simple.py:6-24): [sin || cos], divisor half-1."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -k)
    e = t.float()[:, None] * f[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1, 0, 0))
    return e


def _swish(x: torch.Tensor) -> torch.Tensor:
    """nonlinearity (src/unet_simple.py:27-29): x*sigmoid(x), not F.silu (differs in the last ulp)."""
    return x * torch.sigmoid(x)


def _norm(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Normalize: GroupNorm(32, eps=1e-6) (src/unet_simple.py:32-33)."""
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet_block(sd: SD, p: str, x: torch.Tensor, temb: Optional[torch.Tensor]) -> torch.Tensor:
    """ResnetBlock.forward (src/unet_simple.py:115-134); temb=None -> PureResnetBlock (:461-478)."""
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)))
    if temb is not None:
        h = h + F.linear(_swish(temb), sd[p + ".temb_proj.weight"], sd[p + ".temb_proj.bias"])[:, :, None, None]
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    elif (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x)
    return x + h


def attn_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward (src/unet_simple.py:164-189): single head, logits scaled by c^-1/2 after the bmm."""
    h_ = _norm(sd, p + ".norm", x)
    q, k, v = (_conv(sd, f"{p}.{n}", h_, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h_, padding=0)


def downsample(sd: SD, p: str, x: torch.Tensor, with_conv: bool) -> torch.Tensor:
    """Downsample.forward (src/unet_simple.py:67-74): zero-pad right/bottom, conv s2 p0."""
    if with_conv:
        return _conv(sd, p + ".conv", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)
    return F.avg_pool2d(x, 2, 2)


def unet(sd: SD, cfg: SimpleConfig, x: torch.Tensor, t: torch.Tensor, mode: str = "forward"):
    """Model.forward / encode / forward_and_encode (src/unet_simple.py:302-423)."""
    assert x.shape[2] == x.shape[3] == cfg.resolution
    nres = len(cfg.ch_mult)
    temb = timestep_embedding(t, cfg.ch)
    temb = F.linear(temb, sd["temb.dense.0.weight"], sd["temb.dense.0.bias"])
    temb = F.linear(_swish(temb), sd["temb.dense.1.weight"], sd["temb.dense.1.bias"])

    hs = [_conv(sd, "conv_in", x)]
    res = cfg.resolution
    for lvl in range(nres):
        for blk in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"down.{lvl}.block.{blk}", hs[-1], temb)
            if res in cfg.attn_resolutions:
                h = attn_block(sd, f"down.{lvl}.attn.{blk}", h)
            hs.append(h)
        if lvl != nres - 1:
            hs.append(downsample(sd, f"down.{lvl}.downsample", hs[-1], cfg.resamp_with_conv))
            res //= 2

    h = resnet_block(sd, "mid.block_1", hs[-1], temb)
    if mode == "encode":                                    # :370-376
        if cfg.feat_layer == 0:
            return attn_block(sd, "mid.attn_1", h)
        return resnet_block(sd, "mid.block_2", attn_block(sd, "mid.attn_1", h), temb)
    h = attn_block(sd, "mid.attn_1", h)
    feat = h if cfg.feat_layer == 0 else None
    h = resnet_block(sd, "mid.block_2", h, temb)
    if feat is None:
        feat = h

    for lvl in reversed(range(nres)):
        for blk in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"up.{lvl}.block.{blk}", torch.cat([h, hs.pop()], dim=1), temb)
            if res in cfg.attn_resolutions:
                h = attn_block(sd, f"up.{lvl}.attn.{blk}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            if cfg.resamp_with_conv:
                h = _conv(sd, f"up.{lvl}.upsample.conv", h)
            res *= 2
    out = _conv(sd, "conv_out", _swish(_norm(sd, "norm_out", h)))
    return out if mode == "forward" else (out, feat)


def sigma_net(sd: SD, dim: int, n_blocks: int, feat: torch.Tensor) -> torch.Tensor:
    """SigmaModel.forward (src/unet_simple.py:509-517), down_layer indices as built at :486-499."""
    h = feat
    inp_dim, idx = dim, 0
    for i in range(n_blocks):
        if inp_dim % 2 != 0:
            h = F.pad(h, (0, 1, 0, 1))
            inp_dim += 1
        idx += 1
        h = resnet_block(sd, f"down_layer.{idx}", h, None)
        idx += 1
        if i == 0:
            h = attn_block(sd, f"down_layer.{idx}", h)
            idx += 1
        h = downsample(sd, f"down_layer.{idx}", h, True)
        idx += 1
        inp_dim //= 2
    h = h.flatten(1)
    h = F.linear(h, sd["fc_layer.1.weight"], sd["fc_layer.1.bias"])
    h = F.batch_norm(h, sd["fc_layer.2.running_mean"], sd["fc_layer.2.running_var"], sd["fc_layer.2.weight"],
                     sd["fc_layer.2.bias"], training=False, eps=1e-5)
    h = F.gelu(h)
    return F.linear(h, sd["final_mlp.weight"], sd["final_mlp.bias"])[:, :, None, None]


def sigma_dims(cfg: SimpleConfig):
    """create_simple_sigma_eps_model (src/script_util.py:209-219)."""
    inp_channels = int(cfg.ch * cfg.ch_mult[-1])
    inp_dim = int(cfg.resolution * 0.5 ** (len(cfg.ch_mult) - 1))
    return inp_channels, inp_dim
