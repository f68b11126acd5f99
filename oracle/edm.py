"""Oracle: EDM DDPM++ SongUNet + its sigma net (src/edm_networks.py), functional PyTorch-CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Only the configuration
create_edm_sigma_eps_model builds is restated (src/script_util.py:222-270): positional
embedding, 'standard' encoder/decoder, resample_filter [1,1], adaptive_scale False,
num_heads 1, skip_scale sqrt(1/2), GroupNorm eps 1e-6.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.functional import silu

SD = Dict[str, torch.Tensor]
SKIP_SCALE = float(np.sqrt(0.5))     # block_kwargs skip_scale (src/edm_networks.py:766)
GN_EPS = 1e-6


@dataclass
class EdmConfig:
    img_resolution: int
    in_channels: int = 3
    out_channels: int = 3
    augment_dim: int = 0
    model_channels: int = 128
    channel_mult: Tuple[int, ...] = (1, 2, 2, 2)
    channel_mult_emb: int = 4
    num_blocks: int = 4
    attn_resolutions: Tuple[int, ...] = (16,)
    sigma_block: int = 2


def positional_embedding(x: torch.Tensor, num_channels: int, max_positions: int = 10000) -> torch.Tensor:
    """PositionalEmbedding(endpoint=True).forward (src/edm_networks.py:219-225)."""
    freqs = torch.arange(0, num_channels // 2, dtype=torch.float32)
    freqs = freqs / (num_channels // 2 - 1)
    freqs = (1 / max_positions) ** freqs
    x = x.ger(freqs.to(x.dtype))
    return torch.cat([x.cos(), x.sin()], dim=1)


def _gn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """GroupNorm: groups = min(32, C//4) (src/edm_networks.py:105-116)."""
    w = sd[p + ".weight"]
    groups = min(32, w.shape[0] // 4)
    return F.group_norm(x, groups, w, sd[p + ".bias"], GN_EPS)


def _conv2d(sd: SD, p: str, x: torch.Tensor, up: bool = False, down: bool = False) -> torch.Tensor:
    """Conv2d.forward, non-fused path with resample_filter [1,1] (src/edm_networks.py:73-98)."""
    w = sd.get(p + ".weight")
    b = sd.get(p + ".bias")
    c = x.shape[1]
    if up or down:
        f = torch.ones(1, 1, 2, 2) / 4.0          # f.ger(f) / f.sum()^2  with f = [1,1]
        if up:
            x = F.conv_transpose2d(x, f.mul(4).tile([c, 1, 1, 1]), groups=c, stride=2, padding=0)
        else:
            x = F.conv2d(x, f.tile([c, 1, 1, 1]), groups=c, stride=2, padding=0)
    if w is not None:
        x = F.conv2d(x, w, padding=w.shape[-1] // 2)
    if b is not None:
        x = x + b.reshape(1, -1, 1, 1)
    return x


def unet_block(sd: SD, p: str, x: torch.Tensor, emb: Optional[torch.Tensor], up: bool = False, down: bool = False,
               attention: bool = False) -> torch.Tensor:
    """UNetBlock.forward with adaptive_scale=False (src/edm_networks.py:183-205); emb=None -> PureUNetBlock (:942-955)."""
    orig = x
    x = _conv2d(sd, p + ".conv0", silu(_gn(sd, p + ".norm0", x)), up=up, down=down)
    if emb is not None:
        params = F.linear(emb, sd[p + ".affine.weight"], sd[p + ".affine.bias"])[:, :, None, None]
        x = silu(_gn(sd, p + ".norm1", x + params))
    # NB PureUNetBlock applies conv1 directly to conv0's output: there is no norm1/silu in its forward (:945)
    x = _conv2d(sd, p + ".conv1", x)
    if (p + ".skip.weight") in sd or up or down:
        x = x + _conv2d(sd, p + ".skip", orig, up=up, down=down)
    else:
        x = x + orig
    x = x * SKIP_SCALE
    if attention:
        n, c = x.shape[0], x.shape[1]
        qkv = _conv2d(sd, p + ".qkv", _gn(sd, p + ".norm2", x))
        q, k, v = qkv.reshape(n * 1, c // 1, 3, -1).unbind(2)
        w = torch.einsum("ncq,nck->nqk", q.to(torch.float32), (k / np.sqrt(k.shape[1])).to(torch.float32)).softmax(dim=2)
        a = torch.einsum("nqk,nck->ncq", w, v)
        x = _conv2d(sd, p + ".proj", a.reshape(*x.shape)) + x
        x = x * SKIP_SCALE
    return x


def _encoder_layout(cfg: EdmConfig):
    """Names/flags of self.enc in construction order (src/edm_networks.py:782-809)."""
    out = []
    for level, mult in enumerate(cfg.channel_mult):
        res = cfg.img_resolution >> level
        if level == 0:
            out.append((f"{res}x{res}_conv", "conv", False))
        else:
            out.append((f"{res}x{res}_down", "down", False))
        for idx in range(cfg.num_blocks):
            out.append((f"{res}x{res}_block{idx}", "block", res in cfg.attn_resolutions))
    return out


def _decoder_layout(cfg: EdmConfig):
    """Names/flags of self.dec in construction order (src/edm_networks.py:812-833)."""
    out = []
    nlev = len(cfg.channel_mult)
    for level, mult in reversed(list(enumerate(cfg.channel_mult))):
        res = cfg.img_resolution >> level
        if level == nlev - 1:
            out.append((f"{res}x{res}_in0", "block_noskip", True))
            out.append((f"{res}x{res}_in1", "block_noskip", False))
        else:
            out.append((f"{res}x{res}_up", "up", False))
        for idx in range(cfg.num_blocks + 1):
            out.append((f"{res}x{res}_block{idx}", "block", idx == cfg.num_blocks and res in cfg.attn_resolutions))
        if level == 0:
            out.append((f"{res}x{res}_aux_norm", "aux_norm", False))
            out.append((f"{res}x{res}_aux_conv", "aux_conv", False))
    return out


def unet(sd: SD, cfg: EdmConfig, x: torch.Tensor, noise_labels: torch.Tensor, mode: str = "forward"):
    """SongUNet.forward / encode (second definition, src/edm_networks.py:835-909); augment_labels=None."""
    emb = positional_embedding(noise_labels, cfg.model_channels)
    emb = emb.reshape(emb.shape[0], 2, -1).flip(1).reshape(*emb.shape)      # swap sin/cos (:838)
    emb = silu(F.linear(emb, sd["map_layer0.weight"], sd["map_layer0.bias"]))
    emb = silu(F.linear(emb, sd["map_layer1.weight"], sd["map_layer1.bias"]))

    skips: List[torch.Tensor] = []
    for name, kind, attn in _encoder_layout(cfg):
        p = "enc." + name
        if kind == "conv":
            x = _conv2d(sd, p, x)
        else:
            x = unet_block(sd, p, x, emb, down=(kind == "down"), attention=attn)
        skips.append(x)
    if mode == "encode":
        return x

    out = None
    tmp = None
    for name, kind, attn in _decoder_layout(cfg):
        p = "dec." + name
        if kind == "aux_norm":
            tmp = _gn(sd, p, x)
        elif kind == "aux_conv":
            out = _conv2d(sd, p, silu(tmp))
        else:
            in_ch = sd[p + ".norm0.weight"].shape[0]
            if x.shape[1] != in_ch:
                x = torch.cat([x, skips.pop()], dim=1)
            x = unet_block(sd, p, x, emb, up=(kind == "up"), attention=attn)
    return out


def sigma_net(sd: SD, dim: int, n_blocks: int, feat: torch.Tensor) -> torch.Tensor:
    """SigmaModel.forward (src/edm_networks.py:1014-1022); down_layer as built at :994-1004."""
    h = feat
    inp_dim, idx = dim, 0
    for i in range(n_blocks):
        if inp_dim % 2 != 0:
            h = F.pad(h, (0, 1, 0, 1))
            inp_dim += 1
        idx += 1
        h = unet_block(sd, f"down_layer.{idx}", h, None, attention=(i % 2 == 0))
        idx += 1
        # Downsample(channels, True): pad (0,1,0,1) + conv s2 p0 (:970-974)
        h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"down_layer.{idx}.conv.weight"], sd[f"down_layer.{idx}.conv.bias"], stride=2)
        idx += 1
        inp_dim //= 2
    h = h.flatten(1)
    h = F.linear(h, sd["fc_layer.1.weight"], sd["fc_layer.1.bias"])
    h = F.batch_norm(h, sd["fc_layer.2.running_mean"], sd["fc_layer.2.running_var"], sd["fc_layer.2.weight"],
                     sd["fc_layer.2.bias"], training=False, eps=1e-5)
    h = silu(h)
    return F.linear(h, sd["final_mlp.weight"], sd["final_mlp.bias"])[:, :, None, None]


def sigma_dims(cfg: EdmConfig):
    """create_edm_sigma_eps_model (src/script_util.py:264-268)."""
    inp_channels = int(cfg.model_channels * cfg.channel_mult[-1])
    inp_dim = int(cfg.img_resolution * 0.5 ** (len(cfg.channel_mult) - 1))
    return inp_channels, inp_dim
