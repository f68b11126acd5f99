"""Oracle: ADM (guided-diffusion) UNet + its sigma net, functional PyTorch-CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works directly on a reference-format
``state_dict`` (same key names as src/unet_adm.py modules produce), NCHW f32 tensors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class AdmConfig:
    """Constructor arguments of UNetModel (src/unet_adm.py:427-449) that shape the graph."""
    image_size: int
    in_channels: int = 3
    model_channels: int = 128
    out_channels: int = 3
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = ()     # downsample rates, as UNetModel receives them
    channel_mult: Tuple[float, ...] = (1, 2, 4, 8)
    conv_resample: bool = True
    num_heads: int = 1
    num_head_channels: int = -1
    num_heads_upsample: int = -1
    use_scale_shift_norm: bool = False
    resblock_updown: bool = False
    use_new_attention_order: bool = False
    feat_layer: int = 1


def _hi(x: torch.Tensor) -> torch.Tensor:
    """The reference's ``.float()`` up-casts (GroupNorm32, softmax, timestep embedding) - except that a float64 tensor stays
    float64: with a float64 state_dict and input the same graph evaluates in double precision end to end (the "f64 leg" of
    tools/parity_trace.py: the function the f32 arithmetic approximates, same parameters)."""
    return x if x.dtype == torch.float64 else x.float()


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0, dtype=torch.float32) -> torch.Tensor:
    """src/nn_util.py:103-121: [cos(t f) || sin(t f)], f_i = exp(-ln(max_period) i / half).  The frequency table is computed in
    f32 as upstream (it is a parameter of the function); ``dtype`` = float64 evaluates t f, cos and sin in double."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(dtype)
    args = t[:, None].to(dtype) * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn32(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """GroupNorm32: 32 groups, eps 1e-5, computed in f32 (src/nn_util.py:17-19,93-100)."""
    return F.group_norm(_hi(x), 32, sd[p + ".weight"], sd[p + ".bias"], 1e-5).type(x.dtype)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _heads(cfg_heads: int, head_channels: int, channels: int) -> int:
    """AttentionBlock.__init__ (src/unet_adm.py:277-283)."""
    return cfg_heads if head_channels == -1 else channels // head_channels


def res_block(sd: SD, p: str, x: torch.Tensor, emb: Optional[torch.Tensor], *, scale_shift: bool,
              up: bool = False, down: bool = False) -> torch.Tensor:
    """ResBlock._forward (src/unet_adm.py:236-256); with emb=None it is PureResNetBlock (:793-796)."""
    h = F.silu(_gn32(sd, p + ".in_layers.0", x))
    if up:                                  # Upsample(use_conv=False): nearest x2 on both branches (:190-192)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:                              # Downsample(use_conv=False): AvgPool2d(2) (:193-195,136)
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = _conv(sd, p + ".in_layers.2", h)
    if emb is not None:
        emb_out = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
        if scale_shift:
            scale, shift = torch.chunk(emb_out, 2, dim=1)
            h = _gn32(sd, p + ".out_layers.0", h) * (1 + scale) + shift
            h = F.silu(h)
        else:
            h = F.silu(_gn32(sd, p + ".out_layers.0", h + emb_out))
    else:
        h = F.silu(_gn32(sd, p + ".out_layers.0", h))
    h = _conv(sd, p + ".out_layers.3", h)   # Dropout is the identity in eval mode
    if (p + ".skip_connection.weight") in sd:
        w = sd[p + ".skip_connection.weight"]
        x = F.conv2d(x, w, sd[p + ".skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def attention_block(sd: SD, p: str, x: torch.Tensor, heads: int, new_order: bool) -> torch.Tensor:
    """AttentionBlock._forward + QKVAttentionLegacy / QKVAttention (src/unet_adm.py:299-305,337-354,370-389)."""
    b, c, *spatial = x.shape
    x = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn32(sd, p + ".norm", x), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    bs, width, length = qkv.shape
    ch = width // (3 * heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = (q * scale).reshape(bs * heads, ch, length)
        k = (k * scale).reshape(bs * heads, ch, length)
        v = v.reshape(bs * heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * heads, ch * 3, length).split(ch, dim=1)
        q, k = q * scale, k * scale
    weight = torch.einsum("bct,bcs->bts", q, k)
    weight = torch.softmax(_hi(weight), dim=-1).type(weight.dtype)
    a = torch.einsum("bts,bcs->bct", weight, v).reshape(bs, -1, length)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (x + h).reshape(b, c, *spatial)


def _walk_input(cfg: AdmConfig):
    """Yield (block index, [(kind, sub index, channels)...]) as UNetModel.__init__ lays them out (:482-539)."""
    ch = int(cfg.channel_mult[0] * cfg.model_channels)
    yield 0, [("conv_in", 0, ch)]
    n, ds = 1, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            ch = int(mult * cfg.model_channels)
            layers = [("res", 0, ch)]
            if ds in cfg.attention_resolutions:
                layers.append(("attn", 1, ch))
            yield n, layers
            n += 1
        if level != len(cfg.channel_mult) - 1:
            yield n, [("res_down" if cfg.resblock_updown else "downsample", 0, ch)]
            n += 1
            ds *= 2


def unet(sd: SD, cfg: AdmConfig, x: torch.Tensor, timesteps: torch.Tensor, mode: str = "forward", y=None):
    """UNetModel.forward / encode / forward_and_encode (src/unet_adm.py:636-731).

    mode: 'forward' -> out ; 'encode' -> feat ; 'both' -> (out, feat)
    """
    ss = cfg.use_scale_shift_norm
    heads_up = cfg.num_heads if cfg.num_heads_upsample == -1 else cfg.num_heads_upsample
    temb = timestep_embedding(timesteps, cfg.model_channels, dtype=sd["time_embed.0.weight"].dtype)
    emb = F.linear(temb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if y is not None:                                   # class-conditional: + label_emb(y) (src/unet_adm.py:652-654)
        emb = emb + F.embedding(y, sd["label_emb.weight"])

    hs = []
    h = x
    ds = 1
    input_chans = []
    for n, layers in _walk_input(cfg):
        for kind, j, ch in layers:
            p = f"input_blocks.{n}.{j}"
            if kind == "conv_in":
                h = _conv(sd, p, h)
            elif kind == "res":
                h = res_block(sd, p, h, emb, scale_shift=ss)
            elif kind == "attn":
                h = attention_block(sd, p, h, _heads(cfg.num_heads, cfg.num_head_channels, ch), cfg.use_new_attention_order)
            elif kind == "res_down":
                h = res_block(sd, p, h, emb, scale_shift=ss, down=True)
            elif kind == "downsample":      # Downsample (src/unet_adm.py:113-140)
                h = _conv(sd, p + ".op", h, stride=2) if cfg.conv_resample else F.avg_pool2d(h, 2)
        hs.append(h)
        input_chans.append(h.shape[1])

    def middle(hh):
        hh = res_block(sd, "middle_block.0", hh, emb, scale_shift=ss)
        hh = attention_block(sd, "middle_block.1", hh, _heads(cfg.num_heads, cfg.num_head_channels, hh.shape[1]),
                             cfg.use_new_attention_order)
        return res_block(sd, "middle_block.2", hh, emb, scale_shift=ss)

    if mode == "encode":                    # :685-693
        return h if cfg.feat_layer == 0 else middle(h)
    feat = h if cfg.feat_layer == 0 else None
    h = middle(h)
    if feat is None:
        feat = h

    ds = 2 ** (len(cfg.channel_mult) - 1)
    n = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            p = f"output_blocks.{n}"
            h = res_block(sd, p + ".0", h, emb, scale_shift=ss)
            j = 1
            ch = int(cfg.model_channels * mult)
            if ds in cfg.attention_resolutions:
                h = attention_block(sd, f"{p}.{j}", h, _heads(heads_up, cfg.num_head_channels, ch), cfg.use_new_attention_order)
                j += 1
            if level and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    h = res_block(sd, f"{p}.{j}", h, emb, scale_shift=ss, up=True)
                else:                        # Upsample (:81-110)
                    h = F.interpolate(h, scale_factor=2, mode="nearest")
                    if cfg.conv_resample:
                        h = _conv(sd, f"{p}.{j}.conv", h)
                ds //= 2
            n += 1
    out = _conv(sd, "out.2", F.silu(_gn32(sd, "out.0", h)))
    return out if mode == "forward" else (out, feat)


# --------------------------------------------------------------------------------------
@dataclass
class AdmSigmaConfig:
    """SigmaModel.__init__ arguments (src/unet_adm.py:1030-1032)."""
    dim: int
    channels: int
    n_blocks: int = 2
    num_heads: int = 1
    num_head_channels: int = -1
    use_new_attention_order: bool = False


def sigma_net(sd: SD, cfg: AdmSigmaConfig, feat: torch.Tensor) -> torch.Tensor:
    """SigmaModel.forward (src/unet_adm.py:1074-1083); down_layer indexing as built at :1037-1050."""
    h = feat
    inp_dim = cfg.dim
    idx = 0
    for i in range(cfg.n_blocks):
        if inp_dim % 2 != 0:
            h = F.pad(h, (0, 1, 0, 1))      # ConstantPad2d((0,1,0,1), 0)
            inp_dim += 1
        idx += 1                             # pad / Identity slot
        h = res_block(sd, f"down_layer.{idx}", h, None, scale_shift=False)
        idx += 1
        if i == 0:
            h = attention_block(sd, f"down_layer.{idx}", h, _heads(cfg.num_heads, cfg.num_head_channels, cfg.channels),
                                cfg.use_new_attention_order)
            idx += 1
        h = _conv(sd, f"down_layer.{idx}.op", h, stride=2)    # Downsample(channels, True): conv s2 p1
        idx += 1
        inp_dim //= 2
    h = h.flatten(1)
    h = F.linear(h, sd["fc_layer.1.weight"], sd["fc_layer.1.bias"])
    h = F.batch_norm(h, sd["fc_layer.2.running_mean"], sd["fc_layer.2.running_var"], sd["fc_layer.2.weight"],
                     sd["fc_layer.2.bias"], training=False, eps=1e-5)
    h = F.gelu(h)
    out = F.linear(h, sd["final_mlp.weight"], sd["final_mlp.bias"])
    return out[:, :, None, None]


def configs_from_factory(image_size: int, num_channels: int, num_res_blocks: int, channel_mult: str = "",
                         learn_sigma: bool = False, attention_resolutions: str = "16", num_heads: int = 1,
                         num_head_channels: int = -1, num_heads_upsample: int = -1, use_scale_shift_norm: bool = False,
                         resblock_updown: bool = False, use_new_attention_order: bool = False, sigma_block: int = 2,
                         **_unused):
    """create_sigma_eps_model (src/script_util.py:136-206): note feat_layer is swallowed by **kwargs there."""
    if channel_mult == "":
        channel_mult = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4),
                        32: (1, 2, 2, 2)}[image_size]
    else:
        channel_mult = tuple(int(c) for c in channel_mult.split(","))
    att = tuple(image_size // int(r) for r in attention_resolutions.split(","))
    ucfg = AdmConfig(image_size=image_size, in_channels=3, model_channels=num_channels,
                     out_channels=6 if learn_sigma else 3, num_res_blocks=num_res_blocks, attention_resolutions=att,
                     channel_mult=channel_mult, num_heads=num_heads, num_head_channels=num_head_channels,
                     num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                     resblock_updown=resblock_updown, use_new_attention_order=use_new_attention_order, feat_layer=1)
    inp_channels = int(num_channels * channel_mult[-1])
    inp_dim = int(image_size * 0.5 ** (len(channel_mult) - 1))
    scfg = AdmSigmaConfig(dim=inp_dim, channels=inp_channels, n_blocks=sigma_block, num_heads=num_heads,
                          num_head_channels=num_head_channels, use_new_attention_order=use_new_attention_order)
    return ucfg, scfg, (inp_channels, inp_dim, inp_dim)
