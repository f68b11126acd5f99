"""ADM (guided-diffusion) UNet and its sigma net on the HIP kernels.

Drop-in for ``src/unet_adm.py``'s ``UNetModel`` / ``SigmaModel``: same constructor arguments, the
same ``state_dict`` keys, ``model(x, t)``, ``model.encode(x, t)``, ``model.forward_and_encode(x, t)``,
``sigma_model(feat)``.  Internally activations are NHWC in the compute dtype and every op is a
kernel from libnlc_hip.so:

    ResBlock (src/unet_adm.py:236-256)   GroupNorm+SiLU kernel -> 3x3 implicit-GEMM conv (+bias)
                                          -> GroupNorm*(1+scale)+shift+SiLU kernel -> 3x3 conv with the
                                          skip connection added in its epilogue; torch.cat of the UNet
                                          skip is never materialised (two-source GN / conv);
                                          nearest-2x upsample is folded into the conv's gather.
    AttentionBlock (:299-305)            GroupNorm kernel -> 1x1 conv (q/k scale ch^-1/4 and the
                                          legacy/new channel order folded into the weights) -> flash
                                          attention kernel -> 1x1 conv with the residual in its epilogue.
    time_embed / emb_layers (:473-477)   sin/cos kernel + f32 GEMMs; all blocks' emb_layers are one GEMM.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from . import ops
from ._ext import ACT_GELU, ACT_NONE, ACT_SILU
from .hipnet import EmbBank, HipModule, Norm, SigmaHead, SpecBuilder, as_f32_cuda, first_conv_weight, f32, pack

GN_GROUPS, GN_EPS = 32, 1e-5       # GroupNorm32 (src/nn_util.py:93-100)


def _num_heads(num_heads, num_head_channels, channels):
    return num_heads if num_head_channels == -1 else channels // num_head_channels


def _spec_resblock(sb: SpecBuilder, p, cin, cout, emb_ch, scale_shift, with_emb=True):
    sb.norm(p + ".in_layers.0", cin)
    sb.conv(p + ".in_layers.2", cout, cin, 3)
    if with_emb:
        sb.linear(p + ".emb_layers.1", 2 * cout if scale_shift else cout, emb_ch)
    sb.norm(p + ".out_layers.0", cout)
    sb.conv(p + ".out_layers.3", cout, cout, 3)
    if cin != cout:
        sb.conv(p + ".skip_connection", cout, cin, 1)


def _spec_attention(sb: SpecBuilder, p, ch):
    sb.norm(p + ".norm", ch)
    sb.conv(p + ".qkv", 3 * ch, ch, 1, dims=1)
    sb.conv(p + ".proj_out", ch, ch, 1, dims=1)


class _ResBlock:
    def __init__(self, sd, p, dtype, device, bank: Optional[EmbBank], scale_shift: bool, up=False, down=False):
        self.n1 = Norm(sd, p + ".in_layers.0", device, GN_GROUPS, GN_EPS)
        self.c1 = pack(sd, p + ".in_layers.2", dtype, device)
        self.n2 = Norm(sd, p + ".out_layers.0", device, GN_GROUPS, GN_EPS)
        self.c2 = pack(sd, p + ".out_layers.3", dtype, device)
        self.skip = pack(sd, p + ".skip_connection", dtype, device) if (p + ".skip_connection.weight") in sd else None
        self.up, self.down, self.scale_shift = up, down, scale_shift
        self.cout = self.c1.Cout
        self.emb_off = None
        if bank is not None:
            self.emb_off, _ = bank.add(sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])

    def __call__(self, x, x1, emb_all):
        emb = None if self.emb_off is None else emb_all[:, self.emb_off:]
        c1_emb = None if (self.scale_shift or emb is None) else emb                      # h + emb_out (:254)
        if self.down:                                          # AvgPool2d on both branches (:193-195): norm, then pool, then conv
            if x1 is not None:
                raise NotImplementedError("down-sampling ResBlock on a concatenated input (the reference has none)")
            h, x = self.n1.pooled(x, silu=True)                # one read of x for both branches, no full-resolution h
            h = ops.conv2d(h, self.c1, emb=c1_emb)
        res = None
        if not self.down:
            both = self.n1.with_skip(x, self.skip, silu=True, x1=x1) if (self.skip is not None and not self.up) else None
            if both is not None:                               # in_layers(x) and skip_connection(x) (:236, :256) from one read of x
                hn, res = both
                h = ops.conv2d(hn, self.c1, emb=c1_emb)
            else:                                              # GroupNorm+SiLU applied inside the conv's LDS prologue when it can be
                h = self.n1.then_conv(x, self.c1, silu=True, x1=x1, upsample2x=self.up, emb=c1_emb)
        res_ups = False
        if self.up and self.skip is None:
            res_ups = True                                     # x_upd(x): read nearest-2x upsampled by the conv's epilogue, never materialised
        elif self.up:
            x = ops.upsample2x(x)
        if res is not None:
            pass
        elif self.skip is not None:
            res = ops.conv2d(x, self.skip, x1=x1)
        else:
            res = x
        if self.scale_shift and emb is not None:
            return self.n2.then_conv(h, self.c2, silu=True, scale=emb[:, :self.cout], shift=emb[:, self.cout:2 * self.cout], res=res,
                                     res_upsample2x=res_ups)
        return self.n2.then_conv(h, self.c2, silu=True, res=res, res_upsample2x=res_ups)


class _Attention:
    def __init__(self, sd, p, dtype, device, heads: int, new_order: bool):
        ch3 = sd[p + ".qkv.weight"].shape[0]
        c = ch3 // 3
        d = c // heads
        self.heads = heads
        self.norm = Norm(sd, p + ".norm", device, GN_GROUPS, GN_EPS)
        idx = torch.arange(ch3)
        if new_order:                        # [q|k|v][head][ch] already canonical (:380-388)
            perm = idx
        else:                                # legacy [head][q|k|v][ch] (:347) -> [q|k|v][head][ch]
            s_, h_, c_ = idx // (heads * d), (idx // d) % heads, idx % d
            perm = h_ * (3 * d) + s_ * d + c_
        scale = torch.ones(ch3, dtype=torch.float64)
        scale[: 2 * c] = 1.0 / math.sqrt(math.sqrt(d))      # q and k each scaled by ch^-1/4 (:348-351)
        f, self.base2 = ops.attention_logit_scale(dtype)
        scale[:c] *= f                                       # 16-bit models: log2(e) rides along on q, the kernel uses 2^x
        self.qkv = pack(sd, p + ".qkv", dtype, device, row_perm=perm, row_scale=scale)
        self.proj = pack(sd, p + ".proj_out", dtype, device)

    def __call__(self, x):
        B, H, W, C = x.shape
        qkv = ops.conv2d(self.norm(x, silu=False), self.qkv)
        a = ops.attention(qkv.view(B, H * W, 3 * C), self.heads, base2=self.base2)
        return ops.conv2d(a.view(B, H, W, C), self.proj, res=x)


class UNetModel(HipModule):
    """src/unet_adm.py:396-731, unconditional or class-conditional (``model(x, t, y)``: the label embedding row is added
    to the timestep embedding, :479-480,652-654 - here in the epilogue of time_embed's second Linear)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False, feat_layer=1):
        if dims != 2:
            raise NotImplementedError("HIP UNetModel: 2-D only (SURVEY.md §8 scope)")
        self.num_classes = num_classes
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = tuple(channel_mult)
        self.conv_resample = conv_resample
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample
        self.use_scale_shift_norm, self.resblock_updown = use_scale_shift_norm, resblock_updown
        self.use_new_attention_order, self.feat_layer = use_new_attention_order, feat_layer
        self.dropout = dropout
        super().__init__()
        if use_fp16:
            self.convert_to_fp16()

    # ---- structure walk shared by the spec and the plan (src/unet_adm.py:482-618) ------------
    def _layout(self):
        mc, E = self.model_channels, self.model_channels * 4
        ch = int(self.channel_mult[0] * mc)
        inp = [("conv_in", "input_blocks.0.0", self.in_channels, ch, {})]
        chans, ds, n = [ch], 1, 1
        blocks_in: List[List[tuple]] = [[inp[0]]]
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks):
                out_ch = int(mult * mc)
                layers = [("res", f"input_blocks.{n}.0", ch, out_ch, {})]
                ch = out_ch
                if ds in self.attention_resolutions:
                    layers.append(("attn", f"input_blocks.{n}.1", ch, ch, {"heads": _num_heads(self.num_heads, self.num_head_channels, ch)}))
                blocks_in.append(layers)
                chans.append(ch)
                n += 1
            if level != len(self.channel_mult) - 1:
                if self.resblock_updown:
                    blocks_in.append([("res", f"input_blocks.{n}.0", ch, ch, {"down": True})])
                else:
                    blocks_in.append([("downsample", f"input_blocks.{n}.0", ch, ch, {})])
                chans.append(ch)
                n += 1
                ds *= 2
        mid = [("res", "middle_block.0", ch, ch, {}),
               ("attn", "middle_block.1", ch, ch, {"heads": _num_heads(self.num_heads, self.num_head_channels, ch)}),
               ("res", "middle_block.2", ch, ch, {})]
        blocks_out: List[List[tuple]] = []
        n = 0
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(self.num_res_blocks + 1):
                ich = chans.pop()
                out_ch = int(mc * mult)
                layers = [("res", f"output_blocks.{n}.0", ch + ich, out_ch, {"skip_in": ich})]
                ch = out_ch
                j = 1
                if ds in self.attention_resolutions:
                    layers.append(("attn", f"output_blocks.{n}.{j}", ch, ch,
                                   {"heads": _num_heads(self.num_heads_upsample, self.num_head_channels, ch)}))
                    j += 1
                if level and i == self.num_res_blocks:
                    if self.resblock_updown:
                        layers.append(("res", f"output_blocks.{n}.{j}", ch, ch, {"up": True}))
                    else:
                        layers.append(("upsample", f"output_blocks.{n}.{j}", ch, ch, {}))
                    ds //= 2
                blocks_out.append(layers)
                n += 1
        return blocks_in, mid, blocks_out, ch, E

    def param_spec(self):
        sb = SpecBuilder()
        mc = self.model_channels
        blocks_in, mid, blocks_out, ch_last, E = self._layout()
        sb.linear("time_embed.0", E, mc)
        sb.linear("time_embed.2", E, E)
        if self.num_classes is not None:
            sb.add("label_emb.weight", (self.num_classes, E))          # nn.Embedding (src/unet_adm.py:479-480)
        for layers in blocks_in + [mid] + blocks_out:
            for kind, p, cin, cout, kw in layers:
                if kind == "conv_in":
                    sb.conv(p, cout, cin, 3)
                elif kind == "res":
                    _spec_resblock(sb, p, cin, cout, E, self.use_scale_shift_norm)
                elif kind == "attn":
                    _spec_attention(sb, p, cin)
                elif kind == "downsample" and self.conv_resample:
                    sb.conv(p + ".op", cout, cin, 3)
                elif kind == "upsample" and self.conv_resample:
                    sb.conv(p + ".conv", cout, cin, 3)
        sb.norm("out.0", ch_last)
        sb.conv("out.2", self.out_channels, int(self.channel_mult[0] * mc), 3)
        return sb.spec

    # ---- device plan ---------------------------------------------------------------------------
    def _build(self, sd, device, dtype):
        P = type("Plan", (), {})()
        blocks_in, mid, blocks_out, ch_last, E = self._layout()
        half = self.model_channels // 2
        # frequency table computed exactly as timestep_embedding does (src/nn_util.py:113-116)
        P.freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(device)
        P.te0 = pack(sd, "time_embed.0", torch.float32, device)
        P.te2 = pack(sd, "time_embed.2", torch.float32, device)
        P.label = f32(sd["label_emb.weight"], device) if self.num_classes is not None else None
        bank = EmbBank()

        def make(layers):
            out = []
            for kind, p, cin, cout, kw in layers:
                if kind == "conv_in":
                    out.append(("conv_in",) + first_conv_weight(sd, p, device))
                elif kind == "res":
                    out.append(("res", _ResBlock(sd, p, dtype, device, bank, self.use_scale_shift_norm,
                                                 up=kw.get("up", False), down=kw.get("down", False))))
                elif kind == "attn":
                    out.append(("attn", _Attention(sd, p, dtype, device, kw["heads"], self.use_new_attention_order)))
                elif kind == "downsample":
                    out.append(("downsample", pack(sd, p + ".op", dtype, device) if self.conv_resample else None))
                elif kind == "upsample":
                    out.append(("upsample", pack(sd, p + ".conv", dtype, device) if self.conv_resample else None))
            return out
        P.inp = [make(l) for l in blocks_in]
        P.mid = make(mid)
        P.out = [make(l) for l in blocks_out]
        bank.finalize(device, allow_split=ops.is16(dtype))
        P.bank = bank
        P.out_norm = Norm(sd, "out.0", device, GN_GROUPS, GN_EPS)
        P.out_conv = pack(sd, "out.2", dtype, device)
        return P

    # ---- execution -----------------------------------------------------------------------------
    def _emb(self, P, t, y=None):
        temb = ops.timestep_embedding(t, P.freqs, sin_first=False)          # cos || sin
        sp = ops.is16(self.compute_dtype)                            # f32 GEMMs of a bf16 model may split K
        e = ops.conv2d(temb, P.te0, act=ACT_SILU, allow_split=sp)            # Linear -> SiLU
        # emb = time_embed(...) + label_emb(y) (src/unet_adm.py:650-654): the gathered embedding rows ride in the GEMM's
        # per-image add; every consumer starts with SiLU(emb), applied here once
        rows = None if y is None else P.label.index_select(0, y)
        e = ops.conv2d(e, P.te2, emb=rows, act=ACT_SILU, allow_split=sp)
        return P.bank(e)

    @staticmethod
    def _run(layers, h, x1, emb_all, x_nchw=None, in_scale=None, dtype=None):
        for item in layers:
            kind = item[0]
            if kind == "conv_in":
                h = ops.conv_first(x_nchw, item[1], item[2], dtype, in_scale=in_scale)
            elif kind == "res":
                h = item[1](h, x1, emb_all)
                x1 = None
            elif kind == "attn":
                h = item[1](h)
            elif kind == "downsample":
                h = ops.conv2d(h, item[1], stride=2, pad=(1, 1)) if item[1] is not None else ops.avgpool2x2(h)
            elif kind == "upsample":
                h = ops.conv2d(h, item[1], upsample2x=True) if item[1] is not None else ops.upsample2x(h)
        return h

    def run(self, x_nchw: torch.Tensor, t: torch.Tensor, mode: str = "forward", in_scale: Optional[torch.Tensor] = None,
            feat_nhwc: bool = False, y: Optional[torch.Tensor] = None):
        """x_nchw: f32 [B,C,H,W] on the GPU; t: f32 [B].  in_scale[b] multiplies the input (convert_coordinate).

        mode 'forward' -> eps_out NCHW f32 ; 'encode' -> feat ; 'both' -> (out, feat).
        feat is NCHW f32 (the reference's format) unless feat_nhwc (internal fast path to the sigma net).
        """
        P = self.plan()
        dt = self.compute_dtype
        if (y is not None) != (self.num_classes is not None):
            raise AssertionError("must specify y if and only if the model is class-conditional")     # src/unet_adm.py:645-647
        with torch.cuda.device(self.device):
            emb_all = self._emb(P, t, y)
            hs = []
            h = None
            for layers in P.inp:
                h = self._run(layers, h, None, emb_all, x_nchw=x_nchw, in_scale=in_scale, dtype=dt)
                hs.append(h)
            feat = None
            if mode == "encode":
                feat = h if self.feat_layer == 0 else self._run(P.mid, h, None, emb_all)
                return feat if feat_nhwc else ops.nhwc_to_nchw_f32(feat)
            if self.feat_layer == 0:
                feat = h
            h = self._run(P.mid, h, None, emb_all)
            if feat is None:
                feat = h
            for layers in P.out:
                h = self._run(layers, h, hs.pop(), emb_all)
            out = P.out_norm.then_conv(h, P.out_conv, silu=True, out_nchw_f32=True)
            if mode == "forward":
                return out
            return out, (feat if feat_nhwc else ops.nhwc_to_nchw_f32(feat))

    def _prep(self, x, timesteps):
        self._require_gpu()
        x = as_f32_cuda(x, self.device)
        t = as_f32_cuda(timesteps, self.device).reshape(-1)
        if t.numel() == 1 and x.shape[0] > 1:
            t = t.expand(x.shape[0]).contiguous()
        return x, t

    def _labels(self, y, n):
        if y is None:
            return None
        y = torch.as_tensor(y).to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
        assert y.shape == (n,)                                               # src/unet_adm.py:653
        return y

    def forward(self, x, timesteps, y=None):
        x, t = self._prep(x, timesteps)
        return self.run(x, t, mode="forward", y=self._labels(y, x.shape[0]))

    def encode(self, x, timesteps, y=None):
        x, t = self._prep(x, timesteps)
        return self.run(x, t, mode="encode", y=self._labels(y, x.shape[0]))

    def forward_and_encode(self, x, timesteps, y=None):
        x, t = self._prep(x, timesteps)
        return self.run(x, t, mode="both", y=self._labels(y, x.shape[0]))


class SigmaModel(HipModule):
    """src/unet_adm.py:1029-1083: (pad) -> PureResNetBlock -> [Attention on block 0] -> conv s2 p1, then the f32 head."""

    def __init__(self, dim=4, channels=64, n_blocks=2, out_dim=1, dropout=0.1, num_heads=1, num_head_channels=-1,
                 use_new_attention_order=False, use_checkpoint=False, use_fp16=False):
        if out_dim != 1:
            raise NotImplementedError("SigmaModel: out_dim must be 1")
        self.dim, self.channels, self.n_blocks = dim, channels, n_blocks
        self.heads = _num_heads(num_heads, num_head_channels, channels)
        self.new_order = use_new_attention_order
        super().__init__()
        if use_fp16:
            self.convert_to_fp16()

    def _layout(self):
        out, idx, d = [], 0, self.dim
        for i in range(self.n_blocks):
            pad = d % 2 != 0
            if pad:
                d += 1
            idx += 1
            res = f"down_layer.{idx}"; idx += 1
            attn = None
            if i == 0:
                attn = f"down_layer.{idx}"; idx += 1
            down = f"down_layer.{idx}"; idx += 1
            d //= 2
            out.append((pad, res, attn, down))
        return out, d

    def param_spec(self):
        sb = SpecBuilder()
        c = self.channels
        layout, d = self._layout()
        for pad, res, attn, down in layout:
            _spec_resblock(sb, res, c, c, 0, False, with_emb=False)
            if attn:
                _spec_attention(sb, attn, c)
            sb.conv(down + ".op", c, c, 3)
        sb.linear("fc_layer.1", 128, c * d * d)
        sb.batchnorm("fc_layer.2", 128)
        sb.linear("final_mlp", 1, 128)
        return sb.spec

    def _build(self, sd, device, dtype):
        P = type("Plan", (), {})()
        layout, _ = self._layout()
        P.blocks = []
        for pad, res, attn, down in layout:
            P.blocks.append((pad, _ResBlock(sd, res, dtype, device, None, False),
                             _Attention(sd, attn, dtype, device, self.heads, self.new_order) if attn else None,
                             pack(sd, down + ".op", dtype, device)))
        P.head = SigmaHead(sd, device, ACT_GELU, allow_split=ops.is16(dtype))
        return P

    def run_nhwc(self, h: torch.Tensor) -> torch.Tensor:
        """feat NHWC (compute dtype) -> r[b] f32."""
        P = self.plan()
        with torch.cuda.device(self.device):
            for pad, res, attn, down in P.blocks:
                if pad:
                    h = ops.pad_rb(h)
                h = res(h, None, None)
                if attn is not None:
                    h = attn(h)
                h = ops.conv2d(h, down, stride=2, pad=(1, 1))
            return P.head(h)

    def forward(self, feat):
        """feat: NCHW f32 as the reference passes it -> (B,1,1,1) f32."""
        self._require_gpu()
        x = as_f32_cuda(feat, self.device)
        h = ops.nchw_f32_to_nhwc(x, self.compute_dtype)
        return self.run_nhwc(h).view(-1, 1, 1, 1)
