"""DDPM/DDIM "simple" UNet and its sigma net on the HIP kernels (drop-in for src/unet_simple.py).

    ResnetBlock (src/unet_simple.py:115-134)  GN(eps 1e-6)+SiLU kernel -> 3x3 conv whose epilogue adds
                                               temb_proj(SiLU(temb)) per (image, channel) -> GN+SiLU ->
                                               3x3 conv with the (1x1-projected) shortcut added in its epilogue.
    AttnBlock (:164-189)                       GN -> ONE 1x1 GEMM for q|k|v (c^-1/2 split as c^-1/4 on q and k)
                                               -> single-head flash attention -> 1x1 conv + residual.
    Downsample (:67-74)                        conv s2 p0 whose gather zero-fills the right/bottom tap
                                               (the explicit F.pad is not materialised).
    Upsample (:47-52)                          nearest 2x folded into the following conv's gather.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops
from ._ext import ACT_GELU, ACT_NONE, ACT_SILU
from .hipnet import EmbBank, HipModule, Norm, SigmaHead, SpecBuilder, as_f32_cuda, first_conv_weight, pack, pack_w

GN_GROUPS, GN_EPS = 32, 1e-6       # Normalize (src/unet_simple.py:32-33)


def _spec_resnet(sb: SpecBuilder, p, cin, cout, temb_ch):
    sb.norm(p + ".norm1", cin)
    sb.conv(p + ".conv1", cout, cin, 3)
    if temb_ch:
        sb.linear(p + ".temb_proj", cout, temb_ch)
    sb.norm(p + ".norm2", cout)
    sb.conv(p + ".conv2", cout, cout, 3)
    if cin != cout:
        sb.conv(p + ".nin_shortcut", cout, cin, 1)


def _spec_attn(sb: SpecBuilder, p, c):
    sb.norm(p + ".norm", c)
    for n in ("q", "k", "v", "proj_out"):
        sb.conv(f"{p}.{n}", c, c, 1)


class _ResnetBlock:
    def __init__(self, sd, p, dtype, device, bank: Optional[EmbBank]):
        self.n1 = Norm(sd, p + ".norm1", device, GN_GROUPS, GN_EPS)
        self.c1 = pack(sd, p + ".conv1", dtype, device)
        self.n2 = Norm(sd, p + ".norm2", device, GN_GROUPS, GN_EPS)
        self.c2 = pack(sd, p + ".conv2", dtype, device)
        self.nin = pack(sd, p + ".nin_shortcut", dtype, device) if (p + ".nin_shortcut.weight") in sd else None
        self.emb_off = bank.add(sd[p + ".temb_proj.weight"], sd[p + ".temb_proj.bias"])[0] if bank is not None else None

    def __call__(self, x, x1, emb_all):
        both = self.n1.with_skip(x, self.nin, silu=True, x1=x1) if self.nin is not None else None      # norm1 and nin_shortcut: one read
        emb = None if self.emb_off is None else emb_all[:, self.emb_off:]
        if both is not None:
            hn, res = both
            h = ops.conv2d(hn, self.c1, emb=emb)
        else:                              # (small maps: the normalisation is applied by the convolution itself, hipnet.Norm.then_conv)
            h, res = self.n1.then_conv(x, self.c1, silu=True, x1=x1, emb=emb), None
        if res is None:
            res = ops.conv2d(x, self.nin, x1=x1) if self.nin is not None else x
        return self.n2.then_conv(h, self.c2, silu=True, res=res)


class _AttnBlock:
    def __init__(self, sd, p, dtype, device):
        c = sd[p + ".q.weight"].shape[0]
        self.norm = Norm(sd, p + ".norm", device, GN_GROUPS, GN_EPS)
        w = torch.cat([sd[f"{p}.{n}.weight"] for n in ("q", "k", "v")], 0)
        b = torch.cat([sd[f"{p}.{n}.bias"] for n in ("q", "k", "v")], 0)
        scale = torch.ones(3 * c, dtype=torch.float64)
        scale[: 2 * c] = float(c) ** -0.25                  # w_ * c^-1/2 (:177) split evenly over q and k
        f, self.base2 = ops.attention_logit_scale(dtype)
        scale[:c] *= f
        self.qkv = pack_w(w, b, dtype, device, row_scale=scale)
        self.proj = pack(sd, p + ".proj_out", dtype, device)

    def __call__(self, x):
        B, H, W, C = x.shape
        qkv = ops.conv2d(self.norm(x, silu=False), self.qkv)
        a = ops.attention(qkv.view(B, H * W, 3 * C), 1, base2=self.base2)
        return ops.conv2d(a.view(B, H, W, C), self.proj, res=x)


class Model(HipModule):
    """src/unet_simple.py:192-423; ``config`` is the nested namespace the entry points build from YAML."""

    def __init__(self, config):
        m = config.model
        self.config = config
        self.ch, self.out_ch, self.ch_mult = m.ch, m.out_ch, tuple(m.ch_mult)
        self.num_res_blocks = m.num_res_blocks
        self.attn_resolutions = tuple(m.attn_resolutions)
        self.in_channels = m.in_channels
        self.resolution = config.data.image_size
        self.resamp_with_conv = m.resamp_with_conv
        self.feat_layer = getattr(m, "feat_layer", 1)
        self.temb_ch = self.ch * 4
        self.num_resolutions = len(self.ch_mult)
        if getattr(m, "type", "simple") == "bayesian":
            raise NotImplementedError("bayesian logvar head is not on the sampling path")
        super().__init__()

    def _layout(self):
        ch, nres = self.ch, self.num_resolutions
        in_mult = (1,) + self.ch_mult
        res = self.resolution
        down = []
        block_in = None
        for lvl in range(nres):
            block_in, block_out = ch * in_mult[lvl], ch * self.ch_mult[lvl]
            blocks = []
            for b in range(self.num_res_blocks):
                blocks.append((f"down.{lvl}.block.{b}", block_in, block_out, f"down.{lvl}.attn.{b}" if res in self.attn_resolutions else None))
                block_in = block_out
            ds = None
            if lvl != nres - 1:
                ds = f"down.{lvl}.downsample"
                res //= 2
            down.append((blocks, ds, block_in))
        mid_ch = block_in
        up = []
        for lvl in reversed(range(nres)):
            block_out, skip_in = ch * self.ch_mult[lvl], ch * self.ch_mult[lvl]
            blocks = []
            for b in range(self.num_res_blocks + 1):
                if b == self.num_res_blocks:
                    skip_in = ch * in_mult[lvl]
                blocks.append((f"up.{lvl}.block.{b}", block_in + skip_in, block_out, f"up.{lvl}.attn.{b}" if res in self.attn_resolutions else None))
                block_in = block_out
            us = None
            if lvl != 0:
                us = f"up.{lvl}.upsample"
                res *= 2
            up.append((lvl, blocks, us, block_in))
        return down, mid_ch, up, block_in

    def param_spec(self):
        sb = SpecBuilder()
        down, mid_ch, up, last = self._layout()
        sb.linear("temb.dense.0", self.temb_ch, self.ch)
        sb.linear("temb.dense.1", self.temb_ch, self.temb_ch)
        sb.conv("conv_in", self.ch, self.in_channels, 3)
        for blocks, ds, cin in down:
            for p, ci, co, attn in blocks:
                _spec_resnet(sb, p, ci, co, self.temb_ch)
            for p, ci, co, attn in blocks:            # ModuleList order: all blocks, then all attns
                if attn:
                    _spec_attn(sb, attn, co)
            if ds and self.resamp_with_conv:
                sb.conv(ds + ".conv", cin, cin, 3)
        _spec_resnet(sb, "mid.block_1", mid_ch, mid_ch, self.temb_ch)
        _spec_attn(sb, "mid.attn_1", mid_ch)
        _spec_resnet(sb, "mid.block_2", mid_ch, mid_ch, self.temb_ch)
        for lvl, blocks, us, cin in sorted(up, key=lambda u: u[0]):       # self.up.insert(0, ...) -> level order
            for p, ci, co, attn in blocks:
                _spec_resnet(sb, p, ci, co, self.temb_ch)
            for p, ci, co, attn in blocks:
                if attn:
                    _spec_attn(sb, attn, co)
            if us and self.resamp_with_conv:
                sb.conv(us + ".conv", cin, cin, 3)
        sb.norm("norm_out", last)
        sb.conv("conv_out", self.out_ch, last, 3)
        return sb.spec

    def _build(self, sd, device, dtype):
        P = type("Plan", (), {})()
        down, mid_ch, up, last = self._layout()
        half = self.ch // 2
        k = math.log(10000) / (half - 1)                                   # src/unet_simple.py:16-18
        P.freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -k).to(device)
        P.d0 = pack(sd, "temb.dense.0", torch.float32, device)
        P.d1 = pack(sd, "temb.dense.1", torch.float32, device)
        P.conv_in = first_conv_weight(sd, "conv_in", device)
        bank = EmbBank()
        mk = lambda p: _ResnetBlock(sd, p, dtype, device, bank)
        P.down = [([(mk(p), _AttnBlock(sd, a, dtype, device) if a else None) for p, ci, co, a in blocks],
                   (pack(sd, ds + ".conv", dtype, device) if self.resamp_with_conv else "pool") if ds else None)
                  for blocks, ds, cin in down]
        P.mid1, P.mid_attn, P.mid2 = mk("mid.block_1"), _AttnBlock(sd, "mid.attn_1", dtype, device), mk("mid.block_2")
        P.up = [([(mk(p), _AttnBlock(sd, a, dtype, device) if a else None) for p, ci, co, a in blocks],
                 (pack(sd, us + ".conv", dtype, device) if self.resamp_with_conv else "nearest") if us else None)
                for lvl, blocks, us, cin in up]
        bank.finalize(device, allow_split=ops.is16(dtype))
        P.bank = bank
        P.norm_out = Norm(sd, "norm_out", device, GN_GROUPS, GN_EPS)
        P.conv_out = pack(sd, "conv_out", dtype, device)
        return P

    def run(self, x_nchw, t, mode="forward", in_scale=None, feat_nhwc=False):
        assert x_nchw.shape[2] == x_nchw.shape[3] == self.resolution
        P = self.plan()
        with torch.cuda.device(self.device):
            temb = ops.timestep_embedding(t, P.freqs, sin_first=True)
            e = ops.conv2d(temb, P.d0, act=ACT_SILU)
            e = ops.conv2d(e, P.d1, act=ACT_SILU)              # every consumer is temb_proj(nonlinearity(temb))
            emb_all = P.bank(e)
            hs = [ops.conv_first(x_nchw, P.conv_in[0], P.conv_in[1], self.compute_dtype, in_scale=in_scale)]
            for blocks, ds in P.down:
                for blk, attn in blocks:
                    h = blk(hs[-1], None, emb_all)
                    if attn is not None:
                        h = attn(h)
                    hs.append(h)
                if ds is not None:
                    if ds == "pool":
                        hs.append(ops.avgpool2x2(hs[-1]))
                    else:                                       # pad (0,1,0,1) + conv s2 p0, pad folded into the gather
                        x = hs[-1]
                        hs.append(ops.conv2d(x, ds, stride=2, pad=(0, 0), out_hw=(x.shape[1] // 2, x.shape[2] // 2)))
            h = P.mid1(hs[-1], None, emb_all)
            if mode == "encode":
                feat = P.mid_attn(h)
                if self.feat_layer != 0:
                    feat = P.mid2(feat, None, emb_all)
                return feat if feat_nhwc else ops.nhwc_to_nchw_f32(feat)
            h = P.mid_attn(h)
            feat = h if self.feat_layer == 0 else None
            h = P.mid2(h, None, emb_all)
            if feat is None:
                feat = h
            for blocks, us in P.up:
                for blk, attn in blocks:
                    h = blk(h, hs.pop(), emb_all)
                    if attn is not None:
                        h = attn(h)
                if us is not None:
                    h = ops.upsample2x(h) if us == "nearest" else ops.conv2d(h, us, upsample2x=True)
            out = ops.conv2d(P.norm_out(h, silu=True), P.conv_out, out_nchw_f32=True)
            if mode == "forward":
                return out
            return out, (feat if feat_nhwc else ops.nhwc_to_nchw_f32(feat))

    def _prep(self, x, t):
        self._require_gpu()
        x = as_f32_cuda(x, self.device)
        t = as_f32_cuda(t, self.device).reshape(-1)
        if t.numel() == 1 and x.shape[0] > 1:
            t = t.expand(x.shape[0]).contiguous()
        return x, t

    def forward(self, x, t):
        return self.run(*self._prep(x, t), mode="forward")

    def encode(self, x, t):
        return self.run(*self._prep(x, t), mode="encode")

    def forward_and_encode(self, x, t):
        return self.run(*self._prep(x, t), mode="both")


class SigmaModel(HipModule):
    """src/unet_simple.py:481-517."""

    def __init__(self, dim=4, channels=64, n_blocks=2, out_dim=1, dropout=0.1):
        if out_dim != 1:
            raise NotImplementedError("SigmaModel: out_dim must be 1")
        self.dim, self.channels, self.n_blocks = dim, channels, n_blocks
        super().__init__()

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
            _spec_resnet(sb, res, c, c, 0)
            if attn:
                _spec_attn(sb, attn, c)
            sb.conv(down + ".conv", c, c, 3)
        sb.linear("fc_layer.1", 128, c * d * d)
        sb.batchnorm("fc_layer.2", 128)
        sb.linear("final_mlp", 1, 128)
        return sb.spec

    def _build(self, sd, device, dtype):
        P = type("Plan", (), {})()
        layout, _ = self._layout()
        P.blocks = [(pad, _ResnetBlock(sd, res, dtype, device, None), _AttnBlock(sd, attn, dtype, device) if attn else None,
                     pack(sd, down + ".conv", dtype, device)) for pad, res, attn, down in layout]
        P.head = SigmaHead(sd, device, ACT_GELU, allow_split=ops.is16(dtype))
        return P

    def run_nhwc(self, h):
        P = self.plan()
        with torch.cuda.device(self.device):
            for pad, res, attn, down in P.blocks:
                if pad:
                    h = ops.pad_rb(h)
                h = res(h, None, None)
                if attn is not None:
                    h = attn(h)
                h = ops.conv2d(h, down, stride=2, pad=(0, 0), out_hw=(h.shape[1] // 2, h.shape[2] // 2))
            return P.head(h)

    def forward(self, feat):
        self._require_gpu()
        h = ops.nchw_f32_to_nhwc(as_f32_cuda(feat, self.device), self.compute_dtype)
        return self.run_nhwc(h).view(-1, 1, 1, 1)
