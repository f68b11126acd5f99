"""EDM DDPM++ ``SongUNet`` and its sigma net on the HIP kernels (drop-in for src/edm_networks.py).

Only the configuration ``create_edm_sigma_eps_model`` builds is implemented
(src/script_util.py:243-262): positional embedding, 'standard' encoder/decoder,
resample_filter [1,1] (so down = 2x2 mean, up = nearest 2x), adaptive_scale False, one
attention head, skip_scale sqrt(1/2), GroupNorm(min(32,C/4) groups, eps 1e-6).

    UNetBlock (src/edm_networks.py:183-205)  GN+SiLU -> [2x2 mean | nearest-2x folded into the gather]
        -> 3x3 conv whose epilogue adds affine(emb) per (image, channel) -> GN+SiLU -> 3x3 conv with
        (+skip, * sqrt(1/2)) in its epilogue; attention: GN -> 1x1 qkv GEMM (interleaved q,k,v channel
        order and k/sqrt(C) folded into the weights) -> flash attention -> 1x1 proj (+x, * sqrt(1/2)).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import ops
from ._ext import ACT_NONE, ACT_SILU
from .hipnet import EmbBank, HipModule, Norm, SigmaHead, SpecBuilder, as_f32_cuda, first_conv_weight, pack

GN_EPS = 1e-6
SKIP_SCALE = float(np.sqrt(0.5))


def _groups(c):
    return min(32, c // 4)               # GroupNorm.__init__ (src/edm_networks.py:108)


def _spec_block(sb: SpecBuilder, p, cin, cout, emb_ch, attention, up=False, down=False):
    sb.norm(p + ".norm0", cin)
    sb.conv(p + ".conv0", cout, cin, 3)
    if up or down:
        sb.add(p + ".conv0.resample_filter", (1, 1, 2, 2))
    if emb_ch:
        sb.linear(p + ".affine", cout, emb_ch)
    sb.norm(p + ".norm1", cout)
    sb.conv(p + ".conv1", cout, cout, 3)
    if cout != cin or up or down:
        sb.conv(p + ".skip", cout, cin, 1)
        if up or down:
            sb.add(p + ".skip.resample_filter", (1, 1, 2, 2))
    if attention:
        sb.norm(p + ".norm2", cout)
        sb.conv(p + ".qkv", 3 * cout, cout, 1)
        sb.conv(p + ".proj", cout, cout, 1)


class _UNetBlock:
    def __init__(self, sd, p, dtype, device, bank: Optional[EmbBank], attention: bool, up=False, down=False, pure=False):
        cin = sd[p + ".norm0.weight"].shape[0]
        cout = sd[p + ".conv0.weight"].shape[0]
        self.n0 = Norm(sd, p + ".norm0", device, _groups(cin), GN_EPS)
        self.c0 = pack(sd, p + ".conv0", dtype, device)
        self.n1 = Norm(sd, p + ".norm1", device, _groups(cout), GN_EPS)
        self.c1 = pack(sd, p + ".conv1", dtype, device)
        self.skip = pack(sd, p + ".skip", dtype, device) if (p + ".skip.weight") in sd else None
        self.up, self.down, self.pure = up, down, pure
        self.emb_off = bank.add(sd[p + ".affine.weight"], sd[p + ".affine.bias"])[0] if bank is not None else None
        self.attn = None
        if attention:
            c = cout
            idx = torch.arange(3 * c)
            s_, c_ = idx // c, idx % c
            perm = c_ * 3 + s_                      # reshape(n, C, 3, T).unbind(2): channel = c*3 + {q,k,v} (:199-200)
            scale = torch.ones(3 * c, dtype=torch.float64)
            scale[: 2 * c] = float(c) ** -0.25      # k / sqrt(C) (:127) split evenly over q and k
            f, self.base2 = ops.attention_logit_scale(dtype)
            scale[:c] *= f
            self.attn = (Norm(sd, p + ".norm2", device, _groups(c), GN_EPS),
                         pack(sd, p + ".qkv", dtype, device, row_perm=perm, row_scale=scale),
                         pack(sd, p + ".proj", dtype, device))

    def __call__(self, x, x1, emb_all):
        both = None
        if self.skip is not None and not self.down and not self.up:
            both = self.n0.with_skip(x, self.skip, silu=True, x1=x1)                # norm0 and the skip projection from one read
        emb = None if self.emb_off is None else emb_all[:, self.emb_off:]
        res = None
        if both is not None:
            hn, res = both
            h = ops.conv2d(hn, self.c0, emb=emb)
        elif self.down:
            h = ops.conv2d(ops.avgpool2x2(self.n0(x, silu=True, x1=x1)), self.c0, emb=emb)
        else:                              # (small maps: the normalisation is applied by the convolution itself, hipnet.Norm.then_conv)
            h = self.n0.then_conv(x, self.c0, silu=True, x1=x1, upsample2x=self.up, emb=emb)
        # PureUNetBlock.forward feeds conv0's output straight into conv1 (src/edm_networks.py:944-945)
        if res is not None:
            pass
        elif self.skip is not None:
            xs, xs1 = x, x1
            if self.down:
                xs = ops.avgpool2x2(x)
            res = ops.conv2d(xs, self.skip, x1=xs1, upsample2x=self.up)
        else:
            res = x
        if self.pure:
            x = ops.conv2d(h, self.c1, res=res, out_scale=SKIP_SCALE)
        else:
            x = self.n1.then_conv(h, self.c1, silu=True, res=res, out_scale=SKIP_SCALE)
        if self.attn is not None:
            norm, qkv, proj = self.attn
            B, H, W, C = x.shape
            a = ops.attention(ops.conv2d(norm(x, silu=False), qkv).view(B, H * W, 3 * C), 1, base2=self.base2)
            x = ops.conv2d(a.view(B, H, W, C), proj, res=x, out_scale=SKIP_SCALE)
        return x


class SongUNet(HipModule):
    """src/edm_networks.py:732-909 (the second, plain-nn.Module definition that carries ``encode``)."""

    def __init__(self, img_resolution, in_channels, out_channels, label_dim=0, augment_dim=0, model_channels=128,
                 channel_mult=[1, 2, 2, 2], channel_mult_emb=4, num_blocks=4, attn_resolutions=[16], dropout=0.10,
                 label_dropout=0, embedding_type="positional", channel_mult_noise=1, encoder_type="standard",
                 decoder_type="standard", resample_filter=[1, 1], **kwargs):
        if embedding_type != "positional" or encoder_type != "standard" or decoder_type != "standard" \
                or list(resample_filter) != [1, 1] or label_dim != 0 or channel_mult_noise != 1:
            raise NotImplementedError("HIP SongUNet: only the DDPM++ configuration of create_edm_sigma_eps_model")
        self.img_resolution, self.in_channels, self.out_channels = img_resolution, in_channels, out_channels
        self.augment_dim, self.model_channels = augment_dim, model_channels
        self.channel_mult, self.num_blocks = tuple(channel_mult), num_blocks
        self.attn_resolutions = tuple(attn_resolutions)
        self.emb_channels = model_channels * channel_mult_emb
        super().__init__()

    def _layout(self):
        enc, dec = [], []
        cout = self.in_channels
        mc = self.model_channels
        for level, mult in enumerate(self.channel_mult):
            res = self.img_resolution >> level
            if level == 0:
                cin, cout = cout, mc
                enc.append((f"enc.{res}x{res}_conv", "conv", cin, cout, {}))
            else:
                enc.append((f"enc.{res}x{res}_down", "block", cout, cout, {"down": True}))
            for idx in range(self.num_blocks):
                cin, cout = cout, mc * mult
                enc.append((f"enc.{res}x{res}_block{idx}", "block", cin, cout, {"attention": res in self.attn_resolutions}))
        skips = [e[3] for e in enc]
        nlev = len(self.channel_mult)
        for level, mult in reversed(list(enumerate(self.channel_mult))):
            res = self.img_resolution >> level
            if level == nlev - 1:
                dec.append((f"dec.{res}x{res}_in0", "block", cout, cout, {"attention": True}))
                dec.append((f"dec.{res}x{res}_in1", "block", cout, cout, {}))
            else:
                dec.append((f"dec.{res}x{res}_up", "block", cout, cout, {"up": True}))
            for idx in range(self.num_blocks + 1):
                cin = cout + skips.pop()
                cout = mc * mult
                dec.append((f"dec.{res}x{res}_block{idx}", "block", cin, cout,
                            {"attention": idx == self.num_blocks and res in self.attn_resolutions, "cat": True}))
            if level == 0:
                dec.append((f"dec.{res}x{res}_aux_norm", "aux_norm", cout, cout, {}))
                dec.append((f"dec.{res}x{res}_aux_conv", "aux_conv", cout, self.out_channels, {}))
        return enc, dec

    def param_spec(self):
        sb = SpecBuilder()
        mc, E = self.model_channels, self.emb_channels
        if self.augment_dim:
            sb.linear("map_augment", mc, self.augment_dim, bias=False)
        sb.linear("map_layer0", E, mc)
        sb.linear("map_layer1", E, E)
        enc, dec = self._layout()
        for p, kind, cin, cout, kw in enc + dec:
            if kind in ("conv", "aux_conv"):
                sb.conv(p, cout, cin, 3)
            elif kind == "aux_norm":
                sb.norm(p, cin)
            else:
                _spec_block(sb, p, cin, cout, E, kw.get("attention", False), up=kw.get("up", False), down=kw.get("down", False))
        return sb.spec

    def _build(self, sd, device, dtype):
        P = type("Plan", (), {})()
        half = self.model_channels // 2
        fr = torch.arange(0, half, dtype=torch.float32) / (half - 1)         # endpoint=True (:220-222)
        P.freqs = ((1 / 10000) ** fr).to(device)
        P.m0 = pack(sd, "map_layer0", torch.float32, device)
        P.m1 = pack(sd, "map_layer1", torch.float32, device)
        bank = EmbBank()
        enc, dec = self._layout()
        P.enc, P.dec = [], []
        for p, kind, cin, cout, kw in enc:
            if kind == "conv":
                P.enc.append(("conv",) + first_conv_weight(sd, p, device))
            else:
                P.enc.append(("block", _UNetBlock(sd, p, dtype, device, bank, kw.get("attention", False), down=kw.get("down", False))))
        for p, kind, cin, cout, kw in dec:
            if kind == "aux_norm":
                P.aux_norm = Norm(sd, p, device, _groups(cin), GN_EPS)
            elif kind == "aux_conv":
                P.aux_conv = pack(sd, p, dtype, device)
            else:
                P.dec.append((bool(kw.get("cat")), _UNetBlock(sd, p, dtype, device, bank, kw.get("attention", False), up=kw.get("up", False))))
        bank.finalize(device, allow_split=ops.is16(dtype))
        P.bank = bank
        return P

    def run(self, x_nchw, noise_labels, mode="forward", in_scale=None, feat_nhwc=False):
        P = self.plan()
        with torch.cuda.device(self.device):
            # map_noise + the sin/cos swap (:837-838) == [sin || cos]; augment_labels is None on this path
            pe = ops.timestep_embedding(noise_labels, P.freqs, sin_first=True)
            e = ops.conv2d(pe, P.m0, act=ACT_SILU)
            e = ops.conv2d(e, P.m1, act=ACT_SILU)
            emb_all = P.bank(e)
            skips = []
            x = None
            for item in P.enc:
                if item[0] == "conv":
                    x = ops.conv_first(x_nchw, item[1], item[2], self.compute_dtype, in_scale=in_scale)
                else:
                    x = item[1](x, None, emb_all)
                skips.append(x)
            if mode == "encode":
                return x if feat_nhwc else ops.nhwc_to_nchw_f32(x)
            for cat, blk in P.dec:
                x = blk(x, skips.pop() if cat else None, emb_all)
            return ops.conv2d(P.aux_norm(x, silu=True), P.aux_conv, out_nchw_f32=True)

    def _prep(self, x, noise_labels):
        self._require_gpu()
        return as_f32_cuda(x, self.device), as_f32_cuda(noise_labels, self.device).reshape(-1)

    def forward(self, x, noise_labels, class_labels=None, augment_labels=None):
        assert class_labels is None and augment_labels is None
        return self.run(*self._prep(x, noise_labels), mode="forward")

    def encode(self, x, noise_labels, class_labels=None, augment_labels=None):
        assert class_labels is None and augment_labels is None
        return self.run(*self._prep(x, noise_labels), mode="encode")


class SigmaModel(HipModule):
    """src/edm_networks.py:979-1022: PureUNetBlock (attention on even blocks) + pad/conv-s2 downsample, SiLU head."""

    def __init__(self, dim=4, channels=64, n_blocks=2, out_dim=1, dropout=0.1, resample_filter=[1, 1]):
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
            blk = f"down_layer.{idx}"; idx += 1
            down = f"down_layer.{idx}"; idx += 1
            d //= 2
            out.append((pad, blk, i % 2 == 0, down))
        return out, d

    def param_spec(self):
        sb = SpecBuilder()
        c = self.channels
        layout, d = self._layout()
        for pad, blk, attn, down in layout:
            _spec_block(sb, blk, c, c, 0, attn)
            sb.conv(down + ".conv", c, c, 3)
        sb.linear("fc_layer.1", 128, c * d * d)
        sb.batchnorm("fc_layer.2", 128)
        sb.linear("final_mlp", 1, 128)
        return sb.spec

    def _build(self, sd, device, dtype):
        P = type("Plan", (), {})()
        layout, _ = self._layout()
        P.blocks = [(pad, _UNetBlock(sd, blk, dtype, device, None, attn, pure=True), pack(sd, down + ".conv", dtype, device))
                    for pad, blk, attn, down in layout]
        P.head = SigmaHead(sd, device, ACT_SILU, allow_split=ops.is16(dtype))
        return P

    def run_nhwc(self, h):
        P = self.plan()
        with torch.cuda.device(self.device):
            for pad, blk, down in P.blocks:
                if pad:
                    h = ops.pad_rb(h)
                h = blk(h, None, None)
                h = ops.conv2d(h, down, stride=2, pad=(0, 0), out_hw=(h.shape[1] // 2, h.shape[2] // 2))
            return P.head(h)

    def forward(self, feat):
        self._require_gpu()
        h = ops.nchw_f32_to_nhwc(as_f32_cuda(feat, self.device), self.compute_dtype)
        return self.run_nhwc(h).view(-1, 1, 1, 1)
