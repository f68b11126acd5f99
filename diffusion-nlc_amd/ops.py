"""Tensor -> raw-pointer shims over the C ABI (include/nlc_hip.h).

PyTorch is used here only as the owner of device memory and of the HIP stream; every
arithmetic op below is a hand-written gfx950 kernel in libnlc_hip.so.  Activations are
channels-last ``[B, H, W, C]`` tensors in the compute dtype (torch.float32, torch.bfloat16 or torch.float16).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _ext
from ._ext import (ACT_GELU, ACT_NONE, ACT_SILU, MATH_F16X3, MATH_NATIVE, NLC_BF16, NLC_F16, NLC_F32, OUT_NCHW_F32, OUT_NHWC,
                   ConvDesc, SchedDesc, check)


def on_device(fn):
    """Method decorator: run ``fn`` with ``self.device`` as the current HIP device.  Every shim below launches on
    ``torch.cuda.current_stream()`` - the CURRENT device's stream - so an object bound to cuda:K must make K current
    around its launches (and around the allocations feeding them), whatever device the caller left selected."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        dev = torch.device(getattr(self, "device", "cpu"))
        if dev.type == "cuda":
            with torch.cuda.device(dev):
                return fn(self, *a, **k)
        return fn(self, *a, **k)
    return wrapper


def dtype_enum(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return NLC_F32
    if dtype == torch.bfloat16:
        return NLC_BF16
    if dtype == torch.float16:
        return NLC_F16
    raise TypeError(f"compute dtype must be float32, bfloat16 or float16, got {dtype}")


def is16(dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16)


MATH_MODES = {"native": MATH_NATIVE, "f16x3": MATH_F16X3}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise _ext.NlcError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
@dataclass
class PackedConv:
    """A convolution / linear weight packed for nlc_conv2d: [Cout_pad][KH*KW][Cin_pad]."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]      # f32 [Cout] or None
    Cin: int
    Cout: int
    KH: int
    KW: int
    Cin_pad: int
    Cout_pad: int
    dtype: torch.dtype
    math: int = MATH_NATIVE           # MATH_F16X3: f32 tensor holding (hi, lo) f16 halves - nlc_conv2d must be told (desc.math)
    w_scale: Optional[torch.Tensor] = None   # MATH_F16X3: f32 [Cout_pad], the power-of-two factor per output row (desc.w_scale)


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype, device,
              row_perm: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
              col_perm: Optional[torch.Tensor] = None, bias_add: Optional[torch.Tensor] = None,
              math: str = "native") -> PackedConv:
    """Pack a torch-layout weight ([Cout,Cin,KH,KW], [Cout,Cin,K] or [Cout,Cin]) once at load time, on the device,
    through nlc_pack_conv_weights (the layout, permutation and folding rules live behind the C ABI).

    row_perm / row_scale reorder and scale output channels (used to bring the reference's qkv
    channel orders into the canonical [3][H][D] order and to fold the attention scale or an eval-mode
    BatchNorm); bias_add is added to the (permuted, scaled) bias; col_perm reorders input features.
    ``math="f16x3"`` (float32 only): pack for the split-f16 matrix mode (include/nlc_hip.h, NLC_MATH_F16X3).
    """
    lib = _ext.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise _ext.NlcError("pack_conv: weights are packed on the GPU (no CPU fallback)")
    w = weight.detach()
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    Cout, Cin, KH, KW = w.shape
    mcode = MATH_MODES[math]
    if mcode != MATH_NATIVE and dtype != torch.float32:
        raise TypeError("pack_conv: math='f16x3' is a mode of float32 tensors")
    cout_mult, cin_mult = _ext.pack_dims(dtype_enum(dtype))
    Cin_pad, Cout_pad = _round_up(Cin, cin_mult), _round_up(Cout, cout_mult)
    with torch.cuda.device(device):
        wd = w.to(device=device, dtype=torch.float32).contiguous()
        bd = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        rp = None if row_perm is None else row_perm.to(device=device, dtype=torch.int32).contiguous()
        cp = None if col_perm is None else col_perm.to(device=device, dtype=torch.int32).contiguous()
        rs = None if row_scale is None else row_scale.detach().to(device=device, dtype=torch.float64).contiguous()
        ba = None if bias_add is None else bias_add.detach().to(device=device, dtype=torch.float64).contiguous()
        for nm, v, n in (("row_perm", rp, Cout), ("row_scale", rs, Cout), ("bias_add", ba, Cout), ("col_perm", cp, Cin)):
            if v is not None and v.numel() != n:
                raise ValueError(f"pack_conv: {nm} must have {n} entries")
        packed = torch.empty(Cout_pad, KH * KW, Cin_pad, device=device, dtype=dtype)
        has_bias = bd is not None or ba is not None
        bout = torch.empty(Cout, device=device, dtype=torch.float32) if has_bias else None
        wsc = torch.empty(Cout_pad, device=device, dtype=torch.float32) if mcode == MATH_F16X3 else None
        check(lib.nlc_pack_conv_weights_ex(wd.data_ptr(), _ptr(bd), Cout, Cin, KH, KW, _ptr(rp), _ptr(rs), _ptr(ba), _ptr(cp),
                                           dtype_enum(dtype), mcode, packed.data_ptr(), _ptr(bout), _ptr(wsc), _stream()),
              "nlc_pack_conv_weights_ex")
        torch.cuda.current_stream().synchronize()      # load time: the f32 staging copies may be freed after this
    return PackedConv(w=packed, bias=bout, Cin=Cin, Cout=Cout, KH=KH, KW=KW, Cin_pad=Cin_pad, Cout_pad=Cout_pad, dtype=dtype, math=mcode,
                      w_scale=wsc)


# --------------------------------------------------------------------------------------
# kernels
# --------------------------------------------------------------------------------------
# When bench.py sets this to a list, every nlc_conv2d launch appends (start_event, end_event,
# algorithmic FLOPs = 2*M*N*K of the direct convolution, dtype).
CONV_PROFILE = None
# Kernel-selection policy handed to every nlc_conv2d call (nlc_conv_desc.policy): "auto" is the production dispatch;
# tests and A/B tools pin a kernel with "halo" (LDS-halo kernel for every eligible shape), "no_halo" or "generic".
CONV_POLICIES = {"auto": 0, "halo": 1, "no_halo": 2, "generic": 3, "small": 4}
CONV_POLICY = "auto"
CONV_TUNING = 0              # nlc_conv_desc.tuning: schedule A/B switches for tools/ (0 in production)
CONV_DEBUG = 0               # nlc_conv_desc.debug: bit 0 = verify the split-K arrival counters before every split launch; bit 1 = verify
                             # that f16x3 inputs are inside the split's domain |x| < 65504 (both synchronise: tests / triage)
# Ride-along GroupNorm statistics come per 8 output channels - or per 4 when the consumer's groups are 4 / 12 / 20 ... channels wide:
# with the networks' 32 groups that is every <= 128-channel tensor (cfg 4's two highest-resolution levels, EDM's first layer),
# whose GroupNorm otherwise pays a statistics pass over HBM (0.74 ms per NLC step of cfg 4).  A/B switch.
STATS_GRANULE_4 = True


def stats_granule(cout: int, groups_hint: int = 32) -> int:
    gs = cout // groups_hint if cout % groups_hint == 0 else 0
    return 4 if (STATS_GRANULE_4 and cout % 8 == 0 and gs and gs % 8 != 0 and gs % 4 == 0) else 8


def _stats_gran_of(t: torch.Tensor, st: torch.Tensor) -> int:
    return t.shape[-1] // st.shape[1]


# ---- ride-along GroupNorm statistics: zeroed accumulators ---------------------------------------------------------------------
# A convolution ADDS the (sum, sum of squares) totals of its output to an int64 [B, C/g, 4] buffer that must be zero when it starts
# (include/nlc_hip.h, nlc_conv_desc.stats_out).  Inside a network evaluation the buffers of all its convolutions are slices of ONE
# arena that is zeroed by ONE memset when the evaluation begins (HipModule wraps run / run_nhwc in stats_scope); outside of one
# (tests, tools calling conv2d directly) every call gets a zeroed tensor of its own.
class StatsArena:
    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.off = 0
        self.need = 0              # int64 elements the last evaluation asked for
        self.gen = 0               # evaluations begun: a statistics slice is valid only within the evaluation that produced it
        self.overflowed = False
        self.retired = []          # outgrown buffers stay allocated: hipGraphs captured earlier hold their addresses

    def begin(self, device) -> None:
        want = max(self.need, 1 << 16)
        if self.buf is None or self.buf.device != torch.device(device) or self.buf.numel() < self.need:
            if self.buf is not None:
                self.retired.append(self.buf)
            self.buf = torch.zeros(want + want // 4, device=device, dtype=torch.int64)
        else:
            self.buf.zero_()
        self.off, self.need, self.overflowed = 0, 0, False
        self.gen += 1

    def alloc(self, n: int, device) -> torch.Tensor:
        n = (n + 1) & ~1                                    # 16-byte aligned slices
        self.need += n
        if self.buf is not None and self.off + n <= self.buf.numel():
            v = self.buf[self.off:self.off + n]
            self.off += n
            return v
        self.overflowed = True                              # this evaluation: a memset of its own; the arena grows at the next begin()
        return torch.zeros(n, device=device, dtype=torch.int64)


_ARENA: Optional[StatsArena] = None


class stats_scope:
    """``with stats_scope(arena, device):`` - the statistics buffers of every conv2d / conv_first inside come from ``arena``."""

    def __init__(self, arena: StatsArena, device):
        self.arena, self.device = arena, device

    def __enter__(self):
        global _ARENA
        self.prev = _ARENA
        self.arena.begin(self.device)
        _ARENA = self.arena
        return self.arena

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = self.prev
        return False


def _new_stats(out: torch.Tensor, B: int, cout: int, gran: int) -> torch.Tensor:
    """A zeroed int64 [B, cout/gran, 4] totals buffer attached to ``out`` (consumed by groupnorm(); lives and dies with the tensor
    object, and - when it is a slice of an evaluation's arena - is only honoured within that evaluation: ride_stats())."""
    n = B * (cout // gran) * 4
    if _ARENA is not None:
        st = _ARENA.alloc(n, out.device).view(B, cout // gran, 4)
        if not torch.cuda.is_current_stream_capturing():
            # (inside a capture the tensor is a static output of the graph: valid until the next replay that writes the same
            #  arena - the documented contract of HipModule's graph mode - and Python-side generations do not advance on replay)
            out._nlc_stats_gen = (_ARENA, _ARENA.gen)
    else:
        st = torch.zeros(B, cout // gran, 4, device=out.device, dtype=torch.int64)
    out._nlc_stats = st
    return st


def ride_stats(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The totals that rode along with the convolution that produced ``t``, or None (none attached, or their arena has been
    re-zeroed by a later evaluation: the consumer then computes the statistics itself)."""
    if t is None:
        return None
    st = getattr(t, "_nlc_stats", None)
    if st is None:
        return None
    tag = getattr(t, "_nlc_stats_gen", None)
    if tag is not None and tag[0].gen != tag[1]:
        return None
    return st


def conv2d(x0: torch.Tensor, pw: PackedConv, *, x1: Optional[torch.Tensor] = None, stride: int = 1,
           pad: Optional[tuple] = None, out_hw: Optional[tuple] = None, upsample2x: bool = False,
           emb: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, out_scale: float = 1.0,
           act: int = ACT_NONE, out_nchw_f32: bool = False, use_bias: bool = True,
           emit_stats: bool = True, allow_split: bool = False, gn_coef: Optional[torch.Tensor] = None,
           gn_act: int = ACT_NONE, query_prologue: bool = False, res_upsample2x: bool = False,
           norm_out: bool = False, query_norm_out: bool = False, gn_in=None, query_gn_in: bool = False, desc_only: bool = False):
    """Implicit-GEMM conv on [B,H,W,C] (or linear on [M,K] viewed as B=M,H=W=1).
    ``emit_stats``: let the epilogue also write the GroupNorm statistics of the output when the launch supports it
    (bf16 LDS-halo kernel); the following ``groupnorm`` then skips its statistics pass.
    ``allow_split``: an f32 GEMM may split K (bf16 ones always may); the f32 parity path leaves it off so that its
    summation order never depends on the shape.
    ``gn_coef`` / ``gn_act``: the GroupNorm (+FiLM) (+SiLU) in front of this convolution, applied by the convolution itself
    on its way through LDS (coefficients from ``groupnorm_coef``); ``query_prologue=True`` only asks whether this launch
    could do that (returns bool, launches nothing).
    ``res_upsample2x``: ``res`` is [B, Hout/2, Wout/2, Cout] and is added nearest-2x upsampled (the skip branch of an
    up-sampling ResBlock, src/unet_adm.py:186-190) - read in place, no upsampled copy in HBM.
    ``norm_out`` (with ``gn_coef`` / ``gn_act``): a pointwise launch also writes gn_act(GroupNorm(cat(x0, x1))) - the convolution itself
    runs on the input as given - and ``(out, normalised)`` is returned: a ResBlock's skip projection and the GroupNorm + SiLU in front
    of its first 3x3 from one read of the input.  ``query_norm_out=True`` only asks whether this launch could (bool).
    ``gn_in`` (from ``gn_in_spec``): the GroupNorm (+FiLM) (+SiLU) in front of this convolution, applied by the small-map 3x3 kernel on the
    input's way into LDS from the totals that rode along with x0 / x1 - no normalisation launch, no normalised tensor in HBM;
    ``query_gn_in=True`` only asks whether this launch could (bool)."""
    lib = _ext.load()
    dt = pw.dtype
    linear = x0.dim() == 2
    if linear:
        x0 = x0.view(x0.shape[0], 1, 1, x0.shape[1])
    _need(x0, dt, "conv2d x0")
    B, Hin, Win, C0 = x0.shape
    C1 = 0
    if x1 is not None:
        _need(x1, dt, "conv2d x1")
        if tuple(x1.shape[:3]) != (B, Hin, Win):
            raise ValueError("conv2d: x1 spatial shape mismatch")
        C1 = x1.shape[3]
    if C0 + C1 != pw.Cin:
        raise ValueError(f"conv2d: input channels {C0}+{C1} != weight Cin {pw.Cin}")
    if pad is None:
        pad = (pw.KH // 2, pw.KW // 2)
    HL, WL = (2 * Hin, 2 * Win) if upsample2x else (Hin, Win)
    if out_hw is None:
        Hout = (HL + 2 * pad[0] - pw.KH) // stride + 1
        Wout = (WL + 2 * pad[1] - pw.KW) // stride + 1
    else:
        Hout, Wout = out_hw
    if res is not None:
        _need(res, dt, "conv2d res")
        if res_upsample2x and (Hout % 2 or Wout % 2):
            raise ValueError("conv2d: res_upsample2x needs even output dims")
        if res.numel() != B * Hout * Wout * pw.Cout // (4 if res_upsample2x else 1):
            raise ValueError("conv2d: residual shape mismatch")
    elif res_upsample2x:
        raise ValueError("conv2d: res_upsample2x without a residual")
    emb_stride = 0
    if emb is not None:
        # usually a row-strided view into the network's stacked embedding projections (EmbBank)
        if emb.dtype != torch.float32 or not emb.is_cuda or emb.dim() != 2 or emb.stride(1) != 1:
            raise ValueError("conv2d: emb must be a CUDA f32 [B, >=Cout] view with unit inner stride")
        if emb.shape[0] != B or emb.shape[-1] < pw.Cout:
            raise ValueError("conv2d: emb shape mismatch")
        emb_stride = emb.stride(0)
    d = ConvDesc(x0=x0.data_ptr(), x1=_ptr(x1), C0=C0, C1=C1, B=B, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout,
                 Cout=pw.Cout, KH=pw.KH, KW=pw.KW, stride=stride, pad_t=pad[0], pad_l=pad[1],
                 upsample2x=1 if upsample2x else 0, w=pw.w.data_ptr(), Cin_pad=pw.Cin_pad, Cout_pad=pw.Cout_pad,
                 bias=_ptr(pw.bias) if use_bias else None, emb=_ptr(emb), emb_stride=emb_stride, res=_ptr(res),
                 out_scale=out_scale, act=act, out=None,
                 out_mode=OUT_NCHW_F32 if out_nchw_f32 else OUT_NHWC, policy=CONV_POLICIES[CONV_POLICY], tuning=CONV_TUNING,
                 res_upsample2x=1 if res_upsample2x else 0, math=pw.math, debug=CONV_DEBUG, w_scale=_ptr(pw.w_scale))
    if query_prologue:
        return bool(lib.nlc_conv2d_prologue_supported(C.byref(d), dtype_enum(dt)))
    if query_norm_out:
        return bool(lib.nlc_conv2d_norm_out_supported(C.byref(d), dtype_enum(dt)))
    if query_gn_in:
        return bool(lib.nlc_conv2d_gn_in_supported(C.byref(d), dtype_enum(dt)))
    if out_nchw_f32:
        out = torch.empty(B, pw.Cout, Hout, Wout, device=x0.device, dtype=torch.float32)
    else:
        out = torch.empty(B, Hout, Wout, pw.Cout, device=x0.device, dtype=dt)
    d.out = out.data_ptr()
    if gn_in is not None:
        d.gn_in = C.pointer(gn_in[0])              # (gn_in[1:] keeps the tensors it points into alive until the launch is queued)
    hn = None
    if norm_out:
        if gn_coef is None:
            raise ValueError("conv2d: norm_out needs gn_coef (groupnorm_coef)")
        hn = torch.empty(B, Hin, Win, C0 + C1, device=x0.device, dtype=dt)
        d.norm_out = hn.data_ptr()
    if gn_coef is not None:
        if gn_coef.dtype != torch.float32 or not gn_coef.is_cuda or gn_coef.numel() < B * pw.Cin * 2 + 128:
            raise ValueError("conv2d: gn_coef must be a CUDA f32 [B, Cin, 2] table with >= 512 bytes of slack (groupnorm_coef)")
        d.gn_coef, d.gn_act = gn_coef.data_ptr(), gn_act
    if is16(dt) and emit_stats and not out_nchw_f32 and not linear:
        # GroupNorm statistics of the output ride along in the epilogue (totals, added atomically into a zeroed buffer)
        if lib.nlc_conv2d_stats_partials(C.byref(d), dtype_enum(dt)) > 0:
            gran = stats_granule(pw.Cout)
            stats = _new_stats(out, B, pw.Cout, gran)
            d.stats_out, d.stats_bytes, d.stats_granule = stats.data_ptr(), stats.numel() * 8, gran
    if is16(dt) or allow_split:      # split-K scratch for the few-tile / long-K levels (a cheap host query)
        need = lib.nlc_conv2d_workspace_bytes(C.byref(d), dtype_enum(dt))
        if need > 0:
            ws = _conv_workspace(x0.device, need)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if desc_only:                                # (resblock_small: the caller launches; everything the descriptor points into is returned too)
        return d, out, (x0, x1, res, emb, gn_in, pw)
    prof = CONV_PROFILE
    if prof is not None:
        # bench.py's roofline leg: HIP events on the launch stream around this one kernel
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _launch_conv(lib, d, dt)
        e1.record()
        prof.append((e0, e1, 2.0 * B * Hout * Wout * pw.Cout * pw.KH * pw.KW * pw.Cin, dt,
                     (B * Hout * Wout, pw.Cout, pw.KH * pw.KW, pw.Cin, stride, int(upsample2x), C1)))
    else:
        _launch_conv(lib, d, dt)
    if linear and not out_nchw_f32:
        out = out.view(B, pw.Cout)
    if norm_out:
        return out, hn
    return out


def _launch_conv(lib, d: ConvDesc, dt: torch.dtype) -> None:
    try:
        check(lib.nlc_conv2d(C.byref(d), dtype_enum(dt), _stream()), "nlc_conv2d")
    except _ext.NlcError as e:
        if d.workspace and (e.rc == _ext.NLC_ELAUNCH or "poisoned workspace" in str(e)):
            # a split-K launch that did not complete may leave arrival counters non-zero (and the debug check of desc.debug bit 0
            # has just found some), and every later split launch on this workspace would then reduce early or never: drop the
            # workspace, the next call allocates a zeroed one.  (Any other call the library REJECTED - NLC_EINVAL /
            # NLC_EUNSUPPORTED - launched nothing and leaves the workspace as it was.)
            reset_conv_workspaces()
        raise


WS_GENERATION = 0            # bumped by reset_conv_workspaces(): part of config_key(), so captured hipGraphs that reference a dropped
                             # workspace are never replayed (HipModule re-captures under the new key)


def reset_conv_workspaces() -> None:
    """Forget every split-K workspace (they are re-allocated zeroed on demand).  Called after a failed nlc_conv2d launch.  The old
    buffers stay allocated - a hipGraph captured earlier may still hold their addresses - but no such graph is replayed again:
    the generation counter is part of every graph's cache key."""
    global WS_GENERATION
    _ws_retired.extend(_conv_ws.values())
    _conv_ws.clear()
    WS_GENERATION += 1


def gn_in_spec(x0: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], *, groups: int, eps: float, silu: bool,
               x1: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None):
    """(nlc_gn_in, keep-alive...) for conv2d(gn_in=...): GroupNorm (+FiLM) (+SiLU) over cat(x0, x1) described by the totals that rode
    along with the producing convolutions, or None when they are not available (f32 models, inputs without attached statistics,
    group sizes that the totals' chunks cannot express): the caller then runs groupnorm()."""
    if not FUSED_GN_STATS or not is16(x0.dtype):
        return None
    C0 = x0.shape[-1]
    C1 = 0 if x1 is None else x1.shape[-1]
    Ctot = C0 + C1
    if C0 % 8 or C1 % 8 or Ctot % groups:
        return None
    s0, s1 = ride_stats(x0), ride_stats(x1)
    if s0 is None or (x1 is not None and s1 is None):
        return None
    g0 = _stats_gran_of(x0, s0)
    g1 = _stats_gran_of(x1, s1) if s1 is not None else 8
    gs = Ctot // groups
    if gs % g0 or (x1 is not None and gs % g1):
        return None
    ss_stride = 0
    if scale is not None:
        ss_stride = scale.stride(0)
        if shift is None or shift.stride(0) != ss_stride or scale.stride(1) != 1 or shift.stride(1) != 1 or scale.dtype != torch.float32:
            raise ValueError("gn_in_spec: scale/shift must be row-strided f32 views sharing a row stride")
    spec = _ext.GnIn(stats0=s0.data_ptr(), stats1=_ptr(s1), granule0=g0, granule1=g1, groups=groups, eps=eps, gamma=_ptr(gamma),
                     beta=_ptr(beta), scale=_ptr(scale), shift=_ptr(shift), ss_stride=ss_stride, act=ACT_SILU if silu else ACT_NONE)
    return (spec, s0, s1, gamma, beta, scale, shift)


_RB_BARRIER: dict = {}       # device -> two zeroed ints (the one-launch ResBlock's grid barrier; the kernel leaves them zero)


def resblock_small(x: torch.Tensor, c1: PackedConv, c2: PackedConv, g1, b1, g2, b2, *, groups: int, eps: float,
                   scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
                   emb1: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """EXPERIMENT (tools/resblock_bench.py; not used by the networks): conv2(act(GN2(conv1(act(GN1(x))) (+ emb1)) FiLM)) + x in ONE
    launch (nlc_resblock_small), or None when there is no one-launch form for this geometry."""
    lib = _ext.load()
    spec1 = gn_in_spec(x, g1, b1, groups=groups, eps=eps, silu=True)
    if spec1 is None or not conv2d(x, c1, query_gn_in=True):
        return None
    d1, h, keep1 = conv2d(x, c1, gn_in=spec1, emb=emb1, desc_only=True)
    spec2 = gn_in_spec(h, g2, b2, groups=groups, eps=eps, silu=True, scale=scale, shift=shift)
    if spec2 is None or not conv2d(h, c2, query_gn_in=True):
        return None
    d2, out, keep2 = conv2d(h, c2, gn_in=spec2, res=x, desc_only=True)
    bar = _RB_BARRIER.get(x.device)
    if bar is None:
        bar = _RB_BARRIER[x.device] = torch.zeros(2, device=x.device, dtype=torch.int32)
    rc = lib.nlc_resblock_small(C.byref(d1), C.byref(d2), bar.data_ptr(), dtype_enum(x.dtype), _stream())
    if rc == _ext.NLC_EUNSUPPORTED:
        return None
    check(rc, "nlc_resblock_small")
    return out


# networks: on the small maps (8 / 16 / 32 pixels wide) the GroupNorm (+FiLM) + SiLU in front of a ResBlock's 3x3 convolutions is applied
# by the convolution itself on the input's way into LDS (nlc_conv_desc.gn_in, conv_small.hip) instead of by a launch of its own
FUSE_GN_SMALL = True


def config_key() -> tuple:
    """Every module-level switch that changes which kernels / layouts a network evaluation launches.  HipModule keys its captured
    hipGraphs by it (a graph bakes the configuration it was captured under); setters of these switches need no other hook."""
    return (CONV_POLICY, CONV_TUNING, CONV_DEBUG, FUSE_GN_CONV, FUSE_GN_CONV_NT1, FUSE_GN_CONV_MAXC, FUSE_GN_POOL, FUSE_GN_SKIP, FUSE_GN_SMALL, FUSED_GN_STATS, STATS_GRANULE_4, ATTN_BASE2,
            WS_GENERATION)


def conv_first(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype,
               in_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """First-layer conv (Cin<=4) straight from the f32 NCHW sampler state; w: [Cout][KH*KW][Cin] f32."""
    lib = _ext.load()
    _need(x_nchw, torch.float32, "conv_first x")
    B, Cin, H, W = x_nchw.shape
    Cout, taps, cin_w = w.shape
    k = int(round(math.sqrt(taps)))
    if cin_w != Cin or k * k != taps:
        raise ValueError("conv_first: weight shape mismatch")
    out = torch.empty(B, H, W, Cout, device=x_nchw.device, dtype=dtype)
    stats = None
    gran = stats_granule(Cout)
    if lib.nlc_conv_first_stats_partials(Cin, H, W, Cout, k, k, dtype_enum(dtype)) > 0:
        stats = _new_stats(out, B, Cout, gran)
    check(lib.nlc_conv_first(x_nchw.data_ptr(), _ptr(in_scale), w.data_ptr(), _ptr(bias), out.data_ptr(),
                             B, Cin, H, W, Cout, k, k, dtype_enum(dtype), _ptr(stats),
                             0 if stats is None else stats.numel() * 8, gran, _stream()), "nlc_conv_first")
    return out


_gn_ws = {}
_conv_ws = {}
_ws_retired = []             # outgrown workspaces stay allocated (hipGraphs captured earlier keep their addresses); bounded: _grown()
FUSED_GN_STATS = True        # groupnorm() uses statistics emitted by the producing conv2d() when they are attached


def _grown(nbytes: int) -> int:
    """Workspace sizes grow geometrically (next power of two, at least 1 MiB): a sequence of growing requests retires at most
    log2 buffers whose sizes sum to less than the live one - _ws_retired stays bounded."""
    n = 1 << 20
    while n < nbytes:
        n <<= 1
    return n


def _conv_workspace(device, nbytes: int) -> torch.Tensor:
    """Per (device, stream) scratch for nlc_conv2d's split-K partials; grows geometrically to cover the largest request.  Allocated
    ZEROED: its first 4 KiB are the library's arrival counters, which every launch leaves zero (include/nlc_hip.h,
    nlc_conv_desc.workspace)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _conv_ws.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _ws_retired.append(ws)          # a captured hipGraph may still reference the smaller buffer: never hand it back
        ws = torch.zeros(_grown(nbytes) // 4, device=device, dtype=torch.float32)
        _conv_ws[key] = ws
    return ws


def _gn_workspace(device, nbytes: int) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty(_grown(nbytes) // 4, device=device, dtype=torch.float32)
        _gn_ws[key] = ws
    return ws


def groupnorm(x0: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], *, groups: int,
              eps: float, silu: bool, x1: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
              shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm(+FiLM)(+SiLU) over cat(x0,x1) on [B,H,W,C] / [B,T,C]; output has C0+C1 channels."""
    lib = _ext.load()
    dt = x0.dtype
    _need(x0, dt, "groupnorm x0")
    B, C0 = x0.shape[0], x0.shape[-1]
    HW = x0.numel() // (B * C0)
    C1 = 0
    if x1 is not None:
        _need(x1, dt, "groupnorm x1")
        C1 = x1.shape[-1]
    Ctot = C0 + C1
    out = torch.empty(*x0.shape[:-1], Ctot, device=x0.device, dtype=dt)
    ss_stride = 0
    if scale is not None:
        # scale / shift are usually the two halves of one [B, 2C] embedding row: row-strided views
        for nm, v in (("scale", scale), ("shift", shift)):
            if v is None or v.dtype != torch.float32 or not v.is_cuda or v.dim() != 2 or v.stride(1) != 1 \
                    or v.shape[1] < Ctot:
                raise ValueError(f"groupnorm: {nm} must be a CUDA f32 [B, >=C] view with unit inner stride")
        ss_stride = scale.stride(0)
        if shift.stride(0) != ss_stride:
            raise ValueError("groupnorm: scale/shift must share a row stride")
    if FUSED_GN_STATS and is16(dt) and Ctot % groups == 0 and C0 % 8 == 0 and C1 % 8 == 0 and Ctot // 8 <= 256:
        s0 = ride_stats(x0)
        s1 = ride_stats(x1)
        gs = Ctot // groups
        g0 = _stats_gran_of(x0, s0) if s0 is not None else 8
        g1 = _stats_gran_of(x1, s1) if s1 is not None else 8
        if s0 is not None and (x1 is None or s1 is not None) and gs % g0 == 0 and (x1 is None or gs % g1 == 0):
            # the totals came with the producing convolutions: ONE streaming launch (its threads derive their coefficients)
            check(lib.nlc_groupnorm_prestats(x0.data_ptr(), _ptr(x1), C0, C1, B, HW, groups, eps, _ptr(gamma), _ptr(beta),
                                             _ptr(scale), _ptr(shift), ss_stride, 1 if silu else 0, out.data_ptr(),
                                             dtype_enum(dt), s0.data_ptr(), g0, _ptr(s1), g1, _stream()), "nlc_groupnorm_prestats")
            return out
    ws = _gn_workspace(x0.device, lib.nlc_groupnorm_workspace_bytes(B, HW, Ctot, groups))
    check(lib.nlc_groupnorm(x0.data_ptr(), _ptr(x1), C0, C1, B, HW, groups, eps, _ptr(gamma), _ptr(beta),
                            _ptr(scale), _ptr(shift), ss_stride, 1 if silu else 0, out.data_ptr(), ws.data_ptr(),
                            dtype_enum(dt), _stream()), "nlc_groupnorm")
    return out


def groupnorm_pool2x2(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], *, groups: int, eps: float,
                      silu: bool, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None):
    """(avgpool2x2(act(GroupNorm(x))), avgpool2x2(x)) on [B,H,W,C] from one read of x - the two branches of a down-sampling
    ResBlock (src/unet_adm.py:193-195); the full-resolution normalised tensor is never written."""
    lib = _ext.load()
    dt = x.dtype
    _need(x, dt, "groupnorm_pool2x2 x")
    if x.dim() != 4:
        raise ValueError("groupnorm_pool2x2: x must be [B,H,W,C]")
    B, H, W, Cc = x.shape
    out_h = torch.empty(B, H // 2, W // 2, Cc, device=x.device, dtype=dt)
    out_x = torch.empty_like(out_h)
    ss_stride = 0
    if scale is not None:
        for nm, v in (("scale", scale), ("shift", shift)):
            if v is None or v.dtype != torch.float32 or not v.is_cuda or v.dim() != 2 or v.stride(1) != 1 or v.shape[1] < Cc:
                raise ValueError(f"groupnorm_pool2x2: {nm} must be a CUDA f32 [B, >=C] view with unit inner stride")
        ss_stride = scale.stride(0)
        if shift.stride(0) != ss_stride:
            raise ValueError("groupnorm_pool2x2: scale/shift must share a row stride")
    ws = _gn_workspace(x.device, lib.nlc_groupnorm_workspace_bytes(B, H * W, Cc, groups))
    s0, g0 = None, 8
    if FUSED_GN_STATS and is16(dt) and Cc % groups == 0:
        s0 = ride_stats(x)
        if s0 is not None:
            g0 = _stats_gran_of(x, s0)
            if (Cc // groups) % g0:
                s0 = None
    check(lib.nlc_groupnorm_pool2x2(x.data_ptr(), Cc, B, H, W, groups, eps, _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift),
                                    ss_stride, 1 if silu else 0, out_h.data_ptr(), out_x.data_ptr(), ws.data_ptr(), dtype_enum(dt),
                                    _ptr(s0), g0, _stream()), "nlc_groupnorm_pool2x2")
    return out_h, out_x


FUSE_GN_POOL = True        # networks: down-sampling ResBlocks use groupnorm_pool2x2 (A/B switch)


def groupnorm_pool2x2_supported(x: torch.Tensor) -> bool:
    per = 8 if is16(x.dtype) else 4
    return FUSE_GN_POOL and x.dim() == 4 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % per == 0 and x.shape[3] // per <= 256


# networks: apply GroupNorm(+FiLM)+SiLU inside the consuming 3x3 convolution's LDS prologue when the launch supports it.
# OFF by default: measured on MI355X (ADM-256, B=16) the prologue removes the 7 ms / step apply pass but costs the convolutions
# 8.4 ms - every (16x16 patch x 128 cout) tile normalises its own halo, i.e. each input element NT x 1.27 = 2.5 ... 5 times, on
# the same SIMD issue ports the MFMAs need: 5.89 vs 6.16 images/s (profiles/r02_summary.md).  Kept, tested, one switch away.
FUSE_GN_CONV = False
# ... except where the convolution has ONE channel tile (Cout <= 128: the two highest-resolution levels of the CelebA-HQ UNet): every
# input element is then normalised 1.27 times (the halo overlap) instead of 2.5-5 times, and the prologue beats the separate pass
# (measured: profiles/r05_summary.md)
FUSE_GN_CONV_NT1 = True
FUSE_GN_CONV_MAXC = 128      # ... i.e. Cout <= this (A/B: 256 = two channel tiles)
FUSE_GN_SKIP = True          # Norm.with_skip: a ResBlock's skip projection also writes act(GroupNorm(x)) of its input (nlc_conv_desc.norm_out)


def groupnorm_coef(x0: torch.Tensor, gamma, beta, *, groups: int, eps: float, x1: Optional[torch.Tensor] = None,
                   scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Per-(image, channel) coefficients (a, b) of GroupNorm(+FiLM) over cat(x0, x1) from the statistics that rode along
    with the producing convolutions - for conv2d(gn_coef=...).  None when they are not available (f32 models, inputs without
    attached statistics, group sizes that 8-channel chunks cannot express): the caller then runs groupnorm()."""
    lib = _ext.load()
    if not FUSED_GN_STATS or not is16(x0.dtype):             # (totals ride along with 16-bit tensors only)
        return None
    B, C0 = x0.shape[0], x0.shape[-1]
    C1 = 0 if x1 is None else x1.shape[-1]
    Ctot = C0 + C1
    if C0 % 8 or C1 % 8 or Ctot % groups:
        return None
    s0 = ride_stats(x0)
    s1 = ride_stats(x1)
    if s0 is None or (x1 is not None and s1 is None):
        return None
    g0 = _stats_gran_of(x0, s0)
    g1 = _stats_gran_of(x1, s1) if s1 is not None else 8
    if (Ctot // groups) % g0 or (x1 is not None and (Ctot // groups) % g1):
        return None
    HW = x0.numel() // (B * C0)
    ss_stride = 0
    if scale is not None:
        ss_stride = scale.stride(0)
        if shift is None or shift.stride(0) != ss_stride or scale.stride(1) != 1 or shift.stride(1) != 1:
            raise ValueError("groupnorm_coef: scale/shift must be row-strided f32 views sharing a row stride")
    coef = torch.empty(B * Ctot * 2 + 128, device=x0.device, dtype=torch.float32)        # + 512 bytes: the consumer's DMA reads whole 1-KiB pieces
    check(lib.nlc_groupnorm_coef(C0, C1, B, HW, groups, eps, _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), ss_stride,
                                 s0.data_ptr(), g0, _ptr(s1), g1, coef.data_ptr(), _stream()), "nlc_groupnorm_coef")
    return coef


LOG2E = 1.4426950408889634


def attention_logit_scale(spec) -> tuple:
    """(extra factor for the q rows of a qkv projection's row_scale, the matching ``base2`` flag of ``attention``): 16-bit models fold
    log2(e) into q at pack time so that the kernel exponentiates with 2^x and no multiply; f32 models keep natural logits and e^x
    (the reference's op order)."""
    return (LOG2E, True) if (isinstance(spec, torch.dtype) and is16(spec) and ATTN_BASE2) else (1.0, False)


ATTN_BASE2 = True            # A/B switch of attention_logit_scale (takes effect when a model is (re)packed)


ATTN_PROFILE = None          # bench.py's roofline leg: a list that receives (start_event, end_event, QK^T + PV FLOPs) per nlc_attention launch


def attention(qkv: torch.Tensor, heads: int, base2: bool = False) -> torch.Tensor:
    """qkv: [B,T,3*H*D] laid out [3][H][D]; returns [B,T,H*D].  ``base2``: the logits are in log2 units (attention_logit_scale)."""
    lib = _ext.load()
    _need(qkv, qkv.dtype, "attention qkv")
    B, T, C3 = qkv.shape
    D = C3 // (3 * heads)
    out = torch.empty(B, T, heads * D, device=qkv.device, dtype=qkv.dtype)
    prof = ATTN_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.nlc_attention(qkv.data_ptr(), out.data_ptr(), B, T, heads, D, dtype_enum(qkv.dtype), 1 if base2 else 0, _stream()),
          "nlc_attention")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 4.0 * B * heads * T * T * D, (T, heads, D)))
    return out


def avgpool2x2(x: torch.Tensor) -> torch.Tensor:
    lib = _ext.load()
    B, H, W, Cc = x.shape
    out = torch.empty(B, H // 2, W // 2, Cc, device=x.device, dtype=x.dtype)
    check(lib.nlc_avgpool2x2(_need(x, x.dtype, "avgpool x").data_ptr(), out.data_ptr(), B, H, W, Cc,
                             dtype_enum(x.dtype), _stream()), "nlc_avgpool2x2")
    return out


def upsample2x(x: torch.Tensor) -> torch.Tensor:
    lib = _ext.load()
    B, H, W, Cc = x.shape
    out = torch.empty(B, 2 * H, 2 * W, Cc, device=x.device, dtype=x.dtype)
    check(lib.nlc_upsample2x(_need(x, x.dtype, "upsample x").data_ptr(), out.data_ptr(), B, H, W, Cc,
                             dtype_enum(x.dtype), _stream()), "nlc_upsample2x")
    return out


def pad_rb(x: torch.Tensor) -> torch.Tensor:
    lib = _ext.load()
    B, H, W, Cc = x.shape
    out = torch.empty(B, H + 1, W + 1, Cc, device=x.device, dtype=x.dtype)
    check(lib.nlc_pad_rb(_need(x, x.dtype, "pad x").data_ptr(), out.data_ptr(), B, H, W, Cc,
                         dtype_enum(x.dtype), _stream()), "nlc_pad_rb")
    return out


def nhwc_to_nchw_f32(x: torch.Tensor) -> torch.Tensor:
    lib = _ext.load()
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, H, W, device=x.device, dtype=torch.float32)
    check(lib.nlc_nhwc_to_nchw_f32(_need(x, x.dtype, "nhwc_to_nchw x").data_ptr(), out.data_ptr(), B, H, W, Cc,
                                   dtype_enum(x.dtype), _stream()), "nlc_nhwc_to_nchw_f32")
    return out


def nchw_f32_to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    lib = _ext.load()
    _need(x, torch.float32, "nchw_to_nhwc x")
    B, Cc, H, W = x.shape
    out = torch.empty(B, H, W, Cc, device=x.device, dtype=dtype)
    check(lib.nlc_nchw_f32_to_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, Cc, dtype_enum(dtype), _stream()),
          "nlc_nchw_f32_to_nhwc")
    return out


def timestep_embedding(t: torch.Tensor, freqs: torch.Tensor, sin_first: bool) -> torch.Tensor:
    lib = _ext.load()
    _need(t, torch.float32, "timestep_embedding t")
    _need(freqs, torch.float32, "timestep_embedding freqs")
    B, half = t.shape[0], freqs.shape[0]
    out = torch.empty(B, 2 * half, device=t.device, dtype=torch.float32)
    check(lib.nlc_timestep_embedding(t.data_ptr(), freqs.data_ptr(), out.data_ptr(), B, 2 * half,
                                     1 if sin_first else 0, _stream()), "nlc_timestep_embedding")
    return out


# ---- sampler state ---------------------------------------------------------------------
def row_sumsq(x: torch.Tensor, d_used: Optional[int] = None) -> torch.Tensor:
    lib = _ext.load()
    _need(x, torch.float32, "row_sumsq x")
    B = x.shape[0]
    stride = x.numel() // B
    D = stride if d_used is None else d_used
    out = torch.empty(B, device=x.device, dtype=torch.float32)
    check(lib.nlc_row_sumsq(x.data_ptr(), out.data_ptr(), B, stride, D, _stream()), "nlc_row_sumsq")
    return out


def refine_sigma(sumsq, sqrt_dim, norm_max, norm_min, sigma_sched, sigma_prev_sched, refine, sigmas, t_sched,
                 time_shift, sigma_t, sigma_prev, t, c_in, sigma_in=None, t_in=None, prev_is_ratio=False, t_slopes=None):
    """nlc_refine_sigma_ex.  ``sigma_in``/``t_in``: per-sample scheduled values (may alias the outputs);
    ``t_slopes``: continuous-t interpolation slopes (None = discrete searchsorted)."""
    lib = _ext.load()
    B = sigma_t.shape[0]
    for v, n in ((sigma_in, "sigma_in"), (t_in, "t_in")):
        if v is not None and (v.dtype != torch.float32 or v.numel() != B or not v.is_contiguous()):
            raise _ext.NlcError(f"refine_sigma: {n} must be a contiguous float32 [B] tensor")
    if t_slopes is not None and (sigmas is None or t_slopes.numel() != sigmas.numel() - 1):
        raise _ext.NlcError("refine_sigma: t_slopes needs the sigma table and n_sigmas-1 entries")
    d = _ext.SigmaDesc(sumsq=_ptr(sumsq), sigma_in=_ptr(sigma_in), t_in=_ptr(t_in), sigmas=_ptr(sigmas), t_slopes=_ptr(t_slopes),
                       sigma_t=sigma_t.data_ptr(), sigma_prev=sigma_prev.data_ptr(), t=t.data_ptr(), c_in=c_in.data_ptr(),
                       sqrt_dim=sqrt_dim, norm_max=norm_max, norm_min=norm_min, sigma_sched=float(sigma_sched),
                       sigma_prev_sched=float(sigma_prev_sched), t_sched=float(t_sched), time_shift=float(time_shift),
                       refine=1 if refine else 0, prev_is_ratio=1 if prev_is_ratio else 0,
                       n_sigmas=0 if sigmas is None else sigmas.shape[0], B=B)
    check(lib.nlc_refine_sigma_ex(C.byref(d), _stream()), "nlc_refine_sigma_ex")


def sigma_correct(r, partial, sigmas, sigma_t, sigma_prev, t, c_in, t_slopes=None):
    lib = _ext.load()
    _need(r, torch.float32, "sigma_correct r")
    B = sigma_t.shape[0]
    check(lib.nlc_sigma_correct(r.data_ptr(), 1 if partial else 0, sigmas.data_ptr(), _ptr(t_slopes), sigmas.shape[0],
                                sigma_t.data_ptr(), sigma_prev.data_ptr(), t.data_ptr(), c_in.data_ptr(), B,
                                _stream()), "nlc_sigma_correct")


def proj_sigma(sumsq, sqrt_dim, norm_max, norm_max_sq, costheta, term0, r1, r2, r3, sigmas, t_slopes, last_norm, sigma_t,
               sigma_prev, t):
    """projection_loop's sigma re-estimation, in place on (last_norm, sigma_t, t) (image_sample.py:485-497)."""
    lib = _ext.load()
    B = sigma_t.shape[0]
    check(lib.nlc_proj_sigma(sumsq.data_ptr(), sqrt_dim, norm_max, norm_max_sq, costheta, term0, r1, r2, r3, sigmas.data_ptr(),
                             _ptr(t_slopes), sigmas.shape[0], last_norm.data_ptr(), sigma_t.data_ptr(), sigma_prev.data_ptr(),
                             t.data_ptr(), B, _stream()), "nlc_proj_sigma")


_QUANTILE_WS: dict = {}      # (device, B) -> zeroed int32 workspace of nlc_dynamic_threshold_ws (every call leaves it zeroed)


def dynamic_threshold(x0_hat: torch.Tensor, q: float, max_value: float, single_workgroup: bool = False) -> torch.Tensor:
    """Per-sample quantile of |x0_hat| (torch.quantile, exact), clamped to [1, max_value] (src/experiments.py: dynamic thresholding).
    ``single_workgroup``: the one-launch form with one workgroup per sample (tests compare the two)."""
    lib = _ext.load()
    _need(x0_hat, torch.float32, "dynamic_threshold x")
    B = x0_hat.shape[0]
    D = x0_hat.numel() // B
    out = torch.empty(B, device=x0_hat.device, dtype=torch.float32)
    if single_workgroup:
        check(lib.nlc_dynamic_threshold(x0_hat.data_ptr(), q, max_value, out.data_ptr(), B, D, _stream()),
              "nlc_dynamic_threshold")
        return out
    key = (x0_hat.device, B)
    ws = _QUANTILE_WS.get(key)
    if ws is None:
        ws = _QUANTILE_WS[key] = torch.zeros(lib.nlc_dynamic_threshold_ws_bytes(B) // 4, device=x0_hat.device, dtype=torch.int32)
    check(lib.nlc_dynamic_threshold_ws(x0_hat.data_ptr(), q, max_value, out.data_ptr(), B, D, ws.data_ptr(), ws.numel() * 4, _stream()),
          "nlc_dynamic_threshold_ws")
    return out


def sched_x0(desc: SchedDesc):
    check(_ext.load().nlc_sched_x0(C.byref(desc), _stream()), "nlc_sched_x0")


def sched_step(desc: SchedDesc, nan_flag: Optional[torch.Tensor] = None):
    check(_ext.load().nlc_sched_step(C.byref(desc), _ptr(nan_flag), _stream()), "nlc_sched_step")


def scale_rows(x: torch.Tensor, scale: Optional[torch.Tensor], scalar: float = 1.0) -> torch.Tensor:
    lib = _ext.load()
    _need(x, torch.float32, "scale_rows x")
    B = x.shape[0]
    out = torch.empty_like(x)
    check(lib.nlc_scale_rows(x.data_ptr(), _ptr(scale), scalar, out.data_ptr(), B, x.numel() // B, _stream()),
          "nlc_scale_rows")
    return out


def lincomb_rows(x: torch.Tensor, ca: torch.Tensor, y: Optional[torch.Tensor] = None, cb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = ca[b]*x[b] (+ cb[b]*y[b]) on f32 [B, ...] tensors; ca / cb: f32 [B] device tensors."""
    lib = _ext.load()
    _need(x, torch.float32, "lincomb_rows x")
    B = x.shape[0]
    out = torch.empty_like(x)
    if ca.numel() != B or (cb is not None and cb.numel() != B):
        raise ValueError(f"lincomb_rows: coefficient vectors must have one entry per row ({B})")
    if (y is None) != (cb is None):
        raise ValueError("lincomb_rows: y and cb come together")
    if y is not None:
        _need(y, torch.float32, "lincomb_rows y")
        if y.shape != x.shape:
            raise ValueError(f"lincomb_rows: y has shape {tuple(y.shape)}, x {tuple(x.shape)}")
    check(lib.nlc_lincomb_rows(x.data_ptr(), _need(ca, torch.float32, "lincomb_rows ca").data_ptr(), _ptr(y),
                               None if cb is None else _need(cb, torch.float32, "lincomb_rows cb").data_ptr(), out.data_ptr(), B,
                               x.numel() // B, _stream()), "nlc_lincomb_rows")
    return out
