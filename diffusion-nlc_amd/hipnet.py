"""Common machinery of the HIP-backed networks: reference-format parameters on the host,
packed weights on the device, and the building blocks shared by the three UNet families.

A ``HipModule`` looks like the ``nn.Module`` it replaces to the reference's entry points
(``state_dict`` / ``load_state_dict`` with the reference's key names, ``parameters``, ``to``,
``eval``, ``convert_to_fp16``) but owns no autograd state: parameters are plain f32 host tensors
that are packed once per (device, compute dtype) into the layouts the kernels want.

Compute dtype: ``torch.float32`` (the parity path), ``torch.bfloat16`` or ``torch.float16`` (16-bit operands /
f32 accumulate, the throughput paths; the two run at the same MFMA rate on gfx950).  The reference's
``convert_to_fp16`` is ``torch.float16`` here as it is upstream, and like upstream it touches convolution /
attention operands only - GroupNorm statistics, softmax, timestep-embedding MLPs and the sigma head stay f32
(src/fp16_util.py:15-22, src/unet_adm.py:620-626,1060-1065).
A float32 model additionally has a matrix mode (``set_matmul``): ``"native"`` = exact f32 MFMA (1/16 of the 16-bit
rate), ``"f16x3"`` = every convolution operand split into two f16 halves, three 16-bit MFMA passes, f32 accumulate
(include/nlc_hip.h NLC_MATH_F16X3: ~22 significand bits per operand, 3/16 of the 16-bit rate); storage and every
non-convolution kernel stay exact f32 either way.
"""
from __future__ import annotations

import functools
import math
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import ops
from ._ext import ACT_GELU, ACT_NONE, ACT_SILU, NlcError


class SpecBuilder:
    """Collects (key -> shape) in module-construction order, mirroring nn.Module.state_dict()."""

    def __init__(self):
        self.spec: "OrderedDict[str, Tuple[Tuple[int, ...], torch.dtype]]" = OrderedDict()

    def add(self, key, shape, dtype=torch.float32):
        self.spec[key] = (tuple(int(s) for s in shape), dtype)

    def conv(self, p, cout, cin, k, bias=True, dims=2):
        self.add(p + ".weight", (cout, cin) + (k,) * dims)
        if bias:
            self.add(p + ".bias", (cout,))

    def linear(self, p, cout, cin, bias=True):
        self.add(p + ".weight", (cout, cin))
        if bias:
            self.add(p + ".bias", (cout,))

    def norm(self, p, c):
        self.add(p + ".weight", (c,))
        self.add(p + ".bias", (c,))

    def batchnorm(self, p, c):
        self.norm(p, c)
        self.add(p + ".running_mean", (c,))
        self.add(p + ".running_var", (c,))
        self.add(p + ".num_batches_tracked", (), torch.int64)


def _graphable(fn):
    """Network evaluations (``run`` / ``run_nhwc``) replay from a captured hipGraph when ``module.use_graphs`` is set.

    One evaluation is a fixed sequence of ~50-600 kernel launches whose only inputs are device tensors (state, per-sample
    t and input scale): it is captured once per (entry point, shapes, dtypes, options) on a side stream into a private
    memory pool and then replayed with ONE hipGraphLaunch.  Inputs are copied into the capture's static buffers unless
    the caller already passes those; the returned tensors are the capture's static outputs - valid until the next
    replay of the same graph (the sampling loops consume them before they evaluate the network again): a caller that keeps
    an ``eps`` / ``feat`` tensor across another evaluation of the same entry point must ``clone()`` it.  The ride-along
    GroupNorm statistics of a returned tensor are copied out of the module's arena into storage of the graph's own pool
    (``_own_output_stats``), so they live exactly as long as the tensor's data: another graph, an eager call or another entry
    point of the same module re-zeroes the arena, not them."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if not self.use_graphs or torch.cuda.is_current_stream_capturing():
            return self._eval(fn, args, kwargs)
        return self._graph_call(fn, args, kwargs)
    return wrapper


def _own_output_stats(out) -> None:
    """Inside a capture: give every returned tensor a private copy of its ride-along statistics (slices of the module's ONE arena
    otherwise - rewritten by the next evaluation of any entry point of the module, while the tensor's data stays valid in the
    graph's pool; eager calls are protected by the generation tag of ops.ride_stats instead)."""
    if torch.is_tensor(out):
        st = getattr(out, "_nlc_stats", None)
        if st is not None:
            out._nlc_stats = st.clone()
    elif isinstance(out, (tuple, list)):
        for o in out:
            _own_output_stats(o)
    elif isinstance(out, dict):
        for o in out.values():
            _own_output_stats(o)


class HipModule:
    """Parameter container + lazy device plan.  Subclasses implement ``param_spec`` and ``_build``."""

    use_graphs = False

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for name in ("run", "run_nhwc"):
            if name in cls.__dict__:
                setattr(cls, name, _graphable(cls.__dict__[name]))

    def _eval(self, fn, args, kwargs):
        """One network evaluation: the ride-along GroupNorm statistics of all its convolutions are slices of this module's arena,
        zeroed by one memset up front (ops.StatsArena)."""
        if self.device.type != "cuda":
            return fn(self, *args, **kwargs)
        arena = self.__dict__.get("_stats_arena")
        if arena is None:
            arena = self.__dict__["_stats_arena"] = ops.StatsArena()
        with ops.stats_scope(arena, self.device):
            out = fn(self, *args, **kwargs)
            if torch.cuda.is_current_stream_capturing():
                _own_output_stats(out)
            return out

    def _graph_call(self, fn, args, kwargs):
        self._require_gpu()
        items = list(args) + [kwargs[k] for k in sorted(kwargs)]
        # (an input that carries ride-along GroupNorm statistics - ops.conv2d attaches them to its output - keeps them through the
        #  static copy: the captured network must consume the same statistics an eager call would, not recompute them in another
        #  summation order)
        stats_of = lambda a: ops.ride_stats(a) if torch.is_tensor(a) else None
        key = (fn.__name__, self.compute_dtype, self.matmul, ops.config_key(), len(args), tuple(sorted(kwargs)),
               tuple((tuple(a.shape), a.dtype, None if stats_of(a) is None else tuple(stats_of(a).shape)) if torch.is_tensor(a) else a
                     for a in items))
        cache = self.__dict__.setdefault("_graphs", {})
        ent = cache.get(key)
        with torch.cuda.device(self.device):
            if ent is None:
                self.plan()
                # Warm-up AND capture run on one side stream owned by this module: every kernel's one-time launch setup happens in the
                # eager warm-up, and so do the allocations of the per-stream conv / GroupNorm workspaces (ops._conv_ws / _gn_ws are
                # keyed by stream) - they come from the ordinary pool, zeroed counters included, and are merely REFERENCED by the
                # capture instead of being born inside a graph's private pool.
                cs = self.__dict__.get("_capture_stream")
                if cs is None:
                    cs = self.__dict__["_capture_stream"] = torch.cuda.Stream(device=self.device)
                static = [a.clone() if torch.is_tensor(a) else a for a in items]
                for a, st in zip(items, static):
                    if stats_of(a) is not None:
                        st._nlc_stats = stats_of(a).clone()
                s_args = static[:len(args)]
                s_kwargs = dict(zip(sorted(kwargs), static[len(args):]))
                cs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cs):
                    self._eval(fn, s_args, s_kwargs)
                    if self.__dict__["_stats_arena"].overflowed:       # first evaluation ever: the arena is sized now, so that the
                        self._eval(fn, s_args, s_kwargs)              # capture below only zeroes it (no allocation inside the graph)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cs):
                    out = self._eval(fn, s_args, s_kwargs)
                ent = cache[key] = (g, static, out)
            g, static, out = ent
            for a, st in zip(items, static):
                if torch.is_tensor(a) and a.data_ptr() != st.data_ptr():
                    st.copy_(a)
                    if stats_of(a) is not None:
                        st._nlc_stats.copy_(stats_of(a))
            g.replay()
        return out

    def drop_graphs(self):
        self.__dict__.pop("_graphs", None)

    def __init__(self):
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, (shape, dt) in self.param_spec().items():
            self._sd[k] = torch.zeros(shape, dtype=dt)
        self.device = torch.device("cpu")
        self.compute_dtype = torch.float32
        self.matmul = "native"
        self._plan = None
        self.training = False

    # ---- nn.Module look-alike surface ---------------------------------------------------
    def param_spec(self):
        raise NotImplementedError

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, sd, strict: bool = True):
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing keys {missing[:8]}"
                               f"{'...' if len(missing) > 8 else ''}, unexpected keys {unexpected[:8]}")
        for k, v in sd.items():
            if k not in self._sd:
                continue
            if tuple(v.shape) != tuple(self._sd[k].shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(self._sd[k].shape)}")
            self._sd[k] = v.detach().to("cpu", self._sd[k].dtype).clone()
        self._plan = None
        self.drop_graphs()
        return self

    def parameters(self):
        return (v for k, v in self._sd.items() if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var", "resample_filter")))

    def buffers(self):
        return (v for k, v in self._sd.items() if not v.dtype.is_floating_point or k.endswith(("running_mean", "running_var", "resample_filter")))

    def named_parameters(self):
        return ((k, v) for k, v in self._sd.items() if v.dtype.is_floating_point)

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NlcError("the HIP path is inference-only (sigma-net training is out of scope, SURVEY.md §8)")
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def to(self, device=None, dtype=None):
        if device is not None:
            device = torch.device(device)
            if device.type == "cuda" and device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            if device != self.device:
                self.device = device
                self._plan = None
                self.drop_graphs()
        if dtype is not None:
            self.set_compute_dtype(dtype)
        return self

    def cuda(self, index=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if index is None else index))

    def set_compute_dtype(self, dtype: torch.dtype):
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise TypeError("compute dtype must be float32, bfloat16 or float16")
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._plan = None                  # captured graphs are keyed by the compute dtype and hold their own weights' plan
            self.drop_graphs()
        return self

    def set_matmul(self, mode: str):
        """Matrix arithmetic of a float32 model's convolutions: "native" (exact f32 MFMA) or "f16x3" (split-f16, three passes)."""
        if mode not in ops.MATH_MODES:
            raise ValueError(f"matmul mode must be one of {sorted(ops.MATH_MODES)}")
        if mode != self.matmul:
            self.matmul = mode
            self._plan = None
            self.drop_graphs()
        return self

    def convert_to_fp16(self):
        """The reference's half-precision switch (src/fp16_util.py:15-22, src/unet_adm.py:620-626): IEEE half operands."""
        return self.set_compute_dtype(torch.float16)

    def half(self):
        return self.set_compute_dtype(torch.float16)

    def convert_to_fp32(self):
        return self.set_compute_dtype(torch.float32)

    def bfloat16(self):
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        return self.set_compute_dtype(torch.float32)

    # ---- plan ---------------------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise NlcError(f"{type(self).__name__} runs on the HIP kernels only: move it to a GPU with .to('cuda:0') "
                           "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")

    def plan(self):
        # the packed weights bake ops.ATTN_BASE2 in (log2(e) folded into the q rows): a flip re-packs
        if self._plan is not None and self.__dict__.get("_plan_base2") != ops.ATTN_BASE2:
            self._plan = None
            self.drop_graphs()                 # graphs captured under the old value hold raw pointers into the old plan's weights
        if self._plan is None:
            self._require_gpu()
            with torch.cuda.device(self.device):
                spec = F32X3 if (self.compute_dtype == torch.float32 and self.matmul == "f16x3") else self.compute_dtype
                self._plan = self._build(self._sd, self.device, spec)
                self.__dict__["_plan_base2"] = ops.ATTN_BASE2
        return self._plan

    def _build(self, sd, device, dtype):
        raise NotImplementedError

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


# ------------------------------------------------------------------------------------------
# building blocks on packed weights
# ------------------------------------------------------------------------------------------
def f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Norm:
    """GroupNorm affine parameters on the device."""

    def __init__(self, sd, p, device, groups: int, eps: float):
        self.gamma, self.beta = f32(sd[p + ".weight"], device), f32(sd[p + ".bias"], device)
        self.groups, self.eps = groups, eps

    def __call__(self, x, silu: bool, x1=None, scale=None, shift=None):
        return ops.groupnorm(x, self.gamma, self.beta, groups=self.groups, eps=self.eps, silu=silu, x1=x1,
                             scale=scale, shift=shift)

    def pooled(self, x, silu: bool):
        """(avgpool2x2(act(norm(x))), avgpool2x2(x)): both branches of a down-sampling ResBlock from one read of x."""
        if ops.groupnorm_pool2x2_supported(x):
            return ops.groupnorm_pool2x2(x, self.gamma, self.beta, groups=self.groups, eps=self.eps, silu=silu)
        return ops.avgpool2x2(self(x, silu=silu)), ops.avgpool2x2(x)

    def with_skip(self, x, pw_skip, *, silu: bool, x1=None):
        """(act(norm(cat(x, x1))), skip(cat(x, x1))) from ONE read of the input - the 1x1 skip projection of a ResBlock writes the
        normalised activation for the block's first 3x3 as a side output (nlc_conv_desc.norm_out) - or None when this launch cannot
        (the caller then runs the two passes)."""
        if not ops.FUSE_GN_SKIP or not ops.conv2d(x, pw_skip, x1=x1, query_norm_out=True):
            return None
        coef = ops.groupnorm_coef(x, self.gamma, self.beta, groups=self.groups, eps=self.eps, x1=x1)
        if coef is None:
            return None
        res, hn = ops.conv2d(x, pw_skip, x1=x1, gn_coef=coef, gn_act=ACT_SILU if silu else ACT_NONE, norm_out=True, emit_stats=False)
        return hn, res

    def then_conv(self, x, pw, *, silu: bool, x1=None, scale=None, shift=None, **conv_kw):
        """conv(act(norm(cat(x, x1)))) - with the normalisation applied inside the convolution's LDS prologue when the launch
        supports it and the statistics rode along with x (x1), else as the separate GroupNorm pass followed by the conv."""
        geom = {k: v for k, v in conv_kw.items() if k in ("stride", "pad", "out_hw", "upsample2x", "out_nchw_f32", "res_upsample2x")}
        if ops.FUSE_GN_SMALL and not geom.get("res_upsample2x") and not geom.get("upsample2x") and ops.conv2d(x, pw, x1=x1, query_gn_in=True, **geom):
            spec = ops.gn_in_spec(x, self.gamma, self.beta, groups=self.groups, eps=self.eps, silu=silu, x1=x1, scale=scale, shift=shift)
            if spec is not None:
                return ops.conv2d(x, pw, x1=x1, gn_in=spec, **conv_kw)
        if (ops.FUSE_GN_CONV or (ops.FUSE_GN_CONV_NT1 and pw.Cout <= ops.FUSE_GN_CONV_MAXC)) and ops.conv2d(x, pw, x1=x1, query_prologue=True, **{k: v for k, v in conv_kw.items() if k in ("stride", "pad", "out_hw", "upsample2x", "out_nchw_f32")}):
            coef = ops.groupnorm_coef(x, self.gamma, self.beta, groups=self.groups, eps=self.eps, x1=x1, scale=scale, shift=shift)
            if coef is not None:
                return ops.conv2d(x, pw, x1=x1, gn_coef=coef, gn_act=ACT_SILU if silu else ACT_NONE, **conv_kw)
        return ops.conv2d(self(x, silu=silu, x1=x1, scale=scale, shift=shift), pw, **conv_kw)


class _F32X3:
    """Compute spec handed to ``_build`` in place of a torch dtype: float32 storage, split-f16 matrix math in the convolutions
    (weights packed as (hi, lo) halves).  Literal ``torch.float32`` in a ``_build`` (embedding MLPs, sigma head) stays exact."""

    def __repr__(self):
        return "float32/f16x3"


F32X3 = _F32X3()


def storage_dtype(spec) -> torch.dtype:
    return torch.float32 if spec is F32X3 else spec


def pack_w(weight, bias, spec, device, **kw) -> ops.PackedConv:
    return ops.pack_conv(weight, bias, storage_dtype(spec), device, math="f16x3" if spec is F32X3 else "native", **kw)


def pack(sd, p, spec, device, **kw) -> ops.PackedConv:
    return pack_w(sd[p + ".weight"], sd.get(p + ".bias"), spec, device, **kw)


class EmbBank:
    """All per-block embedding projections of a network as ONE f32 GEMM per forward.

    Every ResBlock's ``Linear(emb)`` reads the same [B, E] vector, so their weights are stacked
    row-wise at load time and evaluated once; blocks then take row-strided views of the result
    (the 'timestep broadcast' of the reference: src/unet_adm.py:245-247, src/unet_simple.py:121,
    src/edm_networks.py:187).
    """

    def __init__(self):
        self._w: List[torch.Tensor] = []
        self._b: List[torch.Tensor] = []
        self._off = 0
        self.packed: Optional[ops.PackedConv] = None

    def add(self, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> Tuple[int, int]:
        n = weight.shape[0]
        self._w.append(weight.detach().float().cpu())
        self._b.append(torch.zeros(n) if bias is None else bias.detach().float().cpu())
        off = self._off
        self._off += n
        return off, n

    def finalize(self, device, allow_split: bool = False):
        """``allow_split``: the (f32) GEMM may split K - set by bf16 models; f32 models keep one summation order."""
        self.allow_split = allow_split
        if self._w:
            self.packed = ops.pack_conv(torch.cat(self._w, 0), torch.cat(self._b, 0), torch.float32, device)
        self._w, self._b = [], []

    def __call__(self, emb_in: torch.Tensor) -> Optional[torch.Tensor]:
        if self.packed is None:
            return None
        return ops.conv2d(emb_in, self.packed, allow_split=getattr(self, "allow_split", False))


class SigmaHead:
    """Flatten -> Linear -> BatchNorm1d(eval) -> act -> Linear, all f32 (src/unet_adm.py:1051-1058,1078-1083).

    BatchNorm in eval mode is an affine map, folded into the first Linear at load time; the
    activation runs in that GEMM's epilogue.  The feature map arrives NHWC in the compute dtype and is
    flattened in the reference's NCHW order by the layout kernel.
    """

    def __init__(self, sd, device, act: int, allow_split: bool = False):
        self.allow_split = allow_split
        g, beta = sd["fc_layer.2.weight"].double(), sd["fc_layer.2.bias"].double()
        mean, var = sd["fc_layer.2.running_mean"].double(), sd["fc_layer.2.running_var"].double()
        s = g / torch.sqrt(var + 1e-5)                  # f64: the fold itself happens inside nlc_pack_conv_weights
        self.fc = ops.pack_conv(sd["fc_layer.1.weight"], sd["fc_layer.1.bias"], torch.float32, device, row_scale=s,
                                bias_add=beta - mean * s)
        self.final = ops.pack_conv(sd["final_mlp.weight"], sd["final_mlp.bias"], torch.float32, device)
        self.act = act

    def __call__(self, h_nhwc: torch.Tensor) -> torch.Tensor:
        flat = ops.nhwc_to_nchw_f32(h_nhwc).view(h_nhwc.shape[0], -1)
        h = ops.conv2d(flat, self.fc, act=self.act, allow_split=self.allow_split)
        return ops.conv2d(h, self.final).view(-1)          # r[b]


def first_conv_weight(sd, p, device):
    """[Cout,Cin,KH,KW] -> [Cout][KH*KW][Cin] f32 for nlc_conv_first."""
    w = sd[p + ".weight"].float()
    co, ci, kh, kw = w.shape
    return (w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci).contiguous().to(device),
            None if (p + ".bias") not in sd else f32(sd[p + ".bias"], device))


def as_f32_cuda(x: torch.Tensor, device) -> torch.Tensor:
    return x.detach().to(device=device, dtype=torch.float32).contiguous()
