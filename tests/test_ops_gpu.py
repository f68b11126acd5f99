"""Op-level parity of the HIP kernels (through the C ABI) against plain PyTorch-CPU f32 ops.

Tolerances: f32 path 2e-4 relative-to-scale (different summation order only); the 16-bit paths are
compared with the same op evaluated in f32 on operands rounded to that type: bf16 2e-2, f16 3e-3 (output
rounding to 8 / 11 significand bits + f32-accumulate order).  The split-f16 matrix mode of f32 tensors
(NLC_MATH_F16X3) is compared with an f64 evaluation on the EXACT operands at 5e-6.
"""
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
DTYPE_IDS = ["f32", "bf16", "f16"]
T16 = [torch.bfloat16, torch.float16]
T16_IDS = ["bf16", "f16"]


def _seed(obj) -> int:
    """A seed that is the same in every process (hash() of a str is salted per interpreter)."""
    return zlib.crc32(repr(obj).encode()) & 0x7fffffff


def _dev():
    return torch.device("cuda:0")


def _totals(st):
    """Ride-along GroupNorm statistics, int64 [B, C/g, 4] = (sum.hi, sum.lo, sumsq.hi, sumsq.lo) fixed-point totals
    (include/nlc_hip.h: nlc_conv_desc.stats_out) -> double [B, C/g, 2] = (sum, sum of squares) on the CPU."""
    assert st.dtype == torch.int64 and st.shape[-1] == 4
    t = st.cpu().double()
    return torch.stack([t[..., 0] + t[..., 1] * 2.0 ** -44, t[..., 2] + t[..., 3] * 2.0 ** -44], dim=-1)


def _tol(dtype):
    return {torch.float32: 2e-4, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]


def _rt(x, dtype):
    """round-trip through the compute dtype (what the kernel actually sees)."""
    return x.to(dtype).to(torch.float32)


def _nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(device=_dev(), dtype=dtype)


def _close(got, ref, tol, what=""):
    got = got.detach().float().cpu()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item() / scale
    assert err <= tol, f"{what}: max rel-to-scale err {err:.3e} > {tol:.1e} (scale {scale:.3e})"


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, extras
    dict(B=2, Cin=64, H=16, W=16, Cout=128, k=3),
    dict(B=1, Cin=32, H=9, W=7, Cout=6, k=3),                        # ragged spatial, tiny Cout, Cin < k-block
    dict(B=2, Cin=96, H=8, W=8, Cout=200, k=3, bias=False),          # Cin, Cout not tile multiples
    dict(B=3, Cin=128, H=4, W=4, Cout=128, k=1),                     # 1x1
    dict(B=2, Cin=64, H=16, W=16, Cout=64, k=3, stride=2, pad=1),    # ADM Downsample conv (unet_adm.py:131)
    dict(B=2, Cin=64, H=16, W=16, Cout=64, k=3, stride=2, pad=0, asym=True),   # simple Downsample (unet_simple.py:67-71)
    dict(B=2, Cin=64, H=8, W=8, Cout=64, k=3, ups=True),             # conv after nearest 2x (unet_adm.py:107-109)
    dict(B=2, Cin=64, H=8, W=8, Cout=96, k=3, split=24),             # cat(x0, x1) input, 24 + 40 channels
    dict(B=2, Cin=160, H=8, W=8, Cout=64, k=1, split=96),            # 1x1 skip over a concat, crosses a k-block
    dict(B=2, Cin=64, H=8, W=8, Cout=64, k=3, emb=True, res=True, scale=math.sqrt(0.5)),
    dict(B=2, Cin=64, H=8, W=8, Cout=64, k=3, act="silu"),
    dict(B=2, Cin=64, H=8, W=8, Cout=16, k=3, act="gelu", nchw=True),
    dict(B=1, Cin=256, H=32, W=32, Cout=256, k=3),                   # multi-tile M and N
    dict(B=2, Cin=64, H=2, W=2, Cout=64, k=3),                       # 2x2 image (sigma net tail)
    # shapes eligible for the LDS-halo kernel (H, W multiples of 16, whole channel blocks; forced by tests/conftest.py)
    dict(B=2, Cin=96, H=16, W=32, Cout=200, k=3, split=40, emb=True, res=True, scale=math.sqrt(0.5)),
    dict(B=1, Cin=32, H=32, W=16, Cout=6, k=3, act="silu", nchw=True),
    dict(B=3, Cin=192, H=16, W=16, Cout=128, k=3, act="gelu"),
    dict(B=1, Cin=128, H=48, W=32, Cout=256, k=3, bias=False),
    dict(B=2, Cin=192, H=16, W=32, Cout=200, k=3, split=64, emb=True, res=True, scale=math.sqrt(0.5)),   # concat at a block edge
    dict(B=2, Cin=64, H=32, W=32, Cout=6, k=3, act="silu", nchw=True),
    dict(B=2, Cin=128, H=16, W=24, Cout=128, k=3, ups=True, emb=True, res=True),        # fused nearest-2x upsample in the halo kernel
    # > 256 tiles: every persistent workgroup walks 2 tiles (cross-tile DMA streams, accumulator re-init, both N-tiles)
    dict(B=1, Cin=64, H=256, W=256, Cout=256, k=3, emb=True, res=True, scale=math.sqrt(0.5)),
    dict(B=2, Cin=128, H=256, W=128, Cout=256, k=3, split=64, act="silu"),
    # few output tiles + long K: split-K in the 16-bit types (2 / 4 / 2 splits), partials reduced by the last-arriving workgroup
    dict(B=2, Cin=256, H=8, W=8, Cout=128, k=3, emb=True, res=True, scale=math.sqrt(0.5)),
    dict(B=1, Cin=512, H=8, W=8, Cout=72, k=3, act="silu", nchw=True),
    dict(B=2, Cin=1024, H=4, W=4, Cout=256, k=1, res=True),
    # 16x16 / 32x32 maps with too few tiles for the halo kernel under the production dispatch: conv_fast<9> with split-K
    # (4 / 8 / 2 splits in bf16; f32 never splits) and without it - the 16x16 / 32x32 levels of ADM-256
    dict(B=2, Cin=512, H=16, W=16, Cout=256, k=3, emb=True, res=True),
    dict(B=1, Cin=1024, H=16, W=16, Cout=384, k=3, split=512, act="silu"),
    dict(B=2, Cin=256, H=32, W=32, Cout=128, k=3, res=True, scale=math.sqrt(0.5)),
    dict(B=4, Cin=64, H=32, W=32, Cout=128, k=3, act="silu"),
    dict(B=2, Cin=128, H=16, W=16, Cout=256, k=3, ups=True, res=True),                 # fused upsample outside the halo kernel
    # widths / areas that are no powers of two: the tap tables divide by Wout and Hout*Wout through host-computed multiply-shift
    # constants (conv_params.h: FastDiv), tiles straddle image boundaries, strided and upsampled variants
    dict(B=5, Cin=64, H=11, W=13, Cout=128, k=3, res=True),
    dict(B=3, Cin=64, H=10, W=6, Cout=64, k=3, stride=2, pad=1),
    dict(B=7, Cin=128, H=5, W=3, Cout=128, k=3, ups=True),
    dict(B=33, Cin=64, H=3, W=1, Cout=64, k=3),                      # Wout = 1: the d = 1 special case
    dict(B=3, Cin=64, H=100, W=1, Cout=64, k=1, res=True),
]


def _run_conv_case(case, dtype, math_mode="native"):
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(_seed(case))
    B, Cin, H, W, Cout, k = (case[x] for x in ("B", "Cin", "H", "W", "Cout", "k"))
    stride, pad = case.get("stride", 1), case.get("pad", k // 2)
    per = 4 if dtype == torch.float32 else 8
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) * 0.1 if case.get("bias", True) else None
    xr, wr = _rt(x, dtype), _rt(w, dtype)
    x3 = math_mode == "f16x3"
    if x3:                    # reference in f64 on the exact operands: the mode claims near-f32 accuracy, not f16 accuracy
        xr, wr, b = x.double(), w.double(), None if b is None else b.double()
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if case.get("ups") else xr
    if case.get("asym"):
        xin = F.pad(xin, (0, 1, 0, 1))
    ref = F.conv2d(xin, wr, b, stride=stride, padding=pad)
    emb = res = None
    if case.get("emb"):
        emb = torch.randn(B, Cout + 8, generator=g)
        ref = ref + emb[:, :Cout, None, None]
    if case.get("res"):
        res = torch.randn(B, Cout, ref.shape[2], ref.shape[3], generator=g)
        ref = ref + _rt(res, dtype)
    ref = ref * case.get("scale", 1.0)
    act = {"silu": 1, "gelu": 2}.get(case.get("act"), 0)
    if act == 1:
        ref = F.silu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    ref = ref.float()

    pw = ops.pack_conv(w, None if b is None else b.float(), dtype, _dev(), math=math_mode)
    split = case.get("split")
    if split and split % per:
        pytest.skip("split not aligned for this dtype")
    x0 = _nhwc(x[:, :split] if split else x, dtype)
    x1 = _nhwc(x[:, split:], dtype) if split else None
    out_hw = (ref.shape[2], ref.shape[3])
    got = ops.conv2d(x0, pw, x1=x1, stride=stride, pad=(pad, pad), out_hw=out_hw, upsample2x=bool(case.get("ups")),
                     emb=None if emb is None else emb.to(_dev())[:, :], res=None if res is None else _nhwc(res, dtype),
                     out_scale=case.get("scale", 1.0), act=act, out_nchw_f32=bool(case.get("nchw")))
    torch.cuda.synchronize()
    if not case.get("nchw"):
        got = got.permute(0, 3, 1, 2)
    _close(got, ref, 5e-6 if x3 else _tol(dtype), "conv2d" + (" (f16x3)" if x3 else ""))


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d(case, dtype, conv_policy):
    _run_conv_case(case, dtype)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d_split_f16_math(case, conv_policy):
    """NLC_MATH_F16X3 on f32 tensors: every kernel that can take the launch (halo kernel with the in-LDS operand split, conv_fast
    with the in-register split, the generic kernel rebuilding hi + lo) against an f64 reference on the exact operands."""
    _run_conv_case(case, torch.float32, "f16x3")


X3_WEIGHTS = {
    # magnitudes over 10 binades inside every row (exercises hi / lo of elements far below the row's maximum)
    "binades": lambda g, shape: torch.randn(shape, generator=g) * torch.exp2(torch.randint(-10, 0, shape, generator=g).float()),
    # what the networks carry: filler / trained weights of magnitude 1e-2 ... 1e-4.  UNSCALED, hi = f16(w) keeps 11 bits but lo falls
    # into f16's subnormal range (absolute 2^-25): N(0, 0.02^2) carried ~19 bits, N(0, 1e-4^2) ~12.  The per-row power-of-two scale
    # of nlc_pack_conv_weights_ex (row maximum -> [2^14, 2^15)) restores the full 22.
    "n0.02": lambda g, shape: torch.randn(shape, generator=g) * 0.02,
    "n1e-4": lambda g, shape: torch.randn(shape, generator=g) * 1e-4,
    # rows of wildly different magnitude (per-ROW scaling, not per tensor), one all-zero row, one row with a single non-zero weight
    "rows": lambda g, shape: _x3_rows(g, shape),
}


def _x3_rows(g, shape):
    w = torch.randn(shape, generator=g) * torch.exp2(torch.randint(-30, 20, (shape[0], 1, 1, 1), generator=g).float())
    w[3] = 0.0
    w[5] = 0.0
    w[5, 7, 1, 1] = 3.0e-7
    return w


@pytest.mark.parametrize("wkind", sorted(X3_WEIGHTS))
@pytest.mark.parametrize("policy", ["auto", "halo", "no_halo", "generic"])
def test_split_f16_math_carries_22_bits(policy, wkind):
    """Adversarial operands for the operand split: activation magnitudes over 12 binades (the lo halves of the small ones are f16
    subnormals), one input channel block where only lo halves are non-zero (values below half an f16 ulp of nothing: exact powers of
    two times (1 + 2^-12)); weights of four kinds (X3_WEIGHTS), against f64, per OUTPUT ROW relative to that row's largest output
    (a per-row scale must not let a large row hide a small one).  A kernel that dropped a cross term, flushed subnormal halves,
    mis-paired hi / lo lanes or mis-applied the row scale fails this by orders of magnitude (bf16: 4e-3, plain f16: 5e-4,
    this mode: < 2e-6)."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(123)
    B, Cin, H, W, Cout = 2, 64, 16, 16, 128
    mag = torch.exp2(torch.randint(-8, 4, (B, Cin, H, W), generator=g).float())
    x = torch.randn(B, Cin, H, W, generator=g) * mag
    x[:, 32:48] = (1 + 2.0 ** -12) * torch.exp2(torch.randint(-3, 3, (B, 16, H, W), generator=g).float())     # hi = 2^k, lo = 2^(k-12)
    w = X3_WEIGHTS[wkind](g, (Cout, Cin, 3, 3))
    bias = torch.randn(Cout, generator=g) * w.abs().amax(dim=(1, 2, 3))          # the bias must NOT be scaled with the row
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    old = ops.CONV_POLICY
    ops.CONV_POLICY = policy
    try:
        pw = ops.pack_conv(w, bias, torch.float32, _dev(), math="f16x3")
        got = ops.conv2d(_nhwc(x, torch.float32), pw)
        exact = ops.conv2d(_nhwc(x, torch.float32), ops.pack_conv(w, bias, torch.float32, _dev()))
    finally:
        ops.CONV_POLICY = old
    # the row factors are powers of two that bring each row's maximum into [2^14, 2^15) (1 for an all-zero row)
    ws = pw.w_scale.cpu()[:Cout].double()
    assert torch.equal(torch.exp2(torch.round(torch.log2(ws))), ws)
    wmax = w.abs().amax(dim=(1, 2, 3)).double()
    nz = wmax > 0
    scaled = wmax[nz] / ws[nz]
    assert (scaled >= 2.0 ** 14).all() and (scaled < 2.0 ** 15).all() and (ws[~nz] == 1).all()
    scale = ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-300)                      # per output channel
    e3 = ((got.permute(0, 3, 1, 2).double().cpu() - ref).abs().amax(dim=(0, 2, 3)) / scale).max().item()
    e1 = ((exact.permute(0, 3, 1, 2).double().cpu() - ref).abs().amax(dim=(0, 2, 3)) / scale).max().item()
    assert e3 < 2e-6, f"f16x3 under policy {policy}, weights {wkind}: {e3:.3e} (exact-f32 MFMA on the same data: {e1:.3e})"


@pytest.mark.parametrize("policy", ["auto", "halo", "no_halo", "generic"])
def test_split_f16_math_domain_is_guarded(policy):
    """|x| >= 65504 is outside the domain of the f16 operand split (include/nlc_hip.h).  In production such an input must not
    poison the sum: the kernels run with FP16_OVFL set, so the value saturates and the output stays FINITE (round 3 produced
    inf - inf = NaN); with nlc_conv_desc.debug bit 1 the launch is refused loudly instead.  Values just inside the domain are exact
    to the mode's accuracy."""
    from diffusion_nlc_amd import ops
    from diffusion_nlc_amd._ext import NlcError
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 1, 64, 16, 16, 128
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02
    x = torch.randn(B, Cin, H, W, generator=g)
    x[0, 3, 5, 5] = 6.0e4                                       # inside
    old, old_dbg = ops.CONV_POLICY, ops.CONV_DEBUG
    ops.CONV_POLICY = policy
    try:
        pw = ops.pack_conv(w, None, torch.float32, _dev(), math="f16x3")
        ops.CONV_DEBUG = 2
        got = ops.conv2d(_nhwc(x, torch.float32), pw)           # passes the domain check
        ref = F.conv2d(x.double(), w.double(), None, padding=1)
        err = (got.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, err
        for bad in (1.0e5, -3.0e5, 1.0e30):
            xb = x.clone()
            xb[0, 9, 8, 8] = bad
            ops.CONV_DEBUG = 0
            out = ops.conv2d(_nhwc(xb, torch.float32), pw)
            assert torch.isfinite(out).all(), f"input {bad:g} under policy {policy} gave a non-finite output"
            if policy != "generic":                              # (the generic kernel never splits activations: it stays exact)
                # saturation: the out-of-domain element acts as hi + lo with both halves clamped to +-65504
                sat = max(min(bad, 65504.0), -65504.0)
                sat = sat + max(min(bad - sat, 65504.0), -65504.0)
                xs = xb.clone()
                xs[0, 9, 8, 8] = sat
                refs = F.conv2d(xs.double(), w.double(), None, padding=1)
                errs = (out.permute(0, 3, 1, 2).double().cpu() - refs).abs().max().item() / refs.abs().max().item()
                assert errs < 1e-3, (bad, errs)
            ops.CONV_DEBUG = 2
            with pytest.raises(NlcError, match="outside the domain"):
                ops.conv2d(_nhwc(xb, torch.float32), pw)
        ops.CONV_DEBUG = 2
        xn = x.clone()
        xn[0, 0, 0, 0] = float("nan")
        with pytest.raises(NlcError, match="outside the domain"):
            ops.conv2d(_nhwc(xn, torch.float32), pw)
    finally:
        ops.CONV_POLICY, ops.CONV_DEBUG = old, old_dbg


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_linear(dtype):
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(5, 256, generator=g)
    w = torch.randn(192, 256, generator=g) / 16
    b = torch.randn(192, generator=g)
    pw = ops.pack_conv(w, b, dtype, _dev())
    got = ops.conv2d(x.to(_dev(), dtype), pw, act=1)
    ref = F.silu(F.linear(_rt(x, dtype), _rt(w, dtype), b))
    _close(got, ref, _tol(dtype), "linear")


GN_CASES = [
    dict(B=2, C=64, HW=64, G=32),
    dict(B=2, C=256, HW=1024, G=32, silu=True),
    dict(B=1, C=768, HW=256, G=32, silu=True),               # 96 chunks per pixel (not a power of two)
    dict(B=3, C=64, HW=1, G=32),                             # single pixel
    dict(B=2, C=64, HW=36, G=4, split=40, silu=True),        # group 2 straddles the concat boundary
    dict(B=2, C=128, HW=64, G=32, film=True, silu=True),     # scale/shift (unet_adm.py:248-252)
    dict(B=2, C=32, HW=16, G=8, eps=1e-6),                   # EDM groups=min(32,C/4) (edm_networks.py:108)
    dict(B=1, C=2048, HW=64, G=32, silu=True),               # widest ADM-256 concat
    dict(B=1, C=256, HW=65536, G=32, silu=True),             # ADM-256 full-resolution map, 64 stat blocks
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", GN_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_groupnorm(case, dtype):
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(11)
    B, Cc, HW, G = case["B"], case["C"], case["HW"], case["G"]
    eps = case.get("eps", 1e-5)
    x = torch.randn(B, Cc, HW, generator=g) * 1.7 + 0.3
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    beta = 0.1 * torch.randn(Cc, generator=g)
    xr = _rt(x, dtype)
    ref = F.group_norm(xr, G, gamma, beta, eps)
    scale = shift = None
    if case.get("film"):
        ss = torch.randn(B, 2 * Cc, generator=g) * 0.3
        scale, shift = ss[:, :Cc], ss[:, Cc:]
        ref = ref * (1 + scale[:, :, None]) + shift[:, :, None]
        ss_d = ss.to(_dev())
        scale, shift = ss_d[:, :Cc], ss_d[:, Cc:]
    if case.get("silu"):
        ref = F.silu(ref)
    xd = x.permute(0, 2, 1).contiguous().to(_dev(), dtype)
    split = case.get("split")
    x0 = xd[..., :split].contiguous() if split else xd
    x1 = xd[..., split:].contiguous() if split else None
    got = ops.groupnorm(x0, gamma.to(_dev()), beta.to(_dev()), groups=G, eps=eps, silu=bool(case.get("silu")),
                        x1=x1, scale=scale, shift=shift)
    _close(got.permute(0, 2, 1), ref, _tol(dtype) * (1 if dtype == torch.float32 else 1.5), "groupnorm")


GNPOOL_CASES = [
    dict(B=2, C=64, H=8, W=8, G=32, silu=True),
    dict(B=2, C=256, H=32, W=32, G=32, silu=True),               # > 8 statistics blocks: the finalize launch
    dict(B=1, C=512, H=16, W=48, G=32, silu=True, film=True),    # ragged map, FiLM
    dict(B=3, C=32, H=2, W=2, G=8, silu=False),                  # one output pixel
    dict(B=1, C=1024, H=16, W=16, G=32, silu=True),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", GNPOOL_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_groupnorm_pool2x2(case, dtype):
    """nlc_groupnorm_pool2x2 = (AvgPool2d(2)(act(GroupNorm(x))), AvgPool2d(2)(x)) - both branches of a down-sampling ResBlock
    (src/unet_adm.py:193-195) - against torch, and against the two separate passes it replaces (f32: bit-identical)."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(23)
    B, Cc, H, W, G = case["B"], case["C"], case["H"], case["W"], case["G"]
    x = torch.randn(B, Cc, H, W, generator=g) * 1.7 + 0.3
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    beta = 0.1 * torch.randn(Cc, generator=g)
    xr = _rt(x, dtype)
    ref = F.group_norm(xr, G, gamma, beta, 1e-5)
    scale = shift = None
    if case.get("film"):
        ss = torch.randn(B, 2 * Cc, generator=g) * 0.3
        ref = ref * (1 + ss[:, :Cc, None, None]) + ss[:, Cc:, None, None]
        ss_d = ss.to(_dev())
        scale, shift = ss_d[:, :Cc], ss_d[:, Cc:]
    if case.get("silu"):
        ref = F.silu(ref)
    ref_h, ref_x = F.avg_pool2d(ref, 2), F.avg_pool2d(xr, 2)
    xd = _nhwc(x, dtype)
    gd, bd = gamma.to(_dev()), beta.to(_dev())
    got_h, got_x = ops.groupnorm_pool2x2(xd, gd, bd, groups=G, eps=1e-5, silu=bool(case.get("silu")), scale=scale, shift=shift)
    tol = _tol(dtype) * (1 if dtype == torch.float32 else 1.5)
    _close(got_h.permute(0, 3, 1, 2), ref_h, tol, "groupnorm_pool2x2 h")
    _close(got_x.permute(0, 3, 1, 2), ref_x, tol, "groupnorm_pool2x2 x")
    sep_h = ops.avgpool2x2(ops.groupnorm(xd, gd, bd, groups=G, eps=1e-5, silu=bool(case.get("silu")), scale=scale, shift=shift))
    sep_x = ops.avgpool2x2(xd)
    assert torch.equal(got_x, sep_x)
    if dtype == torch.float32:
        assert torch.equal(got_h, sep_h)
    else:
        assert (got_h.float() - sep_h.float()).abs().max().item() <= 2e-2 * ref_h.abs().max().item()


def test_groupnorm_pool2x2_with_ride_along_statistics():
    """bf16 production route: the statistics come from the producing convolution's epilogue (halo kernel, 256 tiles)."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(29)
    B, Cin, H, W, Cout = 16, 64, 32, 32, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    y = ops.conv2d(_nhwc(x, torch.bfloat16), ops.pack_conv(w, torch.zeros(Cout), torch.bfloat16, _dev()))
    assert getattr(y, "_nlc_stats", None) is not None
    gamma, beta = torch.randn(Cout, generator=g).to(_dev()), torch.randn(Cout, generator=g).to(_dev())
    h, xp = ops.groupnorm_pool2x2(y, gamma, beta, groups=32, eps=1e-5, silu=True)
    yf = y.float().cpu().permute(0, 3, 1, 2)
    ref_h = F.avg_pool2d(F.silu(F.group_norm(yf, 32, gamma.cpu(), beta.cpu(), eps=1e-5)), 2)
    _close(h.permute(0, 3, 1, 2), ref_h, 1.5e-2, "groupnorm_pool2x2 (ride-along statistics) h")
    _close(xp.permute(0, 3, 1, 2), F.avg_pool2d(yf, 2), 1e-2, "groupnorm_pool2x2 (ride-along statistics) x")
    ops.FUSED_GN_STATS = False
    try:
        h2, _ = ops.groupnorm_pool2x2(y, gamma, beta, groups=32, eps=1e-5, silu=True)
    finally:
        ops.FUSED_GN_STATS = True
    assert (h.float() - h2.float()).abs().max().item() <= 2e-2 * ref_h.abs().max().item()


ATTN_CASES = [
    dict(B=2, T=64, H=2, D=32), dict(B=1, T=256, H=4, D=64), dict(B=2, T=100, H=1, D=64),
    dict(B=2, T=4, H=2, D=64), dict(B=1, T=1024, H=2, D=64), dict(B=1, T=80, H=1, D=128),
    dict(B=1, T=48, H=1, D=256), dict(B=1, T=70, H=1, D=512),
    # 64 channels per head, T a multiple of 256: bf16 takes the register-resident kernel (attn_d64_kernel)
    dict(B=2, T=512, H=3, D=64), dict(B=1, T=768, H=2, D=64, spike=True), dict(B=1, T=512, H=2, D=64, allneg=True),
    # wide single heads over several key tiles (K / V tiles by LDS-DMA, double buffered, rows >= 256 bytes): the cfg 3 / cfg 4 shapes,
    # a ragged last tile (keys beyond T are fetched from the last key and masked), f32 rows of 256 / 512 / 1024 bytes
    dict(B=2, T=256, H=1, D=512), dict(B=3, T=256, H=1, D=256), dict(B=1, T=1024, H=2, D=128), dict(B=2, T=200, H=1, D=256),
    dict(B=1, T=330, H=1, D=512), dict(B=1, T=300, H=2, D=64, spike=True),
    dict(B=130, T=256, H=1, D=128),      # more than two workgroups per CU: the synchronous-load form of the same kernel
]


@pytest.mark.parametrize("base2", [False, True], ids=["natural-logits", "log2-logits"])
@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", ATTN_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_attention(case, dtype, base2):
    """``base2``: the caller folded log2(e) into q (what 16-bit models do at pack time) and the kernel exponentiates with 2^x.  The
    register-resident kernel (D = 64, T % 256 == 0, 16-bit) subtracts no running maximum while every query of a wave stays within
    2^+-64: the `spike` case drives some waves out of that range mid-row (general path: offset, rescale of O and l) and
    `allneg` starts a row far below it (first-tile offset), both against the plain softmax."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, H, D = case["B"], case["T"], case["H"], case["D"]
    qkv = torch.randn(B, T, 3, H, D, generator=g)
    qkv[:, :, :2] *= D ** -0.25 * 1.5          # realistic logit scale after the folded ch^-1/4
    if case.get("spike"):
        # force the online-softmax rescale branch late in the key loop: a few keys far above the running maximum of
        # every query (tile 9 of 12), and one query row that also spikes against them
        qkv[:, 600:603, 1] *= 6.0
        qkv[:, 17, 0] *= 4.0
    if case.get("allneg"):
        # every logit of some queries far below zero (~ -150 in log2 units): without a first-tile offset 2^s underflows to l = 0
        qkv[:, :, 1, :, 0] = 6.0
        qkv[:, 5:9, 0, :, :] = 0.0
        qkv[:, 5:9, 0, :, 0] = -18.0
    if base2:
        qkv[:, :, 0] *= 1.4426950408889634      # log2(e) folded into q BEFORE the rounding to the compute dtype, as pack_conv does
    r = _rt(qkv, dtype)
    q, k, v = r[:, :, 0], r[:, :, 1], r[:, :, 2]          # [B,T,H,D]
    s = torch.einsum("bthd,bshd->bhts", q, k)
    p = torch.softmax(s * (0.6931471805599453 if base2 else 1.0), dim=-1)      # softmax_2(s) = softmax(s ln 2)
    ref = torch.einsum("bhts,bshd->bthd", p, v).reshape(B, T, H * D)
    got = ops.attention(qkv.reshape(B, T, 3 * H * D).to(_dev(), dtype), H, base2=base2)
    _close(got, ref, _tol(dtype) * (1 if dtype == torch.float32 else 1.5), "attention")


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_resample_and_layout(dtype):
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 6, 10, generator=g)
    xr = _rt(x, dtype)
    xd = _nhwc(x, dtype)
    _close(ops.avgpool2x2(xd).permute(0, 3, 1, 2), F.avg_pool2d(xr, 2), _tol(dtype), "avgpool")
    _close(ops.upsample2x(xd).permute(0, 3, 1, 2), F.interpolate(xr, scale_factor=2, mode="nearest"), 1e-7, "upsample")
    _close(ops.pad_rb(xd).permute(0, 3, 1, 2), F.pad(xr, (0, 1, 0, 1)), 1e-7, "pad_rb")
    _close(ops.nhwc_to_nchw_f32(xd), xr, 1e-7, "nhwc->nchw")
    back = ops.nchw_f32_to_nhwc(x.to(_dev()), dtype)
    _close(back.permute(0, 3, 1, 2), xr, 1e-7, "nchw->nhwc")
    # odd, non-multiple-of-32 sizes through the transpose tiles
    y = torch.randn(3, 40, 5, 7, generator=g)
    _close(ops.nhwc_to_nchw_f32(_nhwc(y, dtype)), _rt(y, dtype), 1e-7, "nhwc->nchw ragged")


def test_timestep_embedding():
    from diffusion_nlc_amd import ops
    t = torch.tensor([0.0, 1.0, 17.5, 954.0, 1000.0])
    half = 64
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    got = ops.timestep_embedding(t.to(_dev()), freqs.to(_dev()), sin_first=False)
    assert (got.cpu() - ref).abs().max().item() < 2e-6
    got = ops.timestep_embedding(t.to(_dev()), freqs.to(_dev()), sin_first=True)
    assert (got.cpu() - torch.cat([torch.sin(args), torch.cos(args)], dim=-1)).abs().max().item() < 2e-6


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_conv_first(dtype):
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 20, 33, generator=g)
    w = torch.randn(40, 3, 3, 3, generator=g) / 5
    b = torch.randn(40, generator=g) * 0.1
    sc = torch.tensor([0.5, 1.25])
    ref = F.conv2d(x * sc[:, None, None, None], w, b, padding=1)
    wp = w.permute(0, 2, 3, 1).reshape(40, 9, 3).contiguous().to(_dev())
    got = ops.conv_first(x.to(_dev()), wp, b.to(_dev()), dtype, in_scale=sc.to(_dev()))
    _close(got.permute(0, 3, 1, 2), ref, {torch.float32: 1e-5, torch.bfloat16: 1e-2, torch.float16: 2e-3}[dtype], "conv_first")
    # ADM-like first layer: whole 64-pixel tiles, 256 channels -> bf16 takes the matrix-core kernel and emits
    # GroupNorm statistics of the stored output
    x = torch.randn(2, 3, 32, 64, generator=g)
    w = torch.randn(256, 3, 3, 3, generator=g) / 5
    b = torch.randn(256, generator=g) * 0.1
    ref = F.conv2d(x * sc[:, None, None, None], w, b, padding=1)
    wp = w.permute(0, 2, 3, 1).reshape(256, 9, 3).contiguous().to(_dev())
    got = ops.conv_first(x.to(_dev()), wp, b.to(_dev()), dtype, in_scale=sc.to(_dev()))
    _close(got.permute(0, 3, 1, 2), ref, {torch.float32: 1e-5, torch.bfloat16: 1e-2, torch.float16: 2e-3}[dtype], "conv_first 256")
    st = getattr(got, "_nlc_stats", None)
    assert (st is not None) == (dtype != torch.float32)
    if st is not None:
        ch = got.float().cpu().view(2, 32 * 64, 32, 8)
        assert (_totals(st)[..., 0] - ch.double().sum(dim=(1, 3))).abs().max() < 5e-2
        assert ((_totals(st)[..., 1] - (ch.double() ** 2).sum(dim=(1, 3))) / (ch.double() ** 2).sum(dim=(1, 3))).abs().max() < 5e-4


def test_row_sumsq_and_quantile():
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 6, 17, 19, generator=g)
    ss = ops.row_sumsq(x.to(_dev())).cpu()
    ref = (x.reshape(3, -1) ** 2).sum(1)
    assert ((ss - ref).abs() / ref).max().item() < 1e-5
    ss3 = ops.row_sumsq(x.to(_dev()), d_used=3 * 17 * 19).cpu()
    ref3 = (x[:, :3].reshape(3, -1) ** 2).sum(1)
    assert ((ss3 - ref3).abs() / ref3).max().item() < 1e-5
    # exact order statistics: must equal torch.quantile bit for bit before the clamp
    for shape, q in (((4, 3, 32, 32), 0.99), ((2, 3, 64, 64), 0.99), ((3, 1, 5, 5), 0.5), ((2, 3, 256, 256), 0.99)):
        y = torch.randn(*shape, generator=g) * 2.5
        y[0].mul_(0.2)                                   # below the clamp floor of 1
        ref = torch.quantile(y.reshape(shape[0], -1).abs(), q, dim=1).clamp(min=1, max=100)
        for single in (False, True, False):              # G workgroups per sample (twice: the workspace must come back zeroed), and one
            got = ops.dynamic_threshold(y.to(_dev()), q, 100.0, single_workgroup=single).cpu()
            assert torch.equal(got, ref), (shape, single, got, ref)
    # ties everywhere
    z = torch.ones(2, 3, 8, 8) * 3.0
    for single in (False, True):
        assert torch.equal(ops.dynamic_threshold(z.to(_dev()), 0.99, 100.0, single_workgroup=single).cpu(), torch.tensor([3.0, 3.0]))
    # the benchmark shape, heavy ties (a clipped image), q at both ends
    y = (torch.randn(16, 3, 256, 256, generator=g) * 0.7).clamp(-1, 1)
    for q in (0.995, 0.0, 1.0, 0.37):
        ref = torch.quantile(y.reshape(16, -1).abs(), q, dim=1).clamp(min=1, max=100)
        ref_raw = torch.quantile(y.reshape(16, -1).abs() + 1.0, q, dim=1).clamp(min=1, max=100)
        assert torch.equal(ops.dynamic_threshold(y.to(_dev()), q, 100.0).cpu(), ref)
        assert torch.equal(ops.dynamic_threshold((y.abs() + 1.0).to(_dev()), q, 100.0).cpu(), ref_raw)      # above the clamp floor


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
def test_conv_emits_groupnorm_statistics_and_groupnorm_uses_them(conv_policy, t16):
    """bf16 conv (LDS-halo kernel when forced, conv_fast<9> under the production dispatch): the epilogue's per-8-channel
    (sum, sumsq) partials describe the STORED output exactly enough, and GroupNorm fed with them (two passes) agrees with the
    three-pass GroupNorm and with the f32 reference."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(11)
    B, Cin, H, W, Cout = 2, 64, 32, 48, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.3 + 0.5                    # non-zero mean: E[x^2]-E[x]^2 is exercised
    res = torch.randn(B, Cout, H, W, generator=g)
    pw = ops.pack_conv(w, b, t16, _dev())
    y = ops.conv2d(_nhwc(x, t16), pw, res=_nhwc(res, t16), act=1)
    st = getattr(y, "_nlc_stats", None)
    # partials per image: 4 per 16x16 patch from the halo kernel, 2 per 128-pixel tile from conv_fast
    P = (H // 16) * (W // 16) * 4 if conv_policy == "halo" else (H * W // 128) * 2
    assert st is not None and st.shape == (B, Cout // 8, 4)
    yf = y.float().cpu()                                              # [B,H,W,C] stored values
    chunks = yf.view(B, H * W, Cout // 8, 8)
    ref_sum, ref_sq = chunks.sum(dim=(1, 3)), (chunks.double() ** 2).sum(dim=(1, 3))
    got = _totals(st)
    assert (got[..., 0] - ref_sum.double()).abs().max() < 1e-2 * ref_sum.abs().max().clamp(min=1.0)
    # (every emitter sums the STORED, bf16-rounded values; what is left is f32 summation order)
    assert ((got[..., 1] - ref_sq) / ref_sq).abs().max() < 5e-4
    # a second conv makes the skip source of a concatenated GroupNorm input; 384 channels / 32 groups = 12: not a
    # multiple of 8 -> falls back; 256 + 256 = 512 -> group size 16, fused
    y2 = ops.conv2d(_nhwc(x, t16), pw)
    gamma, beta = torch.randn(2 * Cout, generator=g).to(_dev()), torch.randn(2 * Cout, generator=g).to(_dev())
    fused = ops.groupnorm(y, gamma, beta, groups=32, eps=1e-5, silu=True, x1=y2)
    ops.FUSED_GN_STATS = False
    try:
        plain = ops.groupnorm(y, gamma, beta, groups=32, eps=1e-5, silu=True, x1=y2)
    finally:
        ops.FUSED_GN_STATS = True
    # split-K shape (8x8 level; not halo-eligible): the statistics come from the last-arriving split workgroup, one partial per pixel
    xs = torch.randn(2, 512, 8, 8, generator=g)
    ws = torch.randn(256, 512, 3, 3, generator=g) / math.sqrt(512 * 9)
    ys = ops.conv2d(_nhwc(xs, t16), ops.pack_conv(ws, b, t16, _dev()))
    sts = getattr(ys, "_nlc_stats", None)
    assert sts is not None and sts.shape == (2, 32, 4)
    chs = ys.float().cpu().view(2, 64, 32, 8)
    ref_s = chs.double().sum(dim=(1, 3))              # 512 stored values per chunk
    assert (_totals(sts)[..., 0] - ref_s).abs().max() < 2e-3 * ref_s.abs().max()
    assert ((_totals(sts)[..., 1] - (chs.double() ** 2).sum(dim=(1, 3))) / (chs.double() ** 2).sum(dim=(1, 3))).abs().max() < 2e-3
    cat = torch.cat([yf, y2.float().cpu()], dim=-1).permute(0, 3, 1, 2)
    ref = F.silu(F.group_norm(cat, 32, gamma.cpu(), beta.cpu(), eps=1e-5)).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    assert (fused.float().cpu() - plain.float().cpu()).abs().max().item() <= 2e-2 * scale      # one bf16 ulp of slack
    assert (fused.float().cpu() - ref).abs().max().item() <= 2e-2 * scale


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("bad", [float("nan"), float("inf")], ids=["nan", "inf"])
def test_non_finite_outputs_poison_the_ride_along_statistics(conv_policy, t16, bad):
    """A convolution whose stored output holds an inf / NaN marks the chunk (bit 62 of sumsq.hi, Stat16::poison) instead of
    converting a non-finite float to an integer; GroupNorm fed with the totals then makes exactly the groups F.group_norm makes
    NaN - the whole group of the affected image - and leaves every other (image, group) as it was."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 2, 64, 32, 32, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    res = torch.zeros(B, Cout, H, W)
    res[1, 40, 7, 9] = bad                                               # image 1, channel 40 -> group 5 of 32 (8 channels wide)
    pw = ops.pack_conv(w, None, t16, _dev())
    y = ops.conv2d(_nhwc(x, t16), pw, res=_nhwc(res, t16))
    st = ops.ride_stats(y)
    assert st is not None
    marks = (st[..., 2] >> 62) & 1
    want = torch.zeros_like(marks)
    want[1, 40 // (Cout // st.shape[1])] = 1
    assert torch.equal(marks.cpu(), want.cpu())
    gamma, beta = torch.randn(Cout, generator=g).to(_dev()), torch.randn(Cout, generator=g).to(_dev())
    fused = ops.groupnorm(y, gamma, beta, groups=32, eps=1e-5, silu=False).float().cpu()
    ref = F.group_norm(y.float().cpu().permute(0, 3, 1, 2), 32, gamma.cpu(), beta.cpu(), eps=1e-5).permute(0, 2, 3, 1)
    assert torch.equal(torch.isnan(fused), torch.isnan(ref))
    assert torch.isnan(fused[1, :, :, 40:48]).all() and not torch.isnan(fused[0]).any()
    ok = ~torch.isnan(ref)
    assert (fused[ok] - ref[ok]).abs().max().item() <= _tol(t16) * ref[ok].abs().max().item()


STAT_CASES = [
    # (B, Cin, H, W, Cout, expected partials per image under the production dispatch, what emits them)
    (2, 512, 16, 16, 256, 256, "split-K (last arriver): one partial per pixel"),
    (1, 1024, 16, 16, 1024, 256, "split-K (last arriver), the ADM-256 16x16 level"),
    (2, 256, 32, 32, 128, 1024, "split-K (last arriver) on a 32x32 map"),
    (4, 64, 32, 32, 128, 16, "conv_fast<9> epilogue: two partials per 128-pixel tile"),
    (16, 128, 32, 32, 512, 16, "halo kernel, 256 tiles: four partials per 16x16 patch"),
]


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("case", STAT_CASES, ids=lambda c: f"B{c[0]}-Cin{c[1]}-{c[2]}x{c[3]}-Cout{c[4]}")
def test_ride_along_statistics_under_the_production_dispatch(case, t16):
    """The statistics every bf16 GroupNorm of the benchmarked path consumes, from each kernel that emits them under the
    PRODUCTION dispatch (ops.groupnorm trusts the attached buffer blindly): sums over partials == sums over the stored
    output, for every 8-channel chunk; then GroupNorm with and without them."""
    from diffusion_nlc_amd import ops
    B, Cin, H, W, Cout, P_expect, what = case
    g = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.3 + 0.5
    res = torch.randn(B, Cout, H, W, generator=g)
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "auto"
    try:
        y = ops.conv2d(_nhwc(x, t16), ops.pack_conv(w, b, t16, _dev()), res=_nhwc(res, t16))
    finally:
        ops.CONV_POLICY = old
    st = getattr(y, "_nlc_stats", None)
    gran = ops.stats_granule(Cout)                                   # 4 for the 128-channel case (32 groups of 4), else 8
    assert st is not None and st.shape == (B, Cout // gran, 4), (what, None if st is None else st.shape)
    ref = F.conv2d(_rt(x, t16), _rt(w, t16), b, padding=1) + _rt(res, t16)
    _close(y.permute(0, 3, 1, 2), ref, 2e-2, what)
    ch = y.float().cpu().view(B, H * W, Cout // gran, gran).double()
    got = _totals(st)
    s_ref, q_ref = ch.sum(dim=(1, 3)), (ch ** 2).sum(dim=(1, 3))
    assert (got[..., 0] - s_ref).abs().max() <= 3e-3 * max(s_ref.abs().max().item(), 1.0), what
    assert ((got[..., 1] - q_ref) / q_ref).abs().max() <= 2e-3, what
    gamma, beta = torch.randn(Cout, generator=g).to(_dev()), torch.randn(Cout, generator=g).to(_dev())
    fused = ops.groupnorm(y, gamma, beta, groups=32, eps=1e-5, silu=True)
    ops.FUSED_GN_STATS = False
    try:
        plain = ops.groupnorm(y, gamma, beta, groups=32, eps=1e-5, silu=True)
    finally:
        ops.FUSED_GN_STATS = True
    refn = F.silu(F.group_norm(y.float().cpu().permute(0, 3, 1, 2), 32, gamma.cpu(), beta.cpu(), eps=1e-5)).permute(0, 2, 3, 1)
    scale = refn.abs().max().item()
    assert (fused.float().cpu() - plain.float().cpu()).abs().max().item() <= 2e-2 * scale, what
    assert (fused.float().cpu() - refn).abs().max().item() <= 2e-2 * scale, what


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("policy", ["auto", "halo"])
def test_groupnorm_with_four_channel_groups_uses_ride_along_statistics(policy, t16):
    """128 channels in 32 groups = 4 channels per group (cfg 4's two highest-resolution levels, EDM's first layer): the producing
    convolutions emit their statistics per 4 channels (nlc_conv_desc.stats_granule), GroupNorm folds them - alone, pooled
    (groupnorm_pool2x2) and over the concatenation of two such tensors (8-channel groups made of two 4-channel chunks) - and agrees
    with the statistics-pass path and with torch."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(41)
    B, Cin, H, W, C = 2, 64, 32, 64, 128
    x = torch.randn(B, Cin, H, W, generator=g)
    old = ops.CONV_POLICY
    ops.CONV_POLICY = policy
    try:
        ys = []
        for k in range(2):
            w = torch.randn(C, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
            b = torch.randn(C, generator=g) * 0.3 + 0.4 * k
            ys.append(ops.conv2d(_nhwc(x, t16), ops.pack_conv(w, b, t16, _dev())))
    finally:
        ops.CONV_POLICY = old
    for y in ys:
        st = getattr(y, "_nlc_stats", None)
        assert st is not None and st.shape[1] == C // 4, "expected statistics per 4 channels"
        ch = y.float().cpu().view(B, H * W, C // 4, 4).double()
        got = _totals(st)
        assert (got[..., 0] - ch.sum(dim=(1, 3))).abs().max() <= 3e-3 * max(ch.sum(dim=(1, 3)).abs().max().item(), 1.0)
        assert ((got[..., 1] - (ch ** 2).sum(dim=(1, 3))) / (ch ** 2).sum(dim=(1, 3))).abs().max() <= 2e-3
    tol = 2e-2 if t16 == torch.bfloat16 else 3e-3

    def both(fn):
        fused = fn()
        ops.FUSED_GN_STATS = False
        try:
            plain = fn()
        finally:
            ops.FUSED_GN_STATS = True
        return fused, plain

    gamma, beta = torch.randn(2 * C, generator=g).to(_dev()), torch.randn(2 * C, generator=g).to(_dev())
    # one 128-channel tensor: groups of 4
    f1, p1 = both(lambda: ops.groupnorm(ys[0], gamma[:C], beta[:C], groups=32, eps=1e-6, silu=True))
    ref1 = F.silu(F.group_norm(ys[0].float().cpu().permute(0, 3, 1, 2), 32, gamma[:C].cpu(), beta[:C].cpu(), eps=1e-6)).permute(0, 2, 3, 1)
    sc = ref1.abs().max().item()
    assert (f1.float().cpu() - p1.float().cpu()).abs().max().item() <= tol * sc and (f1.float().cpu() - ref1).abs().max().item() <= tol * sc
    # cat of two: 256 channels, groups of 8 = two 4-channel chunks each
    f2, p2 = both(lambda: ops.groupnorm(ys[0], gamma, beta, groups=32, eps=1e-5, silu=False, x1=ys[1]))
    cat = torch.cat([ys[0].float().cpu(), ys[1].float().cpu()], dim=-1).permute(0, 3, 1, 2)
    ref2 = F.group_norm(cat, 32, gamma.cpu(), beta.cpu(), eps=1e-5).permute(0, 2, 3, 1)
    sc = ref2.abs().max().item()
    assert (f2.float().cpu() - p2.float().cpu()).abs().max().item() <= tol * sc and (f2.float().cpu() - ref2).abs().max().item() <= tol * sc
    # both branches of a down-sampling block from one read
    (h3, x3), (h3p, x3p) = both(lambda: ops.groupnorm_pool2x2(ys[0], gamma[:C], beta[:C], groups=32, eps=1e-6, silu=True))
    refh = F.avg_pool2d(ref1.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    sc = refh.abs().max().item()
    assert (h3.float().cpu() - h3p.float().cpu()).abs().max().item() <= tol * sc and (h3.float().cpu() - refh).abs().max().item() <= tol * sc
    assert torch.equal(x3, x3p)


def test_pack_conv_weights_abi_matches_the_host_packing_rule():
    """nlc_pack_conv_weights (device) against the packing rule written out on the host: layout, padding, output-row
    permutation, per-row f64 scale (attention scale / BatchNorm fold), bias fold, input-column permutation."""
    from diffusion_nlc_amd import _ext, ops
    g = torch.Generator().manual_seed(4)
    for dtype in DTYPES:
        Cout, Cin, k = 72, 40, 3
        w = torch.randn(Cout, Cin, k, k, generator=g)
        b = torch.randn(Cout, generator=g)
        rp = torch.randperm(Cout, generator=g)
        cp = torch.randperm(Cin, generator=g)
        rs = torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5
        ba = torch.randn(Cout, generator=g, dtype=torch.float64)
        pw = ops.pack_conv(w, b, dtype, _dev(), row_perm=rp, row_scale=rs, col_perm=cp, bias_add=ba)
        cm, km = _ext.pack_dims(ops.dtype_enum(dtype))
        assert pw.w.shape == ((Cout + cm - 1) // cm * cm, k * k, (Cin + km - 1) // km * km) and pw.w.dtype == dtype
        want = (w[rp][:, cp].double() * rs.view(-1, 1, 1, 1)).float().permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)
        got = pw.w.float().cpu()
        assert torch.equal(got[:Cout, :, :Cin], want.to(dtype).float())
        assert got[Cout:].abs().max() == 0 and got[:, :, Cin:].abs().max() == 0
        assert torch.equal(pw.bias.cpu(), (b[rp].double() * rs + ba).float())
    lin = ops.pack_conv(torch.randn(10, 20, generator=g), None, torch.float32, _dev())
    assert lin.bias is None and lin.KH == 1 and lin.Cin == 20


GNCONV_CASES = [
    dict(B=2, Cin=128, H=32, W=48, Cout=256, split=64, silu=True, res=True),          # concat input, edge + interior patches
    dict(B=1, Cin=192, H=16, W=16, Cout=128, silu=False),                               # one patch per image: every halo side is padding
    dict(B=2, Cin=64, H=16, W=32, Cout=128, silu=True, ups=True),                       # fused nearest-2x upsample of the normalised input
    dict(B=1, Cin=128, H=256, W=256, Cout=256, silu=True, emb=True),                    # 512 tiles: two per persistent workgroup (cross-tile streams)
    dict(B=3, Cin=256, H=32, W=32, Cout=6, silu=True, nchw=True),                       # the network's last layer (GroupNorm -> SiLU -> conv, 6 channels)
]


@pytest.mark.parametrize("case", GNCONV_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d_with_groupnorm_prologue(case):
    """nlc_conv_desc.gn_coef: conv(act(a[b][c] x + b[b][c])) with the per-(image, channel) affine map and SiLU applied to the
    halo rows in LDS - against F.conv2d of the explicitly normalised input (zero padding AFTER the normalisation)."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(77)
    B, Cin, H, W, Cout = (case[k] for k in ("B", "Cin", "H", "W", "Cout"))
    x = torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    a_ = torch.rand(B, Cin, generator=g) + 0.5
    b_ = torch.randn(B, Cin, generator=g) * 0.3
    xr = _rt(x, torch.bfloat16)
    y = xr * a_[:, :, None, None] + b_[:, :, None, None]
    if case.get("silu"):
        y = F.silu(y)
    y = _rt(y, torch.bfloat16)                                       # the prologue stores bf16 back into LDS
    if case.get("ups"):
        y = F.interpolate(y, scale_factor=2, mode="nearest")
    ref = F.conv2d(y, _rt(w, torch.bfloat16), b, padding=1)
    emb = res = None
    if case.get("emb"):
        emb = torch.randn(B, Cout, generator=g)
        ref = ref + emb[:, :, None, None]
    if case.get("res"):
        res = torch.randn(B, Cout, ref.shape[2], ref.shape[3], generator=g)
        ref = ref + _rt(res, torch.bfloat16)
    coef = torch.zeros(B * Cin * 2 + 128)
    coef[:B * Cin * 2] = torch.stack([a_, b_], dim=-1).reshape(-1)
    coef = coef.to(_dev())
    pw = ops.pack_conv(w, b, torch.bfloat16, _dev())
    split = case.get("split")
    x0 = _nhwc(x[:, :split] if split else x, torch.bfloat16)
    x1 = _nhwc(x[:, split:], torch.bfloat16) if split else None
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "halo"
    try:
        kw = dict(x1=x1, upsample2x=bool(case.get("ups")), out_nchw_f32=bool(case.get("nchw")))
        assert ops.conv2d(x0, pw, query_prologue=True, **kw)
        got = ops.conv2d(x0, pw, gn_coef=coef, gn_act=1 if case.get("silu") else 0, emb=None if emb is None else emb.to(_dev()),
                         res=None if res is None else _nhwc(res, torch.bfloat16), **kw)
        torch.cuda.synchronize()
        # a launch that cannot take the prologue must refuse the coefficients instead of ignoring them
        ops.CONV_POLICY = "no_halo"
        assert not ops.conv2d(x0, pw, query_prologue=True, **kw)
        with pytest.raises(Exception):
            ops.conv2d(x0, pw, gn_coef=coef, gn_act=1, **kw)
    finally:
        ops.CONV_POLICY = old
    if not case.get("nchw"):
        got = got.permute(0, 3, 1, 2)
    _close(got, ref, 2e-2, "conv2d + GroupNorm prologue")


def test_groupnorm_coef_then_conv_equals_groupnorm_then_conv():
    """The network-level contract: Norm.then_conv (coefficients from the ride-along statistics + prologue) against the separate
    GroupNorm pass followed by the plain convolution - same statistics, same arithmetic, so equal to bf16 rounding of the
    normalised activation (one ulp where the two paths round a*x+b differently: fmaf vs mul+add)."""
    from diffusion_nlc_amd import ops
    from diffusion_nlc_amd.hipnet import Norm
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 128, 32, 32
    x = torch.randn(B, 64, H, W, generator=g)
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "halo"
    try:
        w0 = torch.randn(C, 64, 3, 3, generator=g) / 24
        h0 = ops.conv2d(_nhwc(x, torch.bfloat16), ops.pack_conv(w0, torch.zeros(C), torch.bfloat16, _dev()))      # carries statistics
        h1 = ops.conv2d(_nhwc(x, torch.bfloat16), ops.pack_conv(w0.flip(0), torch.ones(C) * 0.3, torch.bfloat16, _dev()))
        sd = {"n.weight": 1 + 0.2 * torch.randn(2 * C, generator=g), "n.bias": 0.1 * torch.randn(2 * C, generator=g)}
        norm = Norm(sd, "n", _dev(), 32, 1e-5)
        pw = ops.pack_conv(torch.randn(C, 2 * C, 3, 3, generator=g) / 48, torch.zeros(C), torch.bfloat16, _dev())
        ss = (torch.randn(B, 4 * C, generator=g) * 0.3).to(_dev())
        kw = dict(silu=True, x1=h1, scale=ss[:, :2 * C], shift=ss[:, 2 * C:])
        was = ops.FUSE_GN_CONV
        try:
            ops.FUSE_GN_CONV = True
            fused = norm.then_conv(h0, pw, **kw)
            ops.FUSE_GN_CONV = False
            plain = norm.then_conv(h0, pw, **kw)
        finally:
            ops.FUSE_GN_CONV = was
    finally:
        ops.CONV_POLICY = old
    scale = plain.float().abs().max().item()
    err = (fused.float() - plain.float()).abs().max().item()
    assert getattr(fused, "_nlc_stats", None) is not None and err <= 1e-2 * scale, (err, scale)


HALOSPLIT_CASES = [
    dict(B=16, Cin=512, H=16, W=16, Cout=1024, ks=2, res=True, emb=True),                 # 128 tiles x 2: the ADM-256 16x16 level at B = 16
    dict(B=16, Cin=1024, H=16, W=16, Cout=512, ks=4, split=512, res=True),                # 64 tiles x 4; the concat boundary coincides with a split boundary
    dict(B=4, Cin=768, H=32, W=32, Cout=512, ks=3, split=256, scale=math.sqrt(0.5)),      # 64 tiles x 3 (12 channel blocks), a split starts at the second segment
    dict(B=16, Cin=512, H=16, W=16, Cout=1024, ks=2, split=64),                           # the concat boundary lies INSIDE the first split
    dict(B=16, Cin=512, H=16, W=16, Cout=1024, ks=2, ups_from=(8, 8), res=True),          # fused nearest-2x upsample
    dict(B=8, Cin=256, H=16, W=32, Cout=256, ks=2, ups_from=(8, 16)),                     # fused nearest-2x upsample, 32 tiles... x 2 = 64 < 128: NOT split
]


@pytest.mark.parametrize("case", HALOSPLIT_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
def test_conv2d_halo_kernel_split_k(case, t16):
    """conv_halo_kernel<bf16, false, SPLIT> under the production dispatch: launches with fewer tiles than CUs; raw f32 partial sums per
    channel-block range, reduced in split order by whichever workgroup arrives last at the tile, which then runs the normal epilogue
    (bias / embedding / residual / statistics).  Deterministic: two runs are bit-identical."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(47)
    B, Cin, H, W, Cout = (case[k] for k in ("B", "Cin", "H", "W", "Cout"))
    ups = "ups_from" in case
    if ups:
        H, W = case["ups_from"]
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    xr = _rt(x, t16)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr
    ref = F.conv2d(xin, _rt(w, t16), b, padding=1)
    emb = res = None
    if case.get("emb"):
        emb = torch.randn(B, Cout, generator=g)
        ref = ref + emb[:, :, None, None]
    if case.get("res"):
        res = torch.randn(B, Cout, ref.shape[2], ref.shape[3], generator=g)
        ref = ref + _rt(res, t16)
    ref = ref * case.get("scale", 1.0)
    pw = ops.pack_conv(w, b, t16, _dev())
    split = case.get("split")
    x0 = _nhwc(x[:, :split] if split else x, t16)
    x1 = _nhwc(x[:, split:], t16) if split else None
    kw = dict(x1=x1, upsample2x=ups, emb=None if emb is None else emb.to(_dev()), res=None if res is None else _nhwc(res, t16),
              out_scale=case.get("scale", 1.0))
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "auto"
    try:
        got = ops.conv2d(x0, pw, **kw)
        again = ops.conv2d(x0, pw, **kw)
        assert torch.equal(got, again)               # fixed summation order whoever arrives last
        ops.CONV_TUNING = 128                        # the same launch without the halo kernel's split-K
        plain = ops.conv2d(x0, pw, **kw)
        torch.cuda.synchronize()
    finally:
        ops.CONV_POLICY = old
        ops.CONV_TUNING = 0
    Ho, Wo = ref.shape[2], ref.shape[3]
    st = getattr(got, "_nlc_stats", None)
    tiles = B * (Ho // 16) * (Wo // 16) * (Cout // 128)
    gran = ops.stats_granule(Cout)
    if tiles * case["ks"] >= 128:
        assert st is not None and st.shape == (B, Cout // gran, 4), "the halo kernel did not take this launch"
    _close(got.permute(0, 3, 1, 2), ref, 2e-2, "conv2d (halo split-K)")
    assert (got.float() - plain.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    if st is not None:
        ch = got.float().cpu().view(B, Ho * Wo, Cout // gran, gran).double()
        tot = _totals(st)
        s_ref, q_ref = ch.sum(dim=(1, 3)), (ch ** 2).sum(dim=(1, 3))
        assert (tot[..., 0] - s_ref).abs().max() <= 3e-3 * max(s_ref.abs().max().item(), 1.0)
        assert ((tot[..., 1] - q_ref) / q_ref).abs().max() <= 2e-3


def test_split_k_launches_leave_the_arrival_counters_zero():
    """nlc_conv_desc.workspace: the first 4096 bytes are the split-K arrival counters (one per tile, self-resetting); a sequence of
    split launches of different shapes and kernels through ONE workspace leaves them zero and every result right."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(53)
    shapes = [(16, 512, 16, 16, 1024, 3), (16, 1024, 8, 8, 1024, 3), (16, 1024, 8, 8, 3072, 1), (16, 1024, 16, 16, 512, 3), (16, 512, 16, 16, 1024, 3)]
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "auto"
    try:
        for B, Cin, H, W, Cout, k in shapes:
            x = torch.randn(B, Cin, H, W, generator=g)
            w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
            got = ops.conv2d(_nhwc(x, torch.bfloat16), ops.pack_conv(w, None, torch.bfloat16, _dev()))
            ref = F.conv2d(_rt(x, torch.bfloat16), _rt(w, torch.bfloat16), None, padding=k // 2)
            _close(got.permute(0, 3, 1, 2), ref, 2e-2, f"split-K sequence {Cin}->{Cout} @{H}")
            torch.cuda.synchronize()
            ws = [t for t in ops._conv_ws.values() if t.device == got.device]
            assert ws and all(int(t[:1024].view(torch.int32).abs().max().item()) == 0 for t in ws), "arrival counters not left zero"
    finally:
        ops.CONV_POLICY = old


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("with_emb", [False, True], ids=["bias", "bias+emb"])
@pytest.mark.parametrize("with_res", [False, True], ids=["nores", "res"])
def test_halo_conv_aligned_bias_instantiation_matches_the_general_one(dtype, with_emb, with_res):
    """conv_halo_kernel's CF instantiation (whole N-tiles, >= 2 channel blocks, 16-byte-aligned bias / embedding: explicit loads of
    the next tile's initial accumulator values, a tile's rows stored one per k-step under the NEXT tile's first channel block -
    which runs a copy of the k-loop body of its own -, counted waits that allow for those stores, the last tile's rows flushed
    behind the loop) must be bit-identical to the general instantiation, which the same call takes when the bias pointer is not
    16-byte aligned.  768 tiles on 256 persistent workgroups: every workgroup crosses two tile boundaries (rows pending across
    both), onto a new patch and onto the other N-tile of the same patch."""
    import dataclasses
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(_seed(("cf", str(dtype), with_emb, with_res)))
    B, Cin, H, Cout = 6, 128, 128, 256
    x = _nhwc(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    pw = ops.pack_conv(w, b, dtype, _dev())
    assert pw.bias.data_ptr() % 16 == 0
    shifted = torch.zeros(Cout + 4, device=_dev(), dtype=torch.float32)
    shifted[1:Cout + 1] = pw.bias
    pw_unaligned = dataclasses.replace(pw, bias=shifted[1:Cout + 1])
    assert pw_unaligned.bias.data_ptr() % 16 == 4
    emb = torch.randn(B, Cout, generator=g).to(_dev()) if with_emb else None
    res = _nhwc(torch.randn(B, Cout, H, H, generator=g), dtype) if with_res else None
    ya, yb = (ops.conv2d(x, q, emb=emb, res=res) for q in (pw, pw_unaligned))
    sa, sb = getattr(ya, "_nlc_stats", None), getattr(yb, "_nlc_stats", None)
    assert sa is not None and sb is not None, "both launches emit ride-along statistics"
    assert torch.equal(ya, yb)
    assert torch.equal(sa, sb)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), _rt(w, dtype).to(_dev()), b.to(_dev()), padding=1)
    if with_emb:
        ref = ref + emb[:, :, None, None]
    if with_res:
        ref = ref + res.float().permute(0, 3, 1, 2)
    _close(ya.permute(0, 3, 1, 2), ref.cpu(), _tol(dtype), "CF halo conv")


@pytest.mark.parametrize("B", [1, 2, 3])
@pytest.mark.parametrize("cat", [False, True], ids=["one-input", "cat-256+256"])
def test_halo_conv_next_tile_starts_from_landed_bias(B, cat):
    """Regression (round 5): the CF halo kernel fetches the NEXT tile's bias (+ embedding) vector at the top of a tile's epilogue and
    initialises the next tile's accumulators from it at the end.  While those were inline-asm loads the register allocator copied
    their destination registers in front of the inline-asm wait, so a tile started from whatever the registers held whenever the
    bias vector missed in L2 - seen on 256 x 256 maps at B <= 3 (2 ... 6 tiles per workgroup, short epilogues: deferred rows, no
    residual), a whole XCD's round of tiles at a time, only with ride-along statistics on.  Every launch here gets a bias vector
    of its own at a fresh address behind a cache-sweeping copy; compared with the other 3x3 kernel on the same operands
    (tools/asm_load_audit.py checks the compiled code for the pattern itself; tests/test_host_cpu.py runs it)."""
    from diffusion_nlc_amd import ops
    import dataclasses
    g = torch.Generator().manual_seed(_seed(("cold-bias", B, cat)))
    H, Cout = 256, 256
    c0, c1 = (256, 256) if cat else (256, 0)
    x0 = _nhwc(torch.randn(B, c0, H, H, generator=g), torch.bfloat16)
    x1 = _nhwc(torch.randn(B, c1, H, H, generator=g), torch.bfloat16) if c1 else None
    w = torch.randn(Cout, c0 + c1, 3, 3, generator=g) / math.sqrt((c0 + c1) * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    pw = ops.pack_conv(w, b, torch.bfloat16, _dev())
    old = ops.CONV_POLICY
    try:
        ops.CONV_POLICY = "no_halo"
        ref = ops.conv2d(x0, pw, x1=x1, emit_stats=False).float()
        ops.CONV_POLICY = "auto"
        sweep = torch.empty(96 << 20, device=_dev(), dtype=torch.uint8)
        keep = []
        for rep in range(6):
            fresh = torch.empty(Cout + 64 * (rep + 1), device=_dev(), dtype=torch.float32)[:Cout]      # a new allocation each time
            fresh.copy_(pw.bias)
            keep.append(fresh)
            sweep.fill_(rep)                                                                          # 96 MB through every L2
            got = ops.conv2d(x0, dataclasses.replace(pw, bias=fresh), x1=x1, emit_stats=True).float()
            d = (got - ref).abs()
            assert int((d > 0.1).sum().item()) == 0, f"launch {rep}: {int((d > 0.1).sum().item())} outputs off by more than 0.1 (max {d.max().item():.3g})"
    finally:
        ops.CONV_POLICY = old


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("ups", [False, True], ids=["plain", "ups2x"])
@pytest.mark.parametrize("split", [None, (128, 128), (192, 64)], ids=["one-input", "cat-128+128", "cat-192+64"])
@pytest.mark.parametrize("shape", [(6, 128, 128), (3, 64, 256)], ids=["6x128x128", "3x64x256"])
def test_halo_conv_tile_and_segment_changes(dtype, ups, split, shape):
    """A persistent halo workgroup re-targets its halo source addresses at every change of patch and, on a concatenated input, of
    input segment: by one uniform distance where the new patch touches the same image borders (and the two segments have the same
    channel count), by the full per-row computation otherwise.  Several tiles per workgroup (768 / 384 on 256), every kind of
    change among them - down inside an image, onto and off the top / bottom (and, on the wide map, left / right) borders, on to
    the next image, segment 0 -> 1 -> 0 with equal and unequal channel counts, with and without the fused nearest-2x upsample -
    against torch's convolution of the same (concatenated, upsampled) input."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(_seed(("tile-changes", str(dtype), ups, split, shape)))
    B, H, W = shape
    Cin, Cout = 256, 256
    hi, wi = (H // 2, W // 2) if ups else (H, W)
    xf = torch.randn(B, Cin, hi, wi, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    pw = ops.pack_conv(w, b, dtype, _dev())
    if split is None:
        x0, x1 = _nhwc(xf, dtype), None
    else:
        x0, x1 = _nhwc(xf[:, :split[0]], dtype), _nhwc(xf[:, split[0]:], dtype)
    got = ops.conv2d(x0, pw, x1=x1, upsample2x=ups)
    xin = _rt(xf, dtype)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin.to(_dev()), _rt(w, dtype).to(_dev()), b.to(_dev()), padding=1)
    _close(got.permute(0, 3, 1, 2), ref.cpu(), _tol(dtype), "halo conv across tile / segment changes")


def test_split_k_stress():
    """2 000 back-to-back split-K launches, mixed kernels and levels (halo kernel on the 16x16 level, conv_fast<9> on the 8x8 level,
    conv_fast<1>), through ONE workspace, half of them beside a second stream that keeps the memory system busy: every result is
    bit-identical to the first one of its shape, which is bit-identical to the formally fenced variant (tuning bit 10: agent-scope
    release before the arrival add, acquire in the last arriver) and agrees with the un-split kernel (tuning bit 11) to rounding.
    The hand-off rests on measured sc1 write-through / L1-bypass behaviour (conv_halo.hip), not on the HIP memory model: a
    visibility failure would show up here as a mismatch, not a crash."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(77)
    shapes = [(16, 1024, 16, 16, 1024, 3), (16, 2048, 16, 16, 1024, 3), (16, 1024, 8, 8, 1024, 3), (16, 2048, 8, 8, 1024, 3),
              (16, 1024, 8, 8, 3072, 1), (16, 512, 16, 16, 1024, 3)]
    old_p, old_t, old_d = ops.CONV_POLICY, ops.CONV_TUNING, ops.CONV_DEBUG
    ops.CONV_POLICY = "auto"
    try:
        cases = []
        for B, Cin, H, W, Cout, k in shapes:
            x = _nhwc(torch.randn(B, Cin, H, W, generator=g), torch.bfloat16)
            pw = ops.pack_conv(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), torch.randn(Cout, generator=g), torch.bfloat16, _dev())
            ops.CONV_TUNING = 0
            first = ops.conv2d(x, pw, emit_stats=False)
            ops.CONV_TUNING = 1024
            fenced = ops.conv2d(x, pw, emit_stats=False)
            ops.CONV_TUNING = 2048
            unsplit = ops.conv2d(x, pw, emit_stats=False)
            ops.CONV_TUNING = 0
            torch.cuda.synchronize()
            assert torch.equal(first, fenced), f"{Cin}->{Cout}@{H} k{k}: unfenced hand-off differs from the release/acquire variant"
            err = (first.float() - unsplit.float()).abs().max().item() / unsplit.float().abs().max().item()
            assert err <= 1.6e-2, f"{Cin}->{Cout}@{H} k{k}: split vs un-split {err:.3e}"          # one bf16 ulp of the output + order
            cases.append((x, pw, first))
        ops.CONV_DEBUG = 1                           # every split launch first verifies that the arrival counters are zero
        ops.conv2d(cases[0][0], cases[0][1], emit_stats=False)
        ops.CONV_DEBUG = 0
        side = torch.cuda.Stream()
        hog_a = torch.empty(128 << 20, device=_dev(), dtype=torch.float32)
        hog_b = torch.empty_like(hog_a)
        bad = torch.zeros((), device=_dev(), dtype=torch.int32)
        n = 0
        for rnd in range(334):
            if rnd % 2 == 0:
                with torch.cuda.stream(side):        # uneven load: 1 GiB of copies in flight beside the next six launches
                    hog_b.copy_(hog_a)
            for x, pw, first in cases:
                got = ops.conv2d(x, pw, emit_stats=False)
                bad += (got != first).any().to(torch.int32)
                n += 1
        torch.cuda.synchronize()
        assert n >= 2000 and int(bad.item()) == 0, f"{int(bad.item())} of {n} split-K launches differ from the first result of their shape"
        ws = [t for t in ops._conv_ws.values() if t.device == _dev()]
        assert ws and all(int(t[:1024].view(torch.int32).abs().max().item()) == 0 for t in ws), "arrival counters not left zero"
    finally:
        ops.CONV_POLICY, ops.CONV_TUNING, ops.CONV_DEBUG = old_p, old_t, old_d


def test_poisoned_split_k_workspace_is_detected_and_replaced():
    """nlc_conv_desc.debug bit 0: a non-zero arrival counter on entry (an aborted launch, a foreign writer) makes nlc_conv2d return
    NLC_EINVAL instead of silently reducing early / never; ops.conv2d then drops the workspace, so the next call gets a zeroed one."""
    from diffusion_nlc_amd import _ext, ops
    g = torch.Generator().manual_seed(78)
    x = _nhwc(torch.randn(16, 1024, 8, 8, generator=g), torch.bfloat16)
    pw = ops.pack_conv(torch.randn(1024, 1024, 3, 3, generator=g) / 96, None, torch.bfloat16, _dev())
    old_p, old_d = ops.CONV_POLICY, ops.CONV_DEBUG
    ops.CONV_POLICY = "auto"
    try:
        good = ops.conv2d(x, pw)
        torch.cuda.synchronize()
        ws = [t for t in ops._conv_ws.values() if t.device == _dev()]
        assert ws
        for t in ws:
            t[:1024].view(torch.int32)[5] = 1      # poison counter 5
        ops.CONV_DEBUG = 1
        with pytest.raises(_ext.NlcError, match="arrival counter 5"):
            ops.conv2d(x, pw)
        assert not ops._conv_ws, "the poisoned workspace must be dropped"
        again = ops.conv2d(x, pw)                   # fresh zeroed workspace
        torch.cuda.synchronize()
        assert torch.equal(again, good)
    finally:
        ops.CONV_POLICY, ops.CONV_DEBUG = old_p, old_d


@pytest.mark.parametrize("cout,nchw,B,H,W,Cin,scale", [(6, True, 2, 64, 128, 128, 1.0), (3, False, 4, 64, 64, 64, 0.5), (16, False, 16, 32, 32, 192, 1.0),
                                                       (1, True, 16, 32, 32, 64, 1.0), (6, True, 16, 256, 256, 256, 1.0)])    # >= 64 patches each
@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
def test_conv2d_narrow_output_kernel(cout, nchw, B, H, W, Cin, scale, t16):
    """conv_narrow_kernel: 3x3 with <= 16 output channels under the production dispatch (the networks' last layer, 256 -> 6 / 128 -> 3):
    whole halo + all nine taps' weights per channel block in LDS, one barrier per block."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(59 + cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(_rt(x, t16), _rt(w, t16), b, padding=1) * scale
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "auto"
    try:
        pw = ops.pack_conv(w, b, t16, _dev())
        got = ops.conv2d(_nhwc(x, t16), pw, out_nchw_f32=nchw, out_scale=scale)
        ops.CONV_TUNING = 512                        # the same launch through the kernels that took it before
        other = ops.conv2d(_nhwc(x, t16), pw, out_nchw_f32=nchw, out_scale=scale)
        torch.cuda.synchronize()
    finally:
        ops.CONV_POLICY = old
        ops.CONV_TUNING = 0
    _close(got if nchw else got.permute(0, 3, 1, 2), ref, 1e-2 if nchw else 2e-2, f"conv2d narrow Cout={cout}")
    assert (got.float() - other.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_new_entry_points_reject_bad_arguments():
    """Error behaviour of the entry points added this round: loud, with a message, nothing launched."""
    from diffusion_nlc_amd import _ext, ops
    x = torch.randn(1, 6, 6, 64, device=_dev()).to(torch.bfloat16)
    w = ops.pack_conv(torch.randn(64, 64, 3, 3), None, torch.bfloat16, _dev())
    with pytest.raises(ValueError):                                  # residual of the wrong size for res_upsample2x
        ops.conv2d(x, w, res=torch.zeros(1, 6, 6, 64, device=_dev(), dtype=torch.bfloat16), res_upsample2x=True)
    with pytest.raises(ValueError):                                  # res_upsample2x without a residual
        ops.conv2d(x, w, res_upsample2x=True)
    x7 = torch.randn(1, 7, 8, 64, device=_dev()).to(torch.bfloat16)
    g = torch.ones(64, device=_dev())
    with pytest.raises(_ext.NlcError):                               # odd height
        ops.groupnorm_pool2x2(x7, g, g, groups=32, eps=1e-5, silu=True)
    with pytest.raises(_ext.NlcError):                               # unknown conv policy value through the raw descriptor
        d = _ext.ConvDesc(policy=99)
        _ext.check(_ext.load().nlc_conv2d(_ext.C.byref(d), 1, None), "nlc_conv2d")


RESUPS_CASES = [
    # (B, Cin, H, W, Cout, policy, dtype): every kernel family that reads a residual
    (2, 64, 32, 32, 128, "halo", torch.bfloat16),        # halo kernel, hot epilogue
    (1, 64, 16, 48, 72, "halo", torch.bfloat16),         # halo kernel, general epilogue (ragged Cout)
    (2, 64, 16, 16, 128, "no_halo", torch.bfloat16),     # conv_fast<9>, hot epilogue
    (16, 128, 8, 8, 128, "auto", torch.bfloat16),        # conv_fast<9> + split-K (last-arriver reduction)
    (2, 40, 12, 20, 24, "generic", torch.bfloat16),      # generic implicit GEMM
    (2, 32, 16, 16, 64, "auto", torch.float32),          # f32 parity path
]


@pytest.mark.parametrize("case", RESUPS_CASES, ids=lambda c: f"B{c[0]}-Cin{c[1]}-{c[2]}x{c[3]}-Cout{c[4]}-{c[5]}-{'bf16' if c[6] == torch.bfloat16 else 'f32'}")
def test_conv2d_residual_read_nearest_2x_upsampled(case):
    """nlc_conv_desc.res_upsample2x: out = conv(x) + upsample2x(res_small) - the skip branch of an up-sampling ResBlock
    (src/unet_adm.py:186-190) read in place by every epilogue that takes a residual."""
    from diffusion_nlc_amd import ops
    B, Cin, H, W, Cout, policy, dtype = case
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    rs = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    ref = F.conv2d(_rt(x, dtype), _rt(w, dtype), b, padding=1) + F.interpolate(_rt(rs, dtype), scale_factor=2, mode="nearest")
    old = ops.CONV_POLICY
    ops.CONV_POLICY = policy
    try:
        got = ops.conv2d(_nhwc(x, dtype), ops.pack_conv(w, b, dtype, _dev()), res=_nhwc(rs, dtype), res_upsample2x=True)
        same = ops.conv2d(_nhwc(x, dtype), ops.pack_conv(w, b, dtype, _dev()), res=ops.upsample2x(_nhwc(rs, dtype)))
        torch.cuda.synchronize()
    finally:
        ops.CONV_POLICY = old
    _close(got.permute(0, 3, 1, 2), ref, _tol(dtype), f"conv2d res_upsample2x ({policy})")
    assert torch.equal(got, same)                 # identical arithmetic to the materialised upsample


ADM_B1_SHAPES = [
    # (Cin, H, Cout): the 3x3 shapes of ADM-256 at B = 1 that BOTH kernels can take (the 256x256 ones scaled to 128x128)
    (256, 128, 256), (512, 64, 256), (512, 64, 512), (1024, 32, 512), (1024, 16, 1024), (768, 32, 512),
]


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("shape", ADM_B1_SHAPES, ids=lambda s: f"{s[0]}to{s[2]}at{s[1]}")
def test_halo_and_fast_kernels_agree_to_f32_rounding(shape, t16):
    """The same 16-bit-MFMA / f32-accumulate arithmetic in two kernels (LDS-halo forced vs forbidden) on the ADM-256 shapes at B = 1:
    the outputs are the same f32 sums rounded once to the storage type, up to the order of the f32 additions - at most one rounding
    step of the storage type (plus the f32 sums' own few-ulp order dependence) apart on every element and identical on nearly all
    of them, and the ride-along GroupNorm totals agree to f32 rounding.  (A 10x gap between the two dispatches in a trajectory-level statistic is one draw of the corrected sigma, not a kernel
    difference: DESIGN.md §2.)"""
    from diffusion_nlc_amd import ops
    Cin, H, Cout = shape
    g = torch.Generator().manual_seed(_seed(shape))
    x = _nhwc(torch.randn(1, Cin, H, H, generator=g), t16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    pw = ops.pack_conv(w, b, t16, _dev())
    old = ops.CONV_POLICY
    outs = {}
    try:
        for pol in ("halo", "no_halo"):
            ops.CONV_POLICY = pol
            y = ops.conv2d(x, pw)
            outs[pol] = (y.float().cpu(), _totals(ops.ride_stats(y)))
    finally:
        ops.CONV_POLICY = old
    (a, sa), (c, sc) = outs["halo"], outs["no_halo"]
    ulp = 2.0 ** (-7 if t16 == torch.bfloat16 else -10)           # spacing of the storage type relative to the value: at most 2^-7 / 2^-10
    d = (a - c).abs()
    # one rounding step of the storage type + the f32 sums' own order-dependence (K = 9 Cin terms: a few f32 ulps of the LARGEST
    # outputs, which for outputs near zero is many ulps of the small value itself)
    tol = ulp * torch.maximum(a.abs(), c.abs()) * 1.01 + 4e-6 * a.abs().max()
    assert (d <= tol).all(), f"max {(d / tol).max().item():.2f} x the tolerance"
    assert (d > 0).float().mean().item() < 0.05                      # different summation order flips a rounding on a few per cent
    scale = sa[..., 1].abs().max()
    assert (sa[..., 0] - sc[..., 0]).abs().max() <= 2e-3 * sa[..., 0].abs().max().clamp_min(1.0)
    assert (sa[..., 1] - sc[..., 1]).abs().max() <= 2e-3 * scale


PW_CASES = [
    # persistent pointwise kernel (conv_pw.hip): 1x1 launches with >= 768 (128 px x 128 ch) tiles
    dict(B=4, C0=128, C1=64, H=128, W=128, Cout=512, res=True, scale=0.5 ** 0.5),     # tiles inside one image: one statistics atomic per wave
    dict(B=3, C0=64, C1=0, H=148, W=148, Cout=512, res=False, scale=1.0),            # partial last pixel tile, tiles straddle images
    dict(B=16, C0=256, C1=0, H=64, W=64, Cout=512, res=True, scale=1.0, bias=False),
    dict(B=4, C0=128, C1=128, H=256, W=256, Cout=128, res=False, scale=1.0),         # one N-tile, 4 k-steps
    # ... of which the register-resident-weights form (conv_pwr_kernel) takes K <= 512 in whole 128-channel blocks per segment:
    dict(B=2, C0=256, C1=256, H=256, W=256, Cout=256, res=False, scale=1.0),         # K = 512 over both segments, two channel tiles side by side
    dict(B=16, C0=384, C1=0, H=128, W=128, Cout=128, res=True, scale=1.0),           # K = 384
    dict(B=3, C0=128, C1=0, H=148, W=148, Cout=512, res=False, scale=0.5 ** 0.5),    # partial last 64-pixel tile, tiles straddle images
    dict(B=8, C0=128, C1=0, H=256, W=256, Cout=128, res=False, scale=1.0),           # K = 128: cfg 4's shortcut
    dict(B=16, C0=512, C1=0, H=64, W=64, Cout=256, res=False, scale=1.0),            # 1 024 tiles: streaming form, side-by-side channel tiles
]


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d_pointwise_persistent_kernel(case, t16):
    """conv_pw_kernel (ring of three LDS stages across tile boundaries, counted waits behind the epilogue's stores and atomics):
    against the f32 reference on the rounded operands, against conv_fast on the same launch (tuning bit 15 = never take the pointwise
    kernel) to one storage rounding, ride-along statistics against the stored values - and a second launch bit-identical to the
    first (a counted wait that under-waits would read a stage before its DMA landed)."""
    from diffusion_nlc_amd import ops
    B, C0, C1, H, W, Cout = (case[k] for k in ("B", "C0", "C1", "H", "W", "Cout"))
    g = torch.Generator().manual_seed(_seed(case))
    Cin = C0 + C1
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g) * 0.2 if case.get("bias", True) else None
    res = torch.randn(B, Cout, H, W, generator=g) if case["res"] else None
    ref = F.conv2d(_rt(x, t16), _rt(w, t16), b)
    if res is not None:
        ref = ref + _rt(res, t16)
    ref = ref * case["scale"]
    xs = _nhwc(x, t16)
    x0, x1 = (xs[..., :C0].contiguous(), xs[..., C0:].contiguous()) if C1 else (xs, None)
    pw = ops.pack_conv(w, b, t16, _dev())
    rs = None if res is None else _nhwc(res, t16)
    tiles = -(-B * H * W // 128) * (Cout // 128)
    assert tiles >= 768, "the case must reach the pointwise kernel"
    outs = []
    for tuning in (0, 0, 32768, 65536):
        ops.CONV_TUNING = tuning
        try:
            y = ops.conv2d(x0, pw, x1=x1, res=rs, out_scale=case["scale"])
        finally:
            ops.CONV_TUNING = 0
        outs.append((y, ops.ride_stats(y)))
    torch.cuda.synchronize()
    (ya, sa), (yb, sb), (yc, sc_), (ye, se) = outs                         # tuning bit 16: the streaming form where the resident one would run
    assert torch.equal(ya, yb) and torch.equal(sa, sb)                      # deterministic, launch after launch
    for _ in range(16):
        yn = ops.conv2d(x0, pw, x1=x1, res=rs, out_scale=case["scale"])
        assert torch.equal(yn, ya) and torch.equal(ops.ride_stats(yn), sa)
    _close(ya.permute(0, 3, 1, 2), ref, _tol(t16), "conv2d (pointwise kernel)")
    ulp = 2.0 ** (-7 if t16 == torch.bfloat16 else -10)
    a, c = ya.float().cpu(), yc.float().cpu()
    d = (a - c).abs()
    assert (d <= ulp * torch.maximum(a.abs(), c.abs()) * 1.01 + 4e-6 * a.abs().max()).all()      # vs conv_fast: one rounding step
    e = ye.float().cpu()
    assert ((a - e).abs() <= ulp * torch.maximum(a.abs(), e.abs()) * 1.01 + 4e-6 * a.abs().max()).all()
    assert (_totals(se) - _totals(sa)).abs().max() <= 1e-3 * max(_totals(sa).abs().max().item(), 1.0)
    gran = ops.stats_granule(Cout)
    assert sa is not None and sa.shape == (B, Cout // gran, 4)
    ch = a.view(B, H * W, Cout // gran, gran).double()
    tot = _totals(sa)
    s_ref, q_ref = ch.sum(dim=(1, 3)), (ch ** 2).sum(dim=(1, 3))
    assert (tot[..., 0] - s_ref).abs().max() <= 3e-3 * max(s_ref.abs().max().item(), 1.0)
    assert ((tot[..., 1] - q_ref) / q_ref).abs().max() <= 2e-3


NORM_OUT_CASES = [
    dict(B=2, C0=256, C1=256, H=256, W=256, Cout=256),       # ADM-256's 256x256 output blocks: K = 512 over both segments, two channel tiles
    dict(B=4, C0=128, C1=128, H=256, W=256, Cout=128),       # one channel tile: the workgroup normalises every stage itself
    dict(B=8, C0=384, C1=0, H=128, W=128, Cout=512),         # K = 384 from one segment, four channel tiles (stage kb by workgroup kb % 4)
]


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("case", NORM_OUT_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_conv2d_pointwise_with_normalised_side_output(case, t16):
    """nlc_conv_desc.norm_out (conv_pwr_kernel<.., NORM>): the skip projection of a ResBlock and act(GroupNorm(x)) of the same input from
    one read.  The convolution output must equal the plain launch bit for bit (same kernel arithmetic, x is used unnormalised); the
    normalised output must equal the separate GroupNorm pass to one rounding of a*x+b, and the f32 GroupNorm of the rounded input."""
    from diffusion_nlc_amd import ops
    B, C0, C1, H, W, Cout = (case[k] for k in ("B", "C0", "C1", "H", "W", "Cout"))
    g = torch.Generator().manual_seed(_seed(case))
    C = C0 + C1
    src = _nhwc(torch.randn(B, 64, H, W, generator=g), t16)
    def producer(cout, bias):      # a 3x3 convolution whose epilogue leaves the GroupNorm statistics of its output
        w = torch.randn(cout, 64, 3, 3, generator=g) / 24
        return ops.conv2d(src, ops.pack_conv(w, torch.full((cout,), bias), t16, _dev()))
    h0 = producer(C0, 0.3)
    h1 = producer(C1, -0.2) if C1 else None
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(_dev())
    beta = (0.1 * torch.randn(C, generator=g)).to(_dev())
    wsk = torch.randn(Cout, C, 1, 1, generator=g) / math.sqrt(C)
    pw = ops.pack_conv(wsk, torch.randn(Cout, generator=g) * 0.2, t16, _dev())
    plain_skip = ops.conv2d(h0, pw, x1=h1)
    plain_hn = ops.groupnorm(h0, gamma, beta, groups=32, eps=1e-5, silu=True, x1=h1)
    assert ops.conv2d(h0, pw, x1=h1, query_norm_out=True)
    coef = ops.groupnorm_coef(h0, gamma, beta, groups=32, eps=1e-5, x1=h1)
    assert coef is not None
    outs = [ops.conv2d(h0, pw, x1=h1, gn_coef=coef, gn_act=1, norm_out=True) for _ in range(2)]
    torch.cuda.synchronize()
    (skip, hn), (skip2, hn2) = outs
    assert torch.equal(skip, skip2) and torch.equal(hn, hn2)                      # deterministic (counted waits behind two store streams)
    for _ in range(24):                                                           # ... launch after launch, other launches in between
        ops.groupnorm(h0, gamma, beta, groups=32, eps=1e-5, silu=True, x1=h1)
        s_n, h_n = ops.conv2d(h0, pw, x1=h1, gn_coef=coef, gn_act=1, norm_out=True)
        assert torch.equal(s_n, skip) and torch.equal(h_n, hn)
    assert torch.equal(skip, plain_skip)
    assert ops.ride_stats(skip) is not None and torch.equal(ops.ride_stats(skip), ops.ride_stats(plain_skip))
    a, b_ = hn.float().cpu(), plain_hn.float().cpu()
    ulp = 2.0 ** (-7 if t16 == torch.bfloat16 else -10)
    assert ((a - b_).abs() <= 2 * ulp * torch.maximum(a.abs(), b_.abs()) + 1e-3).all()
    xcat = torch.cat([h0.float().cpu()] + ([h1.float().cpu()] if C1 else []), dim=3).permute(0, 3, 1, 2)
    ref = F.silu(F.group_norm(xcat, 32, gamma.cpu(), beta.cpu(), eps=1e-5))
    _close(hn.permute(0, 3, 1, 2), ref, _tol(t16) * 1.5, "normalised side output")
    # a launch the resident pointwise kernel does not take must say so
    small = _nhwc(torch.randn(1, 64, 32, 32, generator=g), t16)
    assert not ops.conv2d(small, ops.pack_conv(torch.randn(128, 64, 1, 1, generator=g), None, t16, _dev()), query_norm_out=True)


SMALL_CASES = [
    # 3x3 / stride 1 / pad 1 on small maps: what conv_small.hip takes (whole 128-pixel x 128-channel tiles, 64-channel blocks)
    dict(B=4, C0=128, H=8, W=8, Cout=128, groups=16),                                   # one block per slice, two splits, 2 images per tile
    dict(B=2, C0=256, H=16, W=16, Cout=256, groups=32, emb=True, res=True),               # half an image per tile, FiLM
    dict(B=1, C0=128, C1=256, H=32, W=32, Cout=128, groups=8, film=True),                 # concatenated input, 48-channel groups straddling the sources
    dict(B=2, C0=256, C1=128, H=8, W=16, Cout=128, groups=8, res=True, act=True),         # non-square map, 8 rows x 16
    dict(B=16, C0=512, H=8, W=8, Cout=1024, groups=32, film=True, res=False),             # two blocks per slice (256 workgroups)
    dict(B=8, C0=512, C1=256, H=16, W=16, Cout=512, groups=32, res=True),                 # three blocks per slice: 768 -> 512 (cfg 4's up path)
    dict(B=16, C0=1024, H=8, W=8, Cout=1024, groups=32, film=True, res=True),             # four blocks per slice: ADM-256's 8x8 level
]


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("case", SMALL_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_small_map_conv_applies_the_groupnorm_of_its_input(case, t16):
    """nlc_conv_desc.gn_in (conv_small.hip): conv3x3(act(GroupNorm(cat(x0, x1)) (FiLM))) in ONE launch, the normalisation applied to
    the input slice on its way into LDS from the totals that rode along with the producers of x0 / x1 - against (i) the f32 reference
    on the stored 16-bit inputs, (ii) the separate nlc_groupnorm_prestats + nlc_conv2d launches (same arithmetic up to the f32
    summation order of the split), (iii) its own ride-along statistics against the stored output; deterministic."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(_seed(case))
    B, C0, H, W, Cout, groups = (case[k] for k in ("B", "C0", "H", "W", "Cout", "groups"))
    C1 = case.get("C1", 0)
    Cin = C0 + C1
    # the inputs are themselves convolution outputs (1x1, so that they carry ride-along totals), with a non-zero mean
    def producer(c):
        z = torch.randn(B, 64, H, W, generator=g)
        wz = torch.randn(c, 64, 1, 1, generator=g) / 8.0
        bz = torch.randn(c, generator=g) * 0.5 + 0.3
        return ops.conv2d(_nhwc(z, t16), ops.pack_conv(wz, bz, t16, _dev()))
    x0 = producer(C0)
    x1 = producer(C1) if C1 else None
    assert ops.ride_stats(x0) is not None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    film = case.get("film")
    ss = (torch.randn(B, 2 * Cin + 16, generator=g) * 0.3).to(_dev()) if film else None
    scale, shift = (ss[:, :Cin], ss[:, Cin:2 * Cin]) if film else (None, None)
    emb = (torch.randn(B, Cout + 16, generator=g)).to(_dev()) if case.get("emb") else None
    res = torch.randn(B, Cout, H, W, generator=g) if case.get("res") else None
    pw = ops.pack_conv(w, b, t16, _dev())
    kw = dict(x1=x1, emb=None if emb is None else emb[:, :Cout], res=None if res is None else _nhwc(res, t16),
              act=1 if case.get("act") else 0)
    gd, bd = gamma.to(_dev()), beta.to(_dev())
    # (ii) separate launches
    hn = ops.groupnorm(x0, gd, bd, groups=groups, eps=1e-5, silu=True, x1=x1, scale=scale, shift=shift)
    sep = ops.conv2d(hn, pw, **{k: v for k, v in kw.items() if k != "x1"})
    # the fused launch
    assert ops.conv2d(x0, pw, x1=x1, query_gn_in=True)
    spec = ops.gn_in_spec(x0, gd, bd, groups=groups, eps=1e-5, silu=True, x1=x1, scale=scale, shift=shift)
    assert spec is not None
    got = ops.conv2d(x0, pw, gn_in=spec, **kw)
    again = ops.conv2d(x0, pw, gn_in=spec, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, again)
    # (i) f32 reference from the stored inputs
    xc = torch.cat([x0.float().cpu()] + ([x1.float().cpu()] if C1 else []), dim=-1).permute(0, 3, 1, 2)
    y = F.group_norm(xc, groups, gamma, beta, eps=1e-5)
    if film:
        y = y * (1 + scale.cpu()[:, :, None, None]) + shift.cpu()[:, :, None, None]
    y = _rt(F.silu(y), t16)
    ref = F.conv2d(y, _rt(w, t16), b, padding=1)
    if emb is not None:
        ref = ref + emb.cpu()[:, :Cout, None, None]
    if res is not None:
        ref = ref + _rt(res, t16)
    if case.get("act"):
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    _close(got, ref, _tol(t16), "fused GroupNorm + conv3x3 vs f32 reference")
    scale_ = ref.abs().max().item()
    assert (got.float() - sep.float()).abs().max().item() <= 1.5 * _tol(t16) * scale_          # two roundings of the normalised tensor apart at most
    # (iii) the output's own ride-along totals
    st = ops.ride_stats(got)
    assert st is not None
    gran = Cout // st.shape[1]
    ch = got.float().cpu().view(B, H * W, Cout // gran, gran)
    tot = _totals(st)
    rs, rq = ch.double().sum(dim=(1, 3)), (ch.double() ** 2).sum(dim=(1, 3))
    assert (tot[..., 0] - rs).abs().max() <= 2e-3 * rs.abs().max().clamp(min=1.0)
    assert ((tot[..., 1] - rq) / rq).abs().max() < 2e-3
    # schedule variants: 128-pixel tiles only (tuning bit 22), and with the last arriver reducing alone instead of the distributed
    # reduction (bit 23) - the two reductions add the same partial sums in the same order: bit-identical
    old_t = ops.CONV_TUNING
    try:
        ops.CONV_TUNING = 1 << 22
        t128 = ops.conv2d(x0, pw, gn_in=spec, **kw)
        ops.CONV_TUNING = (1 << 22) | (1 << 23)
        t128_last = ops.conv2d(x0, pw, gn_in=spec, **kw)
        ops.CONV_TUNING = (1 << 22) | (1 << 23) | 1024
        t128_fenced = ops.conv2d(x0, pw, gn_in=spec, **kw)
    finally:
        ops.CONV_TUNING = old_t
    assert torch.equal(t128, t128_last) and torch.equal(t128, t128_fenced)
    _close(t128, ref, _tol(t16), "fused GroupNorm + conv3x3, 128-pixel tiles")
    # the same kernel without a normalisation (policy "small") against the plain convolution of the production dispatch
    old = ops.CONV_POLICY
    try:
        ops.CONV_POLICY = "small"
        plain_small = ops.conv2d(x0, pw, **kw)
    finally:
        ops.CONV_POLICY = old
    plain = ops.conv2d(x0, pw, **kw)
    refp = F.conv2d(xc, _rt(w, t16), b, padding=1)
    if emb is not None:
        refp = refp + emb.cpu()[:, :Cout, None, None]
    if res is not None:
        refp = refp + _rt(res, t16)
    if case.get("act"):
        refp = F.silu(refp)
    _close(plain_small, refp.permute(0, 2, 3, 1), _tol(t16), "small-map kernel without gn_in")
    assert (plain_small.float() - plain.float()).abs().max().item() <= _tol(t16) * refp.abs().max().item()


@pytest.mark.parametrize("t16", T16, ids=T16_IDS)
@pytest.mark.parametrize("shape", [(4, 8, 8, 128), (2, 16, 16, 256), (8, 8, 8, 512)], ids=lambda s: "x".join(map(str, s)))
def test_one_launch_resblock_equals_its_two_fused_launches(shape, t16):
    """nlc_resblock_small (an experiment, not used by the networks): conv1, a grid barrier and conv2 in one launch - the bits of
    nlc_conv2d(gn_in) twice; repeated launches agree (the barrier words and the arrival counters are left zero)."""
    from diffusion_nlc_amd import ops
    g = torch.Generator().manual_seed(_seed(shape))
    B, H, W, Cc = shape
    z = torch.randn(B, 64, H, W, generator=g)
    x = ops.conv2d(_nhwc(z, t16), ops.pack_conv(torch.randn(Cc, 64, 1, 1, generator=g) / 8, torch.randn(Cc, generator=g) * 0.3, t16, _dev()))
    c1 = ops.pack_conv(torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc), torch.randn(Cc, generator=g) * 0.1, t16, _dev())
    c2 = ops.pack_conv(torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc), torch.randn(Cc, generator=g) * 0.1, t16, _dev())
    g1, b1, g2, b2 = [(torch.rand(Cc, generator=g) + 0.5).to(_dev()) if i % 2 == 0 else (torch.randn(Cc, generator=g) * 0.2).to(_dev()) for i in range(4)]
    ss = (torch.randn(B, 2 * Cc, generator=g) * 0.3).to(_dev())
    sc, sh = ss[:, :Cc], ss[:, Cc:]
    h = ops.conv2d(x, c1, gn_in=ops.gn_in_spec(x, g1, b1, groups=32, eps=1e-5, silu=True))
    want = ops.conv2d(h, c2, gn_in=ops.gn_in_spec(h, g2, b2, groups=32, eps=1e-5, silu=True, scale=sc, shift=sh), res=x)
    got = ops.resblock_small(x, c1, c2, g1, b1, g2, b2, groups=32, eps=1e-5, scale=sc, shift=sh)
    assert got is not None
    for _ in range(3):
        again = ops.resblock_small(x, c1, c2, g1, b1, g2, b2, groups=32, eps=1e-5, scale=sc, shift=sh)
        assert torch.equal(again, got)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(ops.ride_stats(got), ops.ride_stats(want))
    assert int(ops._RB_BARRIER[x.device].abs().sum()) == 0
