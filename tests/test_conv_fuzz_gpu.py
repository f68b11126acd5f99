"""Randomised sweep of nlc_conv2d's PRODUCTION dispatch (16-bit): which kernel a launch takes - LDS-halo (CF / general / split-K),
small-map with or without split-K, conv_fast, resident / streaming pointwise, generic - and how many tiles each persistent workgroup
walks is a function of (batch, map size, channels, concatenation, stride, fused upsample, residual, embedding, statistics), and the
benchmark's shapes visit only a few points of that space, all of them with whole rounds of tiles.  Every case here is launched twice
(a 64 MB fill in between: cold weights / bias) and a third time beside a second stream that hammers the memory system - all three
results must be BIT-IDENTICAL, fixed summation orders everywhere (a counted wait that allows one operation too many shows here) - and is
compared with the library's exact-f32 generic kernel on the operands as rounded to the compute dtype (itself checked against torch's
CPU convolution on the cases below 30 GFLOP), the ride-along GroupNorm totals with the sums of the stored output in f64.

The case list is seeded: the same cases in every run.  tools/-style use for a longer soak:  NLC_FUZZ_CASES=2000 pytest -m gpu -k fuzz
"""
import math
import os
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
N_CASES = int(os.environ.get("NLC_FUZZ_CASES", "160"))


def _cases(n, seed=20261005):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([3, 3, 3, 1])
        B = rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16])
        H = rng.choice([8, 16, 16, 24, 32, 32, 48, 64, 64, 96, 128, 128, 256])
        W = H if rng.random() < 0.7 else rng.choice([8, 16, 32, 64, 128, 256])
        cin = rng.choice([64, 128, 128, 192, 256, 256, 320, 384, 512, 768, 1024])
        cout = rng.choice([64, 128, 128, 256, 256, 384, 512, 1024])
        if B * H * W * (cin + 2 * cout) > 48 << 20:            # keep a case under ~100 MB of bf16 tensors
            continue
        split = 0
        if rng.random() < 0.35 and cin >= 128:
            split = rng.choice([c for c in range(64, cin, 64)])
        stride = 2 if (k == 3 and rng.random() < 0.1 and H % 2 == 0 and W % 2 == 0) else 1
        ups = k == 3 and stride == 1 and rng.random() < 0.15 and H <= 64 and W <= 64
        out.append(dict(k=k, B=B, H=H, W=W, cin=cin, cout=cout, split=split, stride=stride, ups=ups, res=rng.random() < 0.4,
                        emb=rng.random() < 0.4, stats=rng.random() < 0.7, dtype=rng.choice(["bf16", "bf16", "f16"]), seed=rng.randrange(1 << 30)))
    return out


_STATE = {}


def _same(a, b) -> bool:          # (a plain bool: pytest's assertion rewriting would otherwise print both tensors)
    return bool(torch.equal(a, b))


def _side():
    if "side" not in _STATE:
        _STATE["side"] = torch.cuda.Stream()
    return _STATE["side"]


def _hog():
    if "hog" not in _STATE:
        _STATE["hog"] = torch.empty(256 << 20, device=DEV, dtype=torch.uint8)
    return _STATE["hog"]


def _run_case(c, sweep):
    from diffusion_nlc_amd import ops
    dt = torch.bfloat16 if c["dtype"] == "bf16" else torch.float16
    g = torch.Generator(device="cpu").manual_seed(c["seed"])
    B, H, W, cin, cout, k = c["B"], c["H"], c["W"], c["cin"], c["cout"], c["k"]
    x = torch.randn(B, H, W, cin, generator=g).to(DEV).to(dt)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    pw = ops.pack_conv(w, b, dt, torch.device(DEV))
    x0, x1 = (x[..., :c["split"]].contiguous(), x[..., c["split"]:].contiguous()) if c["split"] else (x, None)
    s = c["stride"]
    Ho, Wo = (2 * H, 2 * W) if c["ups"] else ((H + s - 1) // s if k == 3 else H, (W + s - 1) // s if k == 3 else W)
    res = torch.randn(B, Ho, Wo, cout, generator=g).to(DEV).to(dt) if c["res"] else None
    emb = torch.randn(B, cout, generator=g).to(DEV) if c["emb"] else None
    outs = []
    for rep in range(2):
        sweep.fill_(rep)
        y = ops.conv2d(x0, pw, x1=x1, stride=s, upsample2x=c["ups"], res=res, emb=emb, emit_stats=c["stats"])
        outs.append((y, getattr(y, "_nlc_stats", None)))
    assert _same(outs[0][0], outs[1][0]), "two identical launches differ"
    # ... and a third one beside a second stream that keeps the memory system busy (fills of a 256 MB buffer; they need no LDS, so they
    # share the CUs with the convolution's workgroups): every DMA and load of the launch takes longer and lands in a different order
    side = _side()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for rep in range(3):
            _hog().fill_(rep)
    y3 = ops.conv2d(x0, pw, x1=x1, stride=s, upsample2x=c["ups"], res=res, emb=emb, emit_stats=c["stats"])
    torch.cuda.current_stream().wait_stream(side)
    assert _same(y3, outs[0][0]), "a launch beside a busy second stream differs from the undisturbed one"
    if outs[0][1] is not None:
        assert _same(getattr(y3, "_nlc_stats"), outs[0][1]), "ride-along totals differ beside a busy second stream"
    if outs[0][1] is not None:
        assert outs[1][1] is not None and torch.equal(outs[0][1], outs[1][1]), "ride-along totals of two identical launches differ"
    y, st = outs[0]
    assert tuple(y.shape) == (B, Ho, Wo, cout)
    # reference: the same operands (as rounded to the compute dtype) through the library's exact-f32 generic kernel (conv_igemm, f32
    # MFMA, one summation order; pinned against torch's CPU convolution by tests/test_ops_gpu.py) - and, for the smaller cases, torch's
    # CPU convolution itself
    pw32 = ops.pack_conv(w.to(dt).float(), b, torch.float32, torch.device(DEV))
    old = ops.CONV_POLICY
    ops.CONV_POLICY = "generic"
    try:
        ref = ops.conv2d(x0.float(), pw32, x1=None if x1 is None else x1.float(), stride=s, upsample2x=c["ups"],
                         res=None if res is None else res.float(), emb=emb, emit_stats=False)
    finally:
        ops.CONV_POLICY = old
    if 2.0 * B * Ho * Wo * cout * cin * k * k < 3e10:
        xin = x.float().permute(0, 3, 1, 2).cpu()
        if c["ups"]:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        rc = F.conv2d(xin, w.to(dt).float(), b, stride=s, padding=k // 2)
        if emb is not None:
            rc = rc + emb.cpu()[:, :, None, None]
        if res is not None:
            rc = rc + res.float().cpu().permute(0, 3, 1, 2)
        rc = rc.permute(0, 2, 3, 1)
        e32 = (ref.cpu() - rc).abs().max().item() / max(rc.abs().max().item(), 1e-6)
        assert e32 <= 2e-4, f"f32 generic kernel vs torch CPU: {e32:.3e}"
    scale = max(ref.abs().max().item(), 1e-6)
    err = (y.float() - ref).abs().max().item() / scale
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert err <= tol, f"max rel-to-scale error {err:.3e} > {tol:.0e} (scale {scale:.3e})"
    if st is not None:
        gran = cout // st.shape[1]
        t = st.double()
        tot_s, tot_q = t[..., 0] + t[..., 1] * 2.0 ** -44, t[..., 2] + t[..., 3] * 2.0 ** -44      # [B, C/g]
        yd = y.double().view(B, Ho * Wo, cout // gran, gran)
        want_s, want_q = yd.sum(dim=(1, 3)), (yd * yd).sum(dim=(1, 3))
        n = Ho * Wo * gran
        e_s = ((tot_s - want_s).abs() / (want_q * n).sqrt().clamp_min(1e-6)).max().item()          # |sum| <= sqrt(n * sumsq)
        e_q = ((tot_q - want_q).abs() / want_q.clamp_min(1e-6)).max().item()
        assert e_s <= 1e-4 and e_q <= 1e-4, f"ride-along totals off: sum {e_s:.3e}, sum of squares {e_q:.3e} (granule {gran})"


def test_conv2d_dispatch_fuzz():
    from diffusion_nlc_amd import ops
    assert ops.CONV_POLICY == "auto"
    sweep = torch.empty(64 << 20, device=DEV, dtype=torch.uint8)
    failures = []
    for i, c in enumerate(_cases(N_CASES)):
        try:
            _run_case(c, sweep)
        except AssertionError as e:
            failures.append(f"case {i} {c}: {e}")
        except Exception as e:                                    # NlcError etc.: a launch the dispatch should have routed elsewhere
            failures.append(f"case {i} {c}: {type(e).__name__}: {e}")
    assert not failures, f"{len(failures)} of {N_CASES} cases failed:\n" + "\n".join(failures[:12])
