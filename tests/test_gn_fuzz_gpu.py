"""Randomised sweep of the GroupNorm launches (nlc_groupnorm_prestats / the statistics + apply pair / nlc_groupnorm_pool2x2): which form a
call takes depends on whether its input(s) carry ride-along statistics (and from which producing kernel, at which channel granule),
on concatenation, on the group width and on the map size.  Every case: the input(s) either come out of a random convolution of this
library (so the totals ride along - whichever kernel the production dispatch picked for that shape emitted them) or are plain tensors;
GroupNorm (+FiLM) (+SiLU) over cat(x0, x1) is launched twice behind cache sweeps and once beside a busy second stream - bit-identical -
and compared with F.group_norm on the CPU in f32 on the very values the kernel read.  Same for the fused GroupNorm + 2x2 average pool.

Seeded: the same cases in every run.  NLC_FUZZ_CASES=1000 for a soak.
"""
import math
import os
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
N_CASES = int(os.environ.get("NLC_FUZZ_CASES", "120"))
TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3, torch.float32: 2e-4}


def _cases(n, seed=5102026):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        B = rng.choice([1, 2, 3, 4, 5, 8, 16])
        H = rng.choice([4, 8, 8, 16, 16, 32, 32, 64, 128])
        W = H if rng.random() < 0.8 else rng.choice([8, 16, 32, 64])
        c0 = rng.choice([64, 128, 128, 256, 256, 384, 512, 1024])
        c1 = rng.choice([0, 0, 0, 64, 128, 256, 512]) if rng.random() < 0.5 else 0
        if B * H * W * (c0 + c1) > 24 << 20:
            continue
        ctot = c0 + c1
        groups = 32 if rng.random() < 0.8 else min(32, ctot // 4)
        if ctot % groups:
            continue
        out.append(dict(B=B, H=H, W=W, c0=c0, c1=c1, groups=groups, eps=rng.choice([1e-5, 1e-6]), silu=rng.random() < 0.7,
                        film=rng.random() < 0.4, affine=rng.random() < 0.85, prod0=rng.choice([0, 1, 3]), prod1=rng.choice([0, 1, 3]),
                        dtype=rng.choice(["bf16", "bf16", "f16", "f32"]), pool=rng.random() < 0.25, seed=rng.randrange(1 << 30)))
    return out


def _produce(g, B, H, W, c, k, dt):
    """An activation tensor [B, H, W, c]: out of a k x k convolution of this library (k = 1 / 3: ride-along totals attached when the
    launch supports them) or a plain tensor (k = 0)."""
    from diffusion_nlc_amd import ops
    if k == 0 or dt == torch.float32:
        return (torch.randn(B, H, W, c, generator=g) * 1.5 + 0.3).to(DEV).to(dt)
    cin = 64
    x = torch.randn(B, H, W, cin, generator=g).to(DEV).to(dt)
    w = torch.randn(c, cin, k, k, generator=g) / math.sqrt(cin * k * k) * 1.5
    return ops.conv2d(x, ops.pack_conv(w, torch.randn(c, generator=g) * 0.3, dt, torch.device(DEV)), emit_stats=True)


def _same(a, b) -> bool:
    return bool(torch.equal(a, b))


def _run_case(c, sweep, hog, side):
    from diffusion_nlc_amd import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[c["dtype"]]
    g = torch.Generator(device="cpu").manual_seed(c["seed"])
    B, H, W, c0, c1 = c["B"], c["H"], c["W"], c["c0"], c["c1"]
    ctot = c0 + c1
    x0 = _produce(g, B, H, W, c0, c["prod0"], dt)
    x1 = _produce(g, B, H, W, c1, c["prod1"], dt) if c1 else None
    gamma = (torch.randn(ctot, generator=g) * 0.3 + 1.0).to(DEV) if c["affine"] else None
    beta = (torch.randn(ctot, generator=g) * 0.2).to(DEV) if c["affine"] else None
    emb = (torch.randn(B, 2 * ctot, generator=g) * 0.3).to(DEV) if c["film"] else None
    scale, shift = (emb[:, :ctot], emb[:, ctot:]) if emb is not None else (None, None)
    pooled = c["pool"] and x1 is None and ops.groupnorm_pool2x2_supported(x0)

    def launch():
        if pooled:
            return ops.groupnorm_pool2x2(x0, gamma, beta, groups=c["groups"], eps=c["eps"], silu=c["silu"], scale=scale, shift=shift)
        return (ops.groupnorm(x0, gamma, beta, groups=c["groups"], eps=c["eps"], silu=c["silu"], x1=x1, scale=scale, shift=shift),)

    outs = []
    for rep in range(2):
        sweep.fill_(rep)
        outs.append(launch())
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for rep in range(2):
            hog.fill_(rep)
    outs.append(launch())
    torch.cuda.current_stream().wait_stream(side)
    for rep in (1, 2):
        for a, b in zip(outs[rep], outs[0]):
            assert _same(a, b), f"launch {rep} differs from the first{' (beside a busy stream)' if rep == 2 else ''}"
    xc = torch.cat([x0] + ([x1] if x1 is not None else []), dim=-1).float().cpu().permute(0, 3, 1, 2)
    ref = F.group_norm(xc, c["groups"], None if gamma is None else gamma.cpu(), None if beta is None else beta.cpu(), c["eps"])
    if emb is not None:
        ref = ref * (1 + scale.cpu()[:, :, None, None]) + shift.cpu()[:, :, None, None]
    if c["silu"]:
        ref = F.silu(ref)
    tol = TOL[dt]
    if pooled:
        want = (F.avg_pool2d(ref, 2), F.avg_pool2d(xc, 2))
    else:
        want = (ref,)
    for got, w_ in zip(outs[0], want):
        w_ = w_.permute(0, 2, 3, 1)
        sc = max(w_.abs().max().item(), 1e-6)
        err = (got.float().cpu() - w_).abs().max().item() / sc
        assert err <= tol, f"max rel-to-scale error {err:.3e} > {tol:.0e} (scale {sc:.3e}, pooled={pooled})"


def test_groupnorm_fuzz():
    sweep = torch.empty(64 << 20, device=DEV, dtype=torch.uint8)
    hog = torch.empty(256 << 20, device=DEV, dtype=torch.uint8)
    side = torch.cuda.Stream()
    failures = []
    for i, c in enumerate(_cases(N_CASES)):
        try:
            _run_case(c, sweep, hog, side)
        except AssertionError as e:
            failures.append(f"case {i} {c}: {e}")
        except Exception as e:
            failures.append(f"case {i} {c}: {type(e).__name__}: {e}")
    assert not failures, f"{len(failures)} of {N_CASES} cases failed:\n" + "\n".join(failures[:12])
