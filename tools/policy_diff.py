#!/usr/bin/env python3
"""Triage: run one ADM-256 network evaluation (B = 1) under two conv policies and compare every nlc_conv2d output call by call;
also against the f32 model.  Prints the calls whose outputs differ most between the policies.

    python3 tools/policy_diff.py [--dtype bf16]
"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from diffusion_nlc_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(tiny=False, batch=args.batch, timesteps=50, dry_run=False, dtype=args.dtype)
    wl = bench.AdmWorkload(ns, dev, bench.PRECISIONS[args.dtype])
    exp = wl.exp
    z = torch.randn((args.batch, 3, 256, 256), generator=torch.Generator().manual_seed(99))
    S = exp.scheduler
    xT = (z / (1 / (S.sampling_sigmas[0] ** 2 + 1)).sqrt()).to(dev)
    t = torch.full((args.batch,), float(S.timesteps[0]), device=dev)
    c_in = torch.full((args.batch,), float((1 / (S.sampling_sigmas[0] ** 2 + 1)).sqrt()), device=dev)
    real = ops.conv2d
    rec = {}

    def wrap(tag):
        store = rec.setdefault(tag, [])
        def f(x0, pw, **kw):
            out = real(x0, pw, **kw)
            if torch.is_tensor(out):
                store.append((tuple(x0.shape), pw.Cin, pw.Cout, pw.KH, kw.get("stride", 1), bool(kw.get("upsample2x")), kw.get("x1") is not None,
                              out.float().clone()))
            return out
        return f

    outs = {}
    for pol in ("auto", "halo"):
        ops.CONV_POLICY = pol
        ops.conv2d = wrap(pol)
        try:
            outs[pol] = exp.model.run(xT, t, mode="forward", in_scale=c_in).float().clone()
        finally:
            ops.conv2d = real
    ops.CONV_POLICY = "auto"
    for m in (exp.model,):
        bench.set_precision(m, bench.PRECISIONS["f32"])
    ops.conv2d = wrap("f32")
    try:
        outs["f32"] = exp.model.run(xT, t, mode="forward", in_scale=c_in).float().clone()
    finally:
        ops.conv2d = real
    rr = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-12))
    print(f"{args.dtype} B={args.batch}: eps-out relative RMS vs f32: auto {rr(outs['auto'], outs['f32']):.3e}  halo {rr(outs['halo'], outs['f32']):.3e}   "
          f"auto vs halo {rr(outs['auto'], outs['halo']):.3e}")
    n = min(len(rec["auto"]), len(rec["halo"]), len(rec["f32"]))
    print(f"{n} conv calls; per call: relative RMS of the output vs the f32 model's same call (auto | halo) and auto vs halo")
    for i in range(n):
        a, h, f = rec["auto"][i], rec["halo"][i], rec["f32"][i]
        if a[-1].shape != f[-1].shape:
            print(i, "shape mismatch", a[:-1], f[:-1]); break
        ea, eh, d = rr(a[-1], f[-1]), rr(h[-1], f[-1]), rr(a[-1], h[-1])
        flag = "  <<<" if eh > 2.0 * ea + 1e-4 or ea > 2.0 * eh + 1e-4 else ""
        print(f"{i:3d} x{a[0]} Cin {a[1]} Cout {a[2]} k{a[3]} s{a[4]} ups{int(a[5])} cat{int(a[6])}: auto {ea:.2e} | halo {eh:.2e} | a-h {d:.2e}{flag}")


if __name__ == "__main__":
    main()
