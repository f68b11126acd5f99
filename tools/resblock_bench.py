#!/usr/bin/env python3
"""One small-map ResBlock, three ways: today's per-op launches, the small-map kernel with the GroupNorm of its
input fused (two launches per block), and both convolutions in ONE persistent launch behind a grid barrier.

A chain of --blocks ResBlocks (src/unet_adm.py:236-256 with use_scale_shift_norm: GN+SiLU -> conv3x3 -> GN*(1+scale)+shift+SiLU ->
conv3x3 + x) on a [B, H, W, C] 16-bit tensor, every block with weights of its own (so that a replay streams them from HBM as a
network evaluation does), captured in one hipGraph per variant and replayed interleaved.  Reports us per block and the max
difference between the variants' outputs.

    python tools/resblock_bench.py --shape 16,8,8,1024 [--blocks 12] [--reps 20]
"""
import argparse
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="16,8,8,1024", help="B,H,W,C")
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--variants", default="launches,fused,persistent",
                    help="launches | fused | persistent | fused:<tuning> (nlc_conv_desc.tuning of the fused launches: 1 << 22 = 128-pixel tiles, "
                         "1 << 26 = 256-pixel tiles, (1|2|3) << 24 = 1 / 2 / 4 blocks per k-slice, 1 << 23 = last-arriver reduction)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    B, H, W, C = (int(v) for v in args.shape.split(","))
    g = torch.Generator().manual_seed(0)
    z = (torch.randn(B, H, W, 64, generator=g)).to(dev).to(dt)
    p_in = ops.pack_conv(torch.randn(C, 64, 1, 1, generator=g) / 8, torch.randn(C, generator=g) * 0.2, dt, dev)
    blocks = []
    for _ in range(args.blocks):
        blocks.append(dict(
            c1=ops.pack_conv(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C), torch.randn(C, generator=g) * 0.1, dt, dev),
            c2=ops.pack_conv(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C) * 0.3, torch.randn(C, generator=g) * 0.1, dt, dev),
            g1=(torch.rand(C, generator=g) + 0.5).to(dev), b1=(torch.randn(C, generator=g) * 0.1).to(dev),
            g2=(torch.rand(C, generator=g) + 0.5).to(dev), b2=(torch.randn(C, generator=g) * 0.1).to(dev),
            ss=(torch.randn(B, 2 * C, generator=g) * 0.2).to(dev)))
    arena = ops.StatsArena()

    def gn_conv(x, gam, bet, pw, fused, scale=None, shift=None, res=None):
        if fused:
            spec = ops.gn_in_spec(x, gam, bet, groups=32, eps=1e-5, silu=True, scale=scale, shift=shift)
            assert spec is not None and ops.conv2d(x, pw, query_gn_in=True)
            return ops.conv2d(x, pw, gn_in=spec, res=res)
        return ops.conv2d(ops.groupnorm(x, gam, bet, groups=32, eps=1e-5, silu=True, scale=scale, shift=shift), pw, res=res)

    def chain(variant):
        tuning = 0
        if variant.startswith("fused:"):
            variant, tuning = "fused", int(variant.split(":")[1])
        old_tuning, ops.CONV_TUNING = ops.CONV_TUNING, tuning
        try:
            return chain_(variant)
        finally:
            ops.CONV_TUNING = old_tuning

    def chain_(variant):
        with ops.stats_scope(arena, dev):
            x = ops.conv2d(z, p_in)
            for blk in blocks:
                sc, sh = blk["ss"][:, :C], blk["ss"][:, C:]
                if variant == "persistent":
                    x = ops.resblock_small(x, blk["c1"], blk["c2"], blk["g1"], blk["b1"], blk["g2"], blk["b2"], groups=32, eps=1e-5, scale=sc, shift=sh)
                else:
                    f = variant == "fused"
                    h = gn_conv(x, blk["g1"], blk["b1"], blk["c1"], f)
                    x = gn_conv(h, blk["g2"], blk["b2"], blk["c2"], f, scale=sc, shift=sh, res=x)
            return x

    variants = [v for v in args.variants.split(",") if v != "persistent" or hasattr(ops, "resblock_small")]
    graphs, outs = {}, {}
    for v in list(variants):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(2):
                    out = chain(v)
        except AssertionError:
            print(f"# {v}: not admissible for this shape")
            variants.remove(v)
            continue
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = chain(v)
        gr.replay()
        torch.cuda.synchronize()
        graphs[v], outs[v] = gr, out.float().clone()
    ref = outs[variants[0]]
    for v in variants[1:]:
        print(f"# max |{v} - {variants[0]}| = {(outs[v] - ref).abs().max().item():.3e} (output scale {ref.abs().max().item():.2f})")
    times = {v: [] for v in variants}
    for _ in range(args.rounds):                 # interleaved rounds
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                graphs[v].replay()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(1e3 * e0.elapsed_time(e1) / args.reps / args.blocks)
    fl = 2 * 2.0 * B * H * W * C * C * 9
    base = sorted(times[variants[0]])[len(times[variants[0]]) // 2]
    for v in variants:
        t = sorted(times[v])
        med = t[len(t) // 2]
        print(f"{C}ch @{H}x{W} B={B} {args.dtype} {v:10s}: median {med:7.1f} us per ResBlock (min {t[0]:7.1f})  {fl / med / 1e6:6.0f} TFLOP/s   x{base / med:.2f} vs {variants[0]}", flush=True)


if __name__ == "__main__":
    main()
