#!/usr/bin/env python3
"""Which GroupNorm passes still run as launches of their own?  One network evaluation pair (encode + forward) of a workload with
ops.groupnorm / groupnorm_pool2x2 / conv2d hooked: counts per input shape, and per shape of the small-map convolutions that took
(or did not take) the fused form.

    python tools/gn_census.py [adm256|celebahq256|edm32]
"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from diffusion_nlc_amd import ops  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "celebahq256"
    args = bench.argparse.Namespace(tiny=False, batch=0, timesteps=0, dry_run=False, dtype="bf16")
    wl = bench.WORKLOADS[name](args, torch.device("cuda:0"), bench.PRECISIONS["bf16"])
    gn, fused, plain3 = collections.Counter(), collections.Counter(), collections.Counter()
    og, oc = ops.groupnorm, ops.conv2d

    def groupnorm(x0, *a, **k):
        x1 = k.get("x1")
        gn[(tuple(x0.shape), None if x1 is None else x1.shape[-1], ops.ride_stats(x0) is not None)] += 1
        return og(x0, *a, **k)

    def conv2d(x0, pw, **k):
        if not (k.get("query_gn_in") or k.get("query_prologue") or k.get("query_norm_out")) and pw.KH == 3 and x0.dim() == 4 and x0.shape[2] <= 32:
            x1 = k.get("x1")
            key = (tuple(x0.shape), None if x1 is None else x1.shape[-1], pw.Cout, k.get("stride", 1), bool(k.get("upsample2x")))
            (fused if k.get("gn_in") is not None else plain3)[key] += 1
        return oc(x0, pw, **k)

    ops.groupnorm, ops.conv2d = groupnorm, conv2d
    import diffusion_nlc_amd.hipnet as hipnet
    xs = wl.inputs(1, 1, 0)
    exp = wl.exp
    if name == "edm32":
        wl.run(xs[0])
    else:
        import types
        exp.denoise_loop(shape=wl.shape, xT=xs[0], style="pred", norm_eps=True, refine_prior_sigma=True, return_log=False, chunk_size=1,
                         sigma_pred_threshold=960, return_on_device=True, max_steps=1,
                         **({"constrain_fn": wl.bound} if name == "celebahq256" else {}))
    torch.cuda.synchronize()
    print(f"== {name}: GroupNorm launches of their own (input shape, C1, input has ride-along statistics): count")
    for k, v in sorted(gn.items(), key=lambda kv: (-kv[0][0][1], str(kv[0]))):
        print("  ", k, v)
    print("== 3x3 convolutions on maps <= 32 wide that took the fused small-map form (input shape, C1, Cout, stride, ups): count")
    for k, v in sorted(fused.items(), key=str):
        print("  ", k, v)
    print("== ... and those that did not")
    for k, v in sorted(plain3.items(), key=str):
        print("  ", k, v)


if __name__ == "__main__":
    main()
