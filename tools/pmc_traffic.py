#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs of the same command, CSV output) into the HBM-traffic
JSON that bench.py attaches to its `roofline` object.

    python tools/pmc_traffic.py <step_fetch.csv> <step_write.csv> <dominant_fetch.csv> <dominant_write.csv> <nlc_steps> > profiles/rNN_pmc_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of 16-byte-per-lane streaming reads
(global_load and LDS-DMA alike) -> x 2; WRITE_SIZE is exact for 16-byte-per-lane stores; both in KiB.
"""
import collections
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FAMILIES = ["conv_halo", "gn_apply", "conv_fast_kernel<unsigned short, 1", "conv_fast_kernel<unsigned short, 9", "conv_fast_kernel<float",
            "conv_pw", "conv_igemm", "splitk_reduce", "avgpool", "attn_d64", "attn_kernel", "gn_finalize", "gn_stats", "conv_first", "upsample", "quantile",
            "sched_", "row_sumsq"]
CONV = ("conv_halo", "conv_fast_kernel", "conv_pw", "conv_igemm", "splitk_reduce")


def csrc_sha16():
    h = hashlib.sha256()
    for f in sorted((ROOT / "diffusion-nlc_amd" / "csrc").glob("conv_*")):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def read(path, counter):
    out = collections.defaultdict(lambda: [0, 0.0])          # kernel name -> [dispatches, KiB]
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        out[k][1] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            out[k][0] += 1
    return out


def family(name):
    for f in FAMILIES:
        if f in name:
            return f
    return "other"


def main():
    sf, sw, df, dw, nsteps = sys.argv[1:6]
    nsteps = int(nsteps)
    fe, wr = read(sf, "FETCH_SIZE"), read(sw, "WRITE_SIZE")
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k, (n, kib) in fe.items():
        f = fam[family(k)]
        f[0] += n; f[1] += 2.0 * kib * 1024
    for k, (n, kib) in wr.items():
        fam[family(k)][2] += kib * 1024
    conv_bytes = sum(v[1] + v[2] for k, v in fam.items() if any(c in k for c in CONV))
    # nlc_conv2d launches = every conv kernel launch except the split-K reduce launches that belong to a conv_fast launch
    conv_launches = sum(v[0] for k, v in fam.items() if any(c in k for c in CONV) and "splitk_reduce" not in k)
    total = sum(v[1] + v[2] for v in fam.values())
    dfe, dwr = read(df, "FETCH_SIZE"), read(dw, "WRITE_SIZE")
    dk = [k for k in dfe if "conv_halo" in k][0]
    d_read = 2.0 * dfe[dk][1] * 1024 / dfe[dk][0]
    d_write = dwr[dk][1] * 1024 / dwr[dk][0]
    B, H, C = 16, 256, 256
    algo = B * H * H * C * 2 * 2 + 256 * 9 * 256 * 2
    out = {
        "csrc_sha16": csrc_sha16(),
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                   f"--timesteps {nsteps} --no-cpu-baseline --no-roofline   (whole step);  -- python3 tools/conv_bench.py --only 0 --reps 2   (dominant launch)",
        "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of 16-B/lane streaming reads; MI355X_MICROARCH.md HBM section)",
        "nlc_steps": nsteps,
        "total_bytes_per_nlc_step": total / nsteps,
        "conv2d_launches": conv_launches,
        "conv2d_bytes_per_launch": conv_bytes / max(conv_launches, 1),
        "dominant": {"kernel": "conv_halo_kernel<bf16>, 256->256 3x3 @256^2, B=16 (31 % of the conv FLOPs)",
                     "read_bytes_per_launch": d_read, "write_bytes_per_launch": d_write,
                     "traffic_bytes_per_launch": d_read + d_write, "algorithmic_bytes_per_launch": algo},
        # PER NLC STEP (the traces cover `nlc_steps` of them: earlier rounds' files listed the trace totals here and were read as
        # per-step figures)
        "per_kernel_per_nlc_step": [{"kernel": k, "launches": v[0] / nsteps, "read_GB": v[1] / 1e9 / nsteps, "write_GB": v[2] / 1e9 / nsteps}
                                    for k, v in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))],
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
