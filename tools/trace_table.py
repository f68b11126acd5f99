#!/usr/bin/env python3
"""Per-(kernel, grid) table from a rocprofv3 --kernel-trace CSV: launches and time per NLC step.

    python tools/trace_table.py gpurun_out/r02b_stats/s_kernel_trace.csv --nlc-steps 20 [--top 40]
"""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--nlc-steps", type=int, default=20, help="NLC timesteps covered by the trace (bench steps incl. warm-up x timesteps)")
ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()
agg = collections.defaultdict(lambda: [0, 0.0])
byname = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(args.csv)):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
    short = m.group(1) if m else r["Kernel_Name"][:40]
    key = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1; agg[key][1] += d
    byname[short][0] += 1; byname[short][1] += d
tot = sum(v[1] for v in agg.values())
n = args.nlc_steps
print(f"total {tot / 1e3 / n:.3f} ms of kernels per NLC step, {sum(v[0] for v in agg.values()) / n:.0f} launches")
print("-- by kernel")
for k, v in sorted(byname.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k:44s} n/step={v[0] / n:6.1f} avg {v[1] / v[0]:8.1f} us  per-step {v[1] / 1e3 / n:7.3f} ms {100 * v[1] / tot:5.1f}%")
print("-- by (kernel, workgroups)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print(f"{k[0]:44s} blocks {k[1]:6d} x{k[2]:3d}  n/step={v[0] / n:6.1f} avg {v[1] / v[0]:8.1f} us  per-step {v[1] / 1e3 / n:7.3f} ms {100 * v[1] / tot:5.1f}%")
