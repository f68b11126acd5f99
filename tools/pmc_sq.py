#!/usr/bin/env python3
"""Fold the two rocprofv3 --pmc SQ passes of tools/profile_round.sh (dominant conv launch) into profiles/rNN_pmc_sq_conv256.json.

    python tools/pmc_sq.py <a_counter_collection.csv> <b_counter_collection.csv> [kernel substring] > profiles/rNN_pmc_sq_conv256.json

Per launch = mean over the profiled dispatches of the kernel.  Units (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed over the 8
XCDs (/ 8 = kernel cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (x 1024 SIMDs for the chip); SQ_WAVE_CYCLES, SQ_WAIT_* and
SQ_ACTIVE_INST_* count quad-cycles of waves.
"""
import collections
import csv
import json
import sys


def read(path, sub):
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # dispatch -> counter -> value
    dur = {}
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]:
            continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(per)
    out = collections.defaultdict(float)
    for d in per.values():
        for k, v in d.items():
            out[k] += v / n
    return dict(out), (sum(dur.values()) / max(len(dur), 1)), n


def main():
    a, b = sys.argv[1:3]
    sub = sys.argv[3] if len(sys.argv) > 3 else "conv_halo_kernel"
    label = sys.argv[4] if len(sys.argv) > 4 else sub + ", 256->256 3x3 @256^2, B=16: 1.237 TFLOP per launch"
    how = sys.argv[5] if len(sys.argv) > 5 else ("rocprofv3 --kernel-trace --pmc <8 SQ counters (+ GRBM_GUI_ACTIVE)> -- python3 tools/conv_bench.py --only 0 "
                                                 "--reps 2 --rounds 1 (two passes, tools/profile_round.sh)")
    ca, us_a, na = read(a, sub)
    cb, us_b, nb = read(b, sub)
    c = {**ca, **cb}
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    der = {
        "kernel_cycles (GRBM_GUI_ACTIVE/8)": cyc,
        "clock_GHz": cyc / (us_a * 1e3),
        "matrix_pipe_busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
        "wave_time_parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "wave_time_issue_stalled (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "wave_time_issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0.0),
    }
    print(json.dumps({"kernel": label, "launch_us_under_profiler": us_a,
                      "dispatches_profiled": [na, nb], "counters_per_launch": c, "derived": der,
                      "how": how},
                     indent=1))


if __name__ == "__main__":
    main()
