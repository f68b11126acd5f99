#!/bin/bash
# Memory-side (L2 / TCP) and SQ counters of one convolution launch, separate rocprofv3 --pmc passes (counters only, no tracing domain
# besides --kernel-trace; at most 4 TCC or TCP counters fit one pass); per-launch means printed as one JSON object.
# Usage: tools/pw_pmc.sh r04f [H,Cin,Cout,k] [kernel-name substring]      (default: the 512->256 1x1 @256x256 skip projection)
set -uo pipefail
tag="${1:-r04f}"; shape="${2:-256,512,256,1}"; kern="${3:-conv_}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
O="$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
C="python3 $R/tools/conv_bench.py --shape $shape --reps 2 --rounds 1"
pass() { name=$1; shift; timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" -d "$O/${tag}_pw_$name" -o p --output-format csv -- $C > /dev/null 2> "$O/${tag}_pw_$name.err" || echo "pass $name failed"; echo "pass $name done" >> "$O/${tag}_pw_progress.log"; }
# at most 4 TCC / 4 TCP counters per pass (more: "exceeds the capabilities of the hardware to collect")
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcc2 TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_WRITE_sum
pass tcc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc4 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp2 TCP_GATE_EN1_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
python3 - "$O" "$tag" "$kern" <<'PY'
import collections, csv, glob, json, sys
O, tag, kern = sys.argv[1:4]
out = {}
for f in sorted(glob.glob(f"{O}/{tag}_pw_*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]:
            continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = max(len(per), 1)
    for d in per.values():
        for k, v in d.items():
            out[k] = out.get(k, 0.0) + v / n
    out.setdefault("launch_us", {})[f.split("_pw_")[1].split("/")[0]] = round(sum(dur.values()) / max(len(dur), 1), 1)
print(json.dumps(out, indent=1))
PY
