#!/bin/bash
# SQ counters of the register-resident attention kernel at the ADM-256 32x32 level (8 heads x 64, T = 1024, B = 16), two rocprofv3
# --pmc passes (8 SQ slots each; counters only, no tracing domain besides --kernel-trace).  Usage: tools/attn_pmc.sh r03
set -uo pipefail
tag="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
O="$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
C="python3 $R/tools/attn_bench.py 1024 8 5"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$O/${tag}_attn_sq_a" -o a --output-format csv -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_MISC -d "$O/${tag}_attn_sq_b" -o b --output-format csv -- $C > /dev/null 2>&1
a=$(find "$O/${tag}_attn_sq_a" -name '*counter_collection.csv' | head -1)
b=$(find "$O/${tag}_attn_sq_b" -name '*counter_collection.csv' | head -1)
python3 "$R/tools/pmc_sq.py" "$a" "$b" attn_d64_kernel "attn_d64_kernel (bf16, log2 logits, D=64, T=1024, 8 heads, B=16): 34.4 GFLOP of QK^T + PV per launch (+ 25 % MFMA work for the row sums)" \
    "rocprofv3 --kernel-trace --pmc <8 SQ counters> -- python3 tools/attn_bench.py 1024 8 5 (two passes, tools/attn_pmc.sh)" > "$O/${tag}_attn_pmc.json"
python3 - "$O/${tag}_attn_pmc.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["counters_per_launch"]
print("attention d64: launch", round(d["launch_us_under_profiler"], 1), "us;", {k: round(v, 3) for k, v in d["derived"].items() if isinstance(v, float) and v < 10},
      "VALU per MFMA", round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2))
PY
