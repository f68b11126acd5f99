#!/bin/bash
# Collect the round's profiling evidence on a GPU box (rocprofv3; counters in their own passes, never with tracing domains
# other than --kernel-trace).  Usage: tools/profile_round.sh r02   -> gpurun_out/<tag>_*; copy the summaries into profiles/.
set -uo pipefail
tag="${1:-r02}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
O="$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d "$O/${tag}_stats" -o s --output-format csv -- $B --steps 1 --warmup 1 --timesteps 10 > "$O/${tag}_stats.json" 2> "$O/${tag}_stats.err"
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/${tag}_step_fetch" -o f --output-format csv -- $B --steps 1 --warmup 0 --timesteps 2 > /dev/null 2> "$O/${tag}_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/${tag}_step_write" -o w --output-format csv -- $B --steps 1 --warmup 0 --timesteps 2 > /dev/null 2> "$O/${tag}_write.err"
echo "step traffic done"
C="python3 $R/tools/conv_bench.py --only 0 --reps 2 --rounds 1 --no-check"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/${tag}_dom_fetch" -o f --output-format csv -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/${tag}_dom_write" -o w --output-format csv -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$O/${tag}_dom_sq_a" -o a --output-format csv -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM -d "$O/${tag}_dom_sq_b" -o b --output-format csv -- $C > /dev/null 2>&1
echo "dominant launch counters done"
ls "$O" | grep "^${tag}_"
