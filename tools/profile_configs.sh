#!/bin/bash
# rocprofv3 kernel tables of the non-headline workloads (cfg 3 EDM-32, cfg 4 CelebA-HQ-256) and of the f32x3 precision of the
# headline model.  Usage on a GPU box: tools/profile_configs.sh r03   -> gpurun_out/<tag>_kernel_table_*.txt
set -uo pipefail
tag="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
O="$R/gpurun_out"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {  # name, nlc steps covered by the trace, bench args...
  local name="$1" n="$2"; shift 2
  rocprofv3 --kernel-trace --stats -d "$O/${tag}_prof_$name" -o s --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --no-roofline "$@" \
      > "$O/${tag}_prof_$name.json" 2> "$O/${tag}_prof_$name.err"
  local csv; csv=$(find "$O/${tag}_prof_$name" -name 's_kernel_trace.csv' | head -1)
  python3 "$R/tools/trace_table.py" "$csv" --nlc-steps "$n" --top 45 > "$O/${tag}_kernel_table_$name.txt"
  echo "$name done: $(head -1 "$O/${tag}_kernel_table_$name.txt")"
}
run celebahq256 20 --config celebahq256 --steps 1 --warmup 1 --timesteps 10
run edm32 38 --config edm32 --steps 1 --warmup 1 --timesteps 10        # 2 x (2*10 - 1) network evaluations
run adm256_f32x3 8 --dtype f32x3 --steps 1 --warmup 1 --timesteps 4
run adm256_f16 20 --dtype f16 --steps 1 --warmup 1 --timesteps 10
