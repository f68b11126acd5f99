#!/bin/bash
# rocprofv3 kernel tables of two source trees on one box: tools/ab_profile.sh <old tree> <tag> [configs...]
#   -> gpurun_out/<tag>_kernel_table_{old,new}_<config>.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OLD="$R/$1"; tag="$2"; shift 2
cfgs="${*:-edm32 celebahq256}"
O="$R/gpurun_out"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {  # label, tree, name, nlc steps covered by the trace, bench args...
  local label="$1" tree="$2" name="$3" n="$4"; shift 4
  local d="$O/${tag}_prof_${label}_$name"
  rocprofv3 --kernel-trace --stats -d "$d" -o s --output-format csv -- python3 "$tree/bench.py" --no-cpu-baseline --no-roofline "$@" \
      > "$d.json" 2> "$d.err"
  local csv; csv=$(find "$d" -name 's_kernel_trace.csv' | head -1)
  python3 "$R/tools/trace_table.py" "$csv" --nlc-steps "$n" --top 45 > "$O/${tag}_kernel_table_${label}_$name.txt"
  echo "$label $name: $(head -1 "$O/${tag}_kernel_table_${label}_$name.txt")"
  rm -rf "$d"
}
for c in $cfgs; do
  case "$c" in
    edm32) a="--config edm32 --steps 1 --warmup 1 --timesteps 10"; n=38 ;;
    celebahq256) a="--config celebahq256 --steps 1 --warmup 1 --timesteps 10"; n=20 ;;
    adm256) a="--config adm256 --steps 1 --warmup 1 --timesteps 10"; n=20 ;;
  esac
  run old "$OLD" "$c" "$n" $a
  run new "$R" "$c" "$n" $a
done
