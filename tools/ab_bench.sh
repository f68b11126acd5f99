#!/bin/bash
# A/B of two source trees on ONE box, interleaved: tools/ab_bench.sh <old tree> <tag> [rounds]
# e.g. a baseline extracted with `git archive <commit> | tar -x -C build/old_tree` (and built there) against the working tree.
# Writes gpurun_out/<tag>_ab.log: one line per run = tree, config, images/s.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OLD="$1"; TAG="$2"; ROUNDS="${3:-2}"
OUT="$R/gpurun_out/${TAG}_ab.log"
mkdir -p "$R/gpurun_out"; : > "$OUT"
run() {   # tree label, bench args...
  local tree="$1" label="$2"; shift 2
  local line
  line=$(cd "$tree" && timeout -k 10 400 python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep '^{' | tail -1)
  python3 - "$label" "$line" "$*" >> "$OUT" <<'PY'
import json, sys
label, line, args = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(line)
    print(f"{label:4s} {args:40s} {d['value']:.3f} images/s  ({d['ms_per_step']:.1f} ms/step)")
except Exception as e:
    print(f"{label:4s} {args:40s} FAILED ({e})")
PY
  tail -1 "$OUT"
}
for i in $(seq "$ROUNDS"); do
  for cfg in "--config adm256 --steps 3 --warmup 1" "--config celebahq256 --steps 3 --warmup 1" "--config edm32 --steps 4 --warmup 1"; do
    run "$R/$OLD" old $cfg
    run "$R" new $cfg
  done
done
