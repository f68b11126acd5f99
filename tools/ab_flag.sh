#!/bin/bash
# A/B of one module-level switch of diffusion_nlc_amd.ops on ONE box, interleaved: tools/ab_flag.sh <FLAG> <tag> [rounds]
# (bench.py --ops-flag FLAG=0 against FLAG=1 on the three workloads).  Writes gpurun_out/<tag>_ab.log.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
FLAG="$1"; TAG="$2"; ROUNDS="${3:-2}"
OUT="$R/gpurun_out/${TAG}_ab.log"
mkdir -p "$R/gpurun_out"; : > "$OUT"
run() {
  local val="$1"; shift
  local line
  line=$(cd "$R" && timeout -k 10 400 python bench.py --no-cpu-baseline --no-roofline --ops-flag "$FLAG=$val" "$@" 2>/dev/null | grep '^{' | tail -1)
  python3 - "$FLAG=$val" "$line" "$*" >> "$OUT" <<'PY'
import json, sys
label, line, args = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(line)
    print(f"{label:18s} {args:44s} {d['value']:.3f} images/s  ({d['ms_per_step']:.1f} ms/step)")
except Exception as e:
    print(f"{label:18s} {args:44s} FAILED ({e})")
PY
  tail -1 "$OUT"
}
for i in $(seq "$ROUNDS"); do
  for cfg in "--config adm256 --steps 3 --warmup 1" "--config celebahq256 --steps 3 --warmup 1" "--config edm32 --steps 4 --warmup 1"; do
    run 0 $cfg
    run 1 $cfg
  done
done
