#!/bin/bash
# Long run of the seeded fuzzes and the ragged-batch sweep on a GPU box (the suite runs 160 / 120 / 80 cases of each; this runs thousands).
#   tools/soak.sh [conv cases] [groupnorm cases] [attention cases]      -> gpurun_out/soak.log
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
mkdir -p "$R/gpurun_out"; cd "$R"
{
  NLC_FUZZ_CASES="${1:-3000}" timeout -k 10 900 python -m pytest tests/test_conv_fuzz_gpu.py -q 2>&1 | tail -3
  NLC_FUZZ_CASES="${2:-1500}" timeout -k 10 600 python -m pytest tests/test_gn_fuzz_gpu.py -q 2>&1 | tail -3
  NLC_FUZZ_CASES="${3:-800}" timeout -k 10 600 python -m pytest tests/test_attn_fuzz_gpu.py -q 2>&1 | tail -3
  for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_batch_sweep_gpu.py -q 2>&1 | tail -1; done
} | tee "$R/gpurun_out/soak.log"
