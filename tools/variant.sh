#!/bin/bash
# Diagnostic: build a copy of the library with extra -D flags on ONE source file and run a tool with it.
#   tools/variant.sh groupnorm "-DGN_UNR=4" tools/gn_bench.py --prestats
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/diffusion-nlc_amd/csrc"
out="$root/gpurun_out/variant"
mkdir -p "$out"
stem="$1"; flags="$2"; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $flags -I"$root/include" -I"$src" -c "$src/$stem.hip" -o "$out/${stem}_v.o"
objs=$(ls "$src"/obj/*.o | grep -v "/$stem.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libnlc_hip_v.so" $objs "$out/${stem}_v.o"
NLC_HIP_LIB="$out/libnlc_hip_v.so" python "$root/$1" "${@:2}"
