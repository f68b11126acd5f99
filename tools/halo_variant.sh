#!/bin/bash
# Diagnostic: build a copy of the library with extra -D flags on conv_halo.hip and run the conv micro-benchmark with it.
#   tools/halo_variant.sh "-DHALO_NO_STORE" --only 0
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/diffusion-nlc_amd/csrc"
out="$root/gpurun_out/variant"
mkdir -p "$out"
flags="$1"; shift
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $flags -I"$root/include" -I"$src" -c "$src/conv_halo.hip" -o "$out/conv_halo_v.o"
objs=$(ls "$src"/obj/*.o | grep -v conv_halo.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libnlc_hip_v.so" $objs "$out/conv_halo_v.o"
NLC_HIP_LIB="$out/libnlc_hip_v.so" python "$root/tools/conv_bench.py" "$@"
