#!/bin/bash
# Diagnostic: build a copy of the library whose halo conv kernel carries s_memtime stamps (-DHALO_STAMP) and print
# the per-wave phase timeline of one k-step.  Not part of the product build.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/diffusion-nlc_amd/csrc"
out="$root/build/stamp"
mkdir -p "$out"
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -DHALO_STAMP -DHALO_STAMP_TAP=${HALO_STAMP_TAP:-4} -I"$root/include" -I"$src" -c "$src/conv_halo.hip" -o "$out/conv_halo_stamp.o"
objs=$(ls "$src"/obj/*.o | grep -v conv_halo.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libnlc_hip_stamp.so" $objs "$out/conv_halo_stamp.o"
NLC_HIP_LIB="$out/libnlc_hip_stamp.so" python "$root/tools/halo_stamps.py" "$@"
