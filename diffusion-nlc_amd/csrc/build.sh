#!/bin/bash
# Build libnlc_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/../libnlc_hip.so"
mkdir -p "$here/obj"
srcs="abi pack conv_igemm conv_fast conv_halo conv_narrow conv_small conv_pw groupnorm attention elementwise sampler edm constraint"
# objects of sources that no longer exist would be linked in: drop them
for o in "$here"/obj/*.o; do
  [ -e "$o" ] || continue
  b="$(basename "$o" .o)"
  case " $srcs " in *" $b "*) ;; *) rm -f "$o" ;; esac
done
pids=()
for f in $srcs; do
  o="$here/obj/$f.o"
  if [ ! -f "$o" ] || [ "$here/$f.hip" -nt "$o" ] || [ "$here/common.h" -nt "$o" ] || [ "$here/conv_params.h" -nt "$o" ] || [ "$here/conv_small.h" -nt "$o" ] || [ "$root/include/nlc_hip.h" -nt "$o" ] || [ "$here/build.sh" -nt "$o" ]; then
    extra=""
    # the sampler kernels restate the reference's f32 algebra op by op: no FMA contraction there
    # (sqrt(s^2 - sqrt(s^2)^2) must be exactly 0, src/schedulers.py:445-446)
    if [ "$f" = "sampler" ] || [ "$f" = "edm" ]; then extra="-ffp-contract=off"; fi
    hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $extra -I"$root/include" -I"$here" -c "$here/$f.hip" -o "$o" &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ "$rc" = 0 ] || { echo "build failed" >&2; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$here"/obj/*.o
echo "built $out"
