#!/usr/bin/env bash
# Builds librubiks_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# .so travels with the repo snapshot to the GPU box.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off: multiplies and adds round separately, exactly as written, so the fp32 /
# fp64 forward and d(x) are bit-identical to the oracle (the op is HBM-bound; FMA buys nothing).
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$root/include" -I"$here"
       -Wall -Wno-unused-function -Wno-implicit-fallthrough)
objs=()
pids=()
for src in rk_misc rk3d rk3d_slab rk2d rk_tshift rk_bn rk_pw rk_pw2 rk_pw3 rk_pw4 rk_pw16 rk_pw16_odd rk_stem16 rk_clip; do
  rm -f "$here/$src.o"
  "$HIPCC" "${FLAGS[@]}" ${RK_EXTRA_FLAGS:-} -c "$here/$src.hip" -o "$here/$src.o" &
  pids+=($!)
  objs+=("$here/$src.o")
done
for pid in "${pids[@]}"; do
  wait "$pid" || { echo "build.sh: a compile failed" >&2; exit 1; }   # a bare `wait` would swallow the status
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/librubiks_hip.so" "${objs[@]}"
echo "built $here/librubiks_hip.so"
