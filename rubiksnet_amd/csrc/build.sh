#!/usr/bin/env bash
# Builds librubiks_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# .so travels with the repo snapshot to the GPU box.
#   build.sh                 full rebuild of every translation unit (what __graft_entry__.build() runs)
#   RK_INCREMENTAL=1 build.sh   recompile only the units whose source or any included header (.d files) changed
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off: multiplies and adds round separately, exactly as written, so the fp32 /
# fp64 forward and d(x) are bit-identical to the oracle (the op is HBM-bound; FMA buys nothing).
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$root/include" -I"$here"
       -Wall -Wno-unused-function -Wno-implicit-fallthrough)
stale() {   # $1 = unit: true when its object is missing or older than the source / any header it included last time
  local o="$here/$1.o" d="$here/$1.d" f
  [[ -f "$o" && -f "$d" ]] || return 0
  for f in $(sed -e 's/^[^:]*://' -e 's/\\$//' "$d"); do
    [[ -e "$f" && ! "$f" -nt "$o" ]] || return 0
  done
  return 1
}
objs=()
pids=()
for src in rk_misc rk3d rk3d_slab rk2d rk_tshift rk_bn rk_pw rk_pw2 rk_pw3 rk_pw4 rk_pw16 rk_pw16_odd rk_stem16 rk_clip; do
  objs+=("$here/$src.o")
  if [[ "${RK_INCREMENTAL:-0}" == "1" ]] && ! stale "$src"; then continue; fi
  rm -f "$here/$src.o"
  "$HIPCC" "${FLAGS[@]}" ${RK_EXTRA_FLAGS:-} -MD -MF "$here/$src.d" -c "$here/$src.hip" -o "$here/$src.o" &
  pids+=($!)
done
for pid in "${pids[@]:-}"; do
  [[ -n "$pid" ]] || continue
  wait "$pid" || { echo "build.sh: a compile failed" >&2; exit 1; }   # a bare `wait` would swallow the status
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/librubiks_hip.so" "${objs[@]}"
echo "built $here/librubiks_hip.so (${#pids[@]} unit(s) compiled)"
