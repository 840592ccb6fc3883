#!/usr/bin/env bash
# Steady-state kernel tables of the train steps:  bash tools/prof_models.sh <tag>   -> gpurun_out/<tag>_model_<leg>.csv
set -u
tag="$1"; root="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, prof_model.py args...
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d "$root/gpurun_out/prof_$name" -o t -- python "$root/tools/prof_model.py" "$@" --steps 8 > "$root/gpurun_out/prof_$name.log" 2>&1
  python "$root/tools/model_profile_summary.py" "$(ls $root/gpurun_out/prof_$name/*/t_kernel_trace.csv $root/gpurun_out/prof_$name/t_kernel_trace.csv 2>/dev/null | head -1)" "$root/gpurun_out/${tag}_model_${name}.csv" 5
  rm -rf "$root/gpurun_out/prof_$name"
}
for leg in "$@"; do :; done
run tiny_train_steady --tier tiny
run large_train_steady --tier large
if [ "${2:-}" = "all" ]; then
  run large_aq_bf16_steady --tier large --variant rubiks3d-aq --amp bf16
  run small_train_steady --tier small
fi
