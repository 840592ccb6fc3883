#!/usr/bin/env bash
# rocprofv3 kernel traces of the Large / Large-AQ(bf16) train steps:  gpurun -- 'bash tools/profile_models.sh r02'
# (summaries: tools/model_profile_summary.py <dir>/..._kernel_trace.csv profiles/<tag>_model_<name>_steady.csv)
set -u
tag=${1:-r02}
root="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
[ -n "${SKIP_LARGE:-}" ] || timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_model_large" -o model -- \
    python "$root/tools/prof_model.py" --tier large --steps 6 > "$root/gpurun_out/${tag}_model_large.log" 2>&1
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_model_aq" -o model -- \
    python "$root/tools/prof_model.py" --tier large --variant rubiks3d-aq --amp bf16 --steps 6 > "$root/gpurun_out/${tag}_model_aq.log" 2>&1
tail -n 2 "$root/gpurun_out/${tag}_model_large.log" "$root/gpurun_out/${tag}_model_aq.log"
