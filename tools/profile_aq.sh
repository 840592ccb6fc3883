#!/usr/bin/env bash
# rocprofv3 kernel trace of the Large-AQ bf16 train step + steady-state summary:  gpurun -- 'bash tools/profile_aq.sh <tag>'
set -u
tag=${1:-r03}
root="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_model_aq" -o model -- \
    python "$root/tools/prof_model.py" --tier large --variant rubiks3d-aq --amp bf16 --steps 8 > "$root/gpurun_out/${tag}_model_aq.log" 2>&1
tail -n 1 "$root/gpurun_out/${tag}_model_aq.log"
f=$(find "$root/gpurun_out/${tag}_model_aq" -name '*kernel_trace.csv' | head -1)
python "$root/tools/model_profile_summary.py" "$f" "$root/gpurun_out/${tag}_model_large_aq_bf16_steady.csv" 4
rm -rf "$root/gpurun_out/${tag}_model_aq"
