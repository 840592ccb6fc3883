#!/usr/bin/env bash
# rocprofv3 kernel trace of the Tiny (or $TIER) train step + steady-state summary:  gpurun -- 'bash tools/profile_tiny.sh <tag> [tier]'
set -u
tag=${1:-r03}; tier=${2:-tiny}
root="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_model_${tier}" -o model -- \
    python "$root/tools/prof_model.py" --tier "$tier" --steps 8 > "$root/gpurun_out/${tag}_model_${tier}.log" 2>&1
tail -n 1 "$root/gpurun_out/${tag}_model_${tier}.log"
f=$(find "$root/gpurun_out/${tag}_model_${tier}" -name '*kernel_trace.csv' | head -1)
python "$root/tools/model_profile_summary.py" "$f" "$root/gpurun_out/${tag}_model_${tier}_train_steady.csv" 4
rm -rf "$root/gpurun_out/${tag}_model_${tier}"
