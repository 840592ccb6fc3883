#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of bench.py (op only), summary -> gpurun_out/<name>/
# usage (on the GPU box, via gpurun): bash tools/rocprof_stats.sh <name> [bench args]
set -u
name=$1; shift
out="$GRAFT_REPO_ROOT/gpurun_out/$name"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$name" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --models none --no-cpu --no-legs "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "rocprofv3 rc=$?"
ls "$out"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f"
