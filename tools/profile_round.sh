#!/usr/bin/env bash
# Everything profiles/ is refreshed from, in one gpurun call:  gpurun -- 'bash tools/profile_round.sh r01'
# (rocprofv3 --kernel-trace --stats and --pmc always in separate runs; each pass under its own timeout)
set -u
tag=${1:-r01}
cd "$GRAFT_REPO_ROOT"
bash tools/rocprof_stats.sh ${tag}_op > gpurun_out/${tag}_op.log 2>&1
bash tools/rocprof_stats.sh ${tag}_model --model tiny > gpurun_out/${tag}_model.log 2>&1
bash tools/pmc.sh ${tag}_pmc3d > gpurun_out/${tag}_pmc3d.log 2>&1
PROG=tools/prof_2d.py bash tools/pmc.sh ${tag}_pmc2d_f32 256 64 56 56 float32 > gpurun_out/${tag}_pmc2d_f32.log 2>&1
PROG=tools/prof_2d.py bash tools/pmc.sh ${tag}_pmc2d_bf16 256 64 56 56 bfloat16 > gpurun_out/${tag}_pmc2d_bf16.log 2>&1
PROG=tools/prof_bn.py bash tools/pmc.sh ${tag}_pmcbn 256 54 56 56 float32 > gpurun_out/${tag}_pmcbn.log 2>&1
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
ls gpurun_out
