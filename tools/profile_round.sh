#!/usr/bin/env bash
# Everything profiles/ is refreshed from, in one gpurun call:  gpurun -- 'bash tools/profile_round.sh r02'
# (rocprofv3 --kernel-trace --stats and --pmc always in separate runs; each pass under its own timeout)
set -u
tag=${1:-r02}
cd "$GRAFT_REPO_ROOT"
bash tools/rocprof_stats.sh ${tag}_op > gpurun_out/${tag}_op.log 2>&1
python tools/trace_gaps.py gpurun_out/${tag}_op/${tag}_op_kernel_trace.csv > gpurun_out/${tag}_op_gaps.txt 2>&1
bash tools/pmc.sh ${tag}_pmc3d > gpurun_out/${tag}_pmc3d.log 2>&1
bash tools/pmc.sh ${tag}_pmc_tile --shape 32,8,288,14,14 > gpurun_out/${tag}_pmc_tile.log 2>&1
bash tools/pmc.sh ${tag}_pmc_s2 --shape 32,8,54,112,112 --stride 1,2,2 > gpurun_out/${tag}_pmc_s2.log 2>&1
bash tools/pmc.sh ${tag}_pmc_7x7 --shape 32,8,576,7,7 > gpurun_out/${tag}_pmc_7x7.log 2>&1
bash tools/pmc.sh ${tag}_pmc_s2b --shape 32,8,288,28,28 --stride 1,2,2 > gpurun_out/${tag}_pmc_s2b.log 2>&1
PROG=tools/prof_2d.py bash tools/pmc.sh ${tag}_pmc2d_f32 256 64 56 56 float32 > gpurun_out/${tag}_pmc2d_f32.log 2>&1
PROG=tools/prof_2d.py bash tools/pmc.sh ${tag}_pmc2d_bf16 256 64 56 56 bfloat16 > gpurun_out/${tag}_pmc2d_bf16.log 2>&1
cd /tmp && export TMPDIR=/tmp
for shape in 32,8,288,14,14 32,8,576,7,7; do   # (+ the strided 28 -> 14 layer below)
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_small_$shape" -o small -- \
      python "$GRAFT_REPO_ROOT/tools/prof_op.py" --iters 20 --shape $shape > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_small_$shape.log" 2>&1
done
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_s2" -o s2 -- \
    python "$GRAFT_REPO_ROOT/tools/prof_op.py" --iters 20 --shape 32,8,54,112,112 --stride 1,2,2 > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_s2.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_s2b" -o s2b -- \
    python "$GRAFT_REPO_ROOT/tools/prof_op.py" --iters 20 --shape 32,8,288,28,28 --stride 1,2,2 > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_s2b.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_model" -o model -- \
    python "$GRAFT_REPO_ROOT/tools/prof_model.py" --steps 8 > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_model.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.json
ls gpurun_out | head -50
