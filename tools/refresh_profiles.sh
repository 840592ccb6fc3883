#!/usr/bin/env bash
# gpurun_out/<tag>_* (tools/profile_round.sh, tools/profile_models.sh) -> the tracked summaries under profiles/
set -eu
tag=${1:-r02}
cd "$(dirname "$0")/.."
python tools/make_profile_summary.py ${tag} gpurun_out/${tag}_op gpurun_out/${tag}_pmc3d > /dev/null
python tools/make_profile_summary.py ${tag}_tile14 "gpurun_out/${tag}_small_32,8,288,14,14" gpurun_out/${tag}_pmc_tile > /dev/null
if [ -d gpurun_out/${tag}_pmc_7x7 ]; then python tools/make_profile_summary.py ${tag}_7x7 "gpurun_out/${tag}_small_32,8,576,7,7" gpurun_out/${tag}_pmc_7x7 > /dev/null
else python tools/make_profile_summary.py ${tag}_7x7 "gpurun_out/${tag}_small_32,8,576,7,7" > /dev/null; fi
[ -d gpurun_out/${tag}_s2b ] && python tools/make_profile_summary.py ${tag}_stride2_28to14 gpurun_out/${tag}_s2b gpurun_out/${tag}_pmc_s2b > /dev/null
python tools/make_profile_summary.py ${tag}_stride2 gpurun_out/${tag}_s2 gpurun_out/${tag}_pmc_s2 > /dev/null
python tools/make_profile_summary.py ${tag}_2d_f32 - gpurun_out/${tag}_pmc2d_f32 > /dev/null
python tools/make_profile_summary.py ${tag}_2d_bf16 - gpurun_out/${tag}_pmc2d_bf16 > /dev/null
python tools/model_profile_summary.py gpurun_out/${tag}_model/model_kernel_trace.csv profiles/${tag}_model_tiny_train_steady.csv 4 | tail -1
[ -f gpurun_out/${tag}_model_large/model_kernel_trace.csv ] && python tools/model_profile_summary.py gpurun_out/${tag}_model_large/model_kernel_trace.csv profiles/${tag}_model_large_train_steady.csv 4 | tail -1
[ -f gpurun_out/${tag}_model_aq/model_kernel_trace.csv ] && python tools/model_profile_summary.py gpurun_out/${tag}_model_aq/model_kernel_trace.csv profiles/${tag}_model_large_aq_bf16_steady.csv 4 | tail -1
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
cp gpurun_out/${tag}_op_gaps.txt profiles/${tag}_op_trace_gaps.txt
