#!/usr/bin/env bash
# PMC summaries of the fp32 1x1 kernels at [256, 288 -> 288, 14, 14] (rk_pw3.hip GEMM, rk_pw2.hip d(weight)) and of the
# second-generation GEMM at [256, 144 -> 144, 28, 28]:  bash tools/profile_pw.sh <tag>   -> gpurun_out/<tag>_pw_*.txt
set -u
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_any.sh ${tag}_pmc_pw3 k_pw3_gemm python tools/pw2_probe.py one gemm 256 288 288 196 1 0 -1 0 > gpurun_out/${tag}_pw_gemm288_pmc.txt 2>&1
bash tools/pmc_any.sh ${tag}_pmc_w288 k_pw2_wgrad python tools/pw2_probe.py one wgrad 256 288 288 196 -1 0 0 > gpurun_out/${tag}_pw_wgrad288_pmc.txt 2>&1
bash tools/pmc_any.sh ${tag}_pmc_g144 k_pw2_gemm python tools/pw2_probe.py one gemm 256 144 144 784 1 0 -1 0 > gpurun_out/${tag}_pw_gemm144_pmc.txt 2>&1
python tools/pw2_probe.py all > gpurun_out/${tag}_pw_probe.txt 2>&1
./tools/bin/mfma_contention_probe > gpurun_out/${tag}_mfma_contention.txt 2>&1
