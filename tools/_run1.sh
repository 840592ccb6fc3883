bash tools/pmc_pw.sh r03_pmc_gemm16 gemm16 256 288 288 14 14 > gpurun_out/r03_pw16_gemm_pmc.txt 2>&1
bash tools/pmc_mem.sh r03_mem_gemm16 gemm16 256 288 288 14 14 > gpurun_out/r03_pw16_gemm_mem.txt 2>&1
bash tools/pmc_pw.sh r03_pmc_wgrad16 wgrad16 256 288 288 14 14 > gpurun_out/r03_pw16_wgrad_pmc.txt 2>&1
bash tools/pmc_mem.sh r03_mem_wgrad16 wgrad16 256 288 288 14 14 > gpurun_out/r03_pw16_wgrad_mem.txt 2>&1
rm -rf gpurun_out/r03_pmc_gemm16 gpurun_out/r03_mem_gemm16 gpurun_out/r03_pmc_wgrad16 gpurun_out/r03_mem_wgrad16
grep -A22 "k_pw16_gemm" gpurun_out/r03_pw16_gemm_mem.txt | head -30
