cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 300 gpurun_out/r03_bench.json; echo
bash tools/profile_aq.sh r03 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw16 -o pw16 -- python $GRAFT_REPO_ROOT/tools/pw_bf16_time.py > $GRAFT_REPO_ROOT/gpurun_out/r03_pw16_times.txt 2>&1
f=$(find /tmp/pw16 -name '*kernel_stats.csv' | head -1); grep -i "Name\|pw16\|k_pw_gemm_bf16\|k_pw_wgrad_bf16\|k_pw_wgrad_reduce" $f > $GRAFT_REPO_ROOT/gpurun_out/r03_pw16_kernel_stats.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/r03_pw16_kernel_stats.csv | cut -c1-160
