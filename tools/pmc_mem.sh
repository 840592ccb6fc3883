#!/usr/bin/env bash
# memory-side PMC passes for a 1x1-convolution kernel:  bash tools/pmc_mem.sh <outdir> <which> F K M H W
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- \
      python "$GRAFT_REPO_ROOT/tools/prof_pw.py" "${ARGS[@]}" > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
ARGS=("$@")
pass m1 FETCH_SIZE WRITE_SIZE
pass m2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass m3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass m4 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
pass m5 GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-50:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    if "pw" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-34s per launch %.5g  (launches %d)" % (c, v / max(1, n[k][c]), n[k][c]))
PY
