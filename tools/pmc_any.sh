#!/usr/bin/env bash
# PMC passes (separate rocprofv3 --pmc runs, no tracing) for any command:
#   bash tools/pmc_any.sh <outdir under gpurun_out> <kernel-name substring> <command ...>
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; pat="$2"; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  ( cd "$GRAFT_REPO_ROOT" && timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- "${CMD[@]}" > "$out/$name.log" 2>&1 )
  echo "pass $name rc=$?"
}
CMD=("$@")
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
[ -n "${SKIP_M1:-}" ] || pass m1 FETCH_SIZE WRITE_SIZE      # (this pass hung on two boxes of round 5: SKIP_M1=1 leaves it out; TCC_MISS x 128 B gives the same traffic)
pass m2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass m3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass grbm GRBM_GUI_ACTIVE
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    if pat not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-34s per launch %.5g  (launches %d)" % (c, v / max(1, n[k][c]), n[k][c]))
PY
