#!/usr/bin/env bash
# PMC passes (separate rocprofv3 --pmc runs) for a 1x1-convolution kernel:  bash tools/pmc_pw.sh <outdir> <which> F K M H W
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
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass sq4 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM
pass grbm GRBM_GUI_ACTIVE
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "GRBM_GUI_ACTIVE"): cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "pw" not in k: continue
    n = max(1, cnt[(k, "SQ_WAVES")] or cnt[(k, "GRBM_GUI_ACTIVE")])
    print(k)
    for c, v in sorted(d.items()): print("   %-34s %.4g" % (c, v))
PY
