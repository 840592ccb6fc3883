#!/usr/bin/env bash
# Collect PMC counters for the op in separate passes (rocprofv3 --pmc only; no trace domains).
# usage: [PROG=tools/prof_2d.py] tools/pmc.sh <outdir-under-gpurun_out> [extra args of the driver]
# (driver defaults to tools/prof_op.py --iters 6)
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- \
      python "$GRAFT_REPO_ROOT/$PROG" "${PRE[@]}" "${EXTRA[@]}" > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
EXTRA=("$@")
PROG=${PROG:-tools/prof_op.py}
PRE=(); [ "$PROG" = tools/prof_op.py ] && PRE=(--iters 6)
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
find "$out" -name "*.csv" | head -20
