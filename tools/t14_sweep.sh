#!/usr/bin/env bash
# [32,8,C,14,14] forward / backward timing at C = 216 and 288 (hipGraph replay).  The RK_T14V tuning variants this script swept in round 6
# (profiles/r06_tile14_variants.txt) are no longer in the tree: only variant 1 (float4 stores) was kept, as the default.
for v in ${T14_VARIANTS:-0}; do
  for c in 216 288; do
    echo -n "RK_T14V=$v C=$c: "; RK_T14V=$v python tools/op3d_graph_time.py 32 8 $c 14 14 2>&1 | tail -1
  done
done
