#!/usr/bin/env bash
# tile14 backward tuning variants (RK_T14V bits: 1 wide stores, 2 XCD-contiguous map, 4 four waves per SIMD)
for v in ${T14_VARIANTS:-0 1 8 9}; do
  for c in 216 288; do
    echo -n "RK_T14V=$v C=$c: "; RK_T14V=$v python tools/op3d_graph_time.py 32 8 $c 14 14 2>&1 | tail -1
  done
done
