#!/usr/bin/env bash
# usage: tools/sweep_env.sh "<shapes>" "VAR=val VAR=val" ["VAR=val ..." ...]  -- runs tools/shape_sweep.py once per env set
shapes=$1; shift
for e in "$@"; do
  echo "== $e"
  ( for kv in $e; do export "$kv"; done; timeout 150 python tools/shape_sweep.py $shapes; echo "rc=$?" )
done
