cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/aq -o model -- python $GRAFT_REPO_ROOT/tools/prof_model.py --tier large --variant rubiks3d-aq --amp bf16 --steps 4 > /tmp/aq.log 2>&1
f=$(find /tmp/aq -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); lo = n * 3 // 4     # last step
last = rows[lo:]
prev = None
for i, r in enumerate(last):
    nm = r["Kernel_Name"]
    if nm.startswith("void rk::") or nm.startswith("rk::"): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < 8: continue
    print(f"{i:5d} {d:8.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {nm[:90]}")
PY
