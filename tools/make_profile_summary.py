#!/usr/bin/env python
"""Condense rocprofv3 outputs under gpurun_out/ into small tracked files under profiles/.

    python tools/make_profile_summary.py <tag> <kernel-stats-dir | -> [<pmc-dir>]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.strip('"')
    if name.startswith("void (anonymous namespace)::"):
        name = name.replace("(anonymous namespace)::", "rk::", 1)
    if "rk::" in name:
        return name.split("(")[0].replace("void ", "")
    if "distribution_elementwise" in name:
        return "at::native uniform_ fill (bench input generation)"
    return name.split("(")[0][:80]


def main():
    tag, kdir = sys.argv[1], sys.argv[2]
    pdir = sys.argv[3] if len(sys.argv) > 3 else None
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if kdir != "-":
        stats = glob.glob(os.path.join(kdir, "*kernel_stats.csv"))[0]
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
    bj = os.path.join(kdir, "bench.json")
    if kdir != "-" and os.path.exists(bj) and os.path.getsize(bj):
        with open(os.path.join(out, "%s_bench_under_rocprof.json" % tag), "w") as f:
            f.write(open(bj).read())
    if pdir:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for name in ("sq1", "sq2", "fetch", "write", "grbm"):
            fs = glob.glob(os.path.join(pdir, name, "*counter_collection.csv"))
            if not fs:
                continue
            for r in csv.DictReader(open(fs[0])):
                if "rk::" in r["Kernel_Name"]:
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(out, "%s_pmc.csv" % tag), "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Counter", "MeanPerDispatch", "Dispatches"])
            for k in sorted(agg):
                for c in sorted(agg[k]):
                    v = agg[k][c]
                    w.writerow([k, c, "%.1f" % (sum(v) / len(v)), len(v)])
        # HBM bytes per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB, collected in
        # separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read.
        traffic = {}
        for k in agg:
            if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
                fk = sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"])
                wk = sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"])
                traffic[k] = {"fetch_size_kb": fk, "write_size_kb": wk, "hbm_bytes": (2 * fk + wk) * 1024,
                              "note": "2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, KB -> bytes"}
        with open(os.path.join(out, "%s_traffic.json" % tag), "w") as f:
            json.dump(traffic, f, indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
