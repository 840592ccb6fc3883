#!/usr/bin/env python
"""From a rocprofv3 kernel trace of bench.py: per-kernel average durations, the idle gaps between consecutive
kernels of the timed loop and the per-step period.  usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows = [r for r in rows if "rk::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    short = lambda n: n.split("(")[0].replace("void ", "")[:60]   # noqa: E731
    gaps, durs = {}, {}
    for a, b in zip(rows, rows[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g < 50_000:                      # same burst of the loop
            gaps.setdefault((short(a["Kernel_Name"]), short(b["Kernel_Name"])), []).append(g)
    for r in rows:
        durs.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in durs.items():
        v.sort()
        print("%-62s n=%4d  median %8.2f us  mean %8.2f us" % (k, len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
    for k, v in gaps.items():
        v.sort()
        print("gap %-40s -> %-40s n=%4d median %6.2f us mean %6.2f us" % (k[0][-40:], k[1][-40:], len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
    fw = [r for r in rows if "interp" in r["Kernel_Name"]]
    per = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(fw, fw[1:])]
    per = sorted(p for p in per if p < 1_000_000)
    if per:
        print("forward-to-forward period: median %.2f us (n=%d)" % (per[len(per) // 2] / 1e3, len(per)))


if __name__ == "__main__":
    main()
