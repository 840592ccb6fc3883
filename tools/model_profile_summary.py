#!/usr/bin/env python
"""Steady-state per-kernel table of a RubiksNet train step from a rocprofv3 kernel trace of tools/prof_model.py.
The first steps of a fresh process are dominated by MIOpen's find mode (naive_conv_* reference kernels, solver
trials), so only the dispatches of the LAST `steps` train steps are aggregated: the trace is cut at the start of
the (n - steps)-th optimizer kernel burst, found through the periodic `multi_tensor_apply` (Adam) kernels.

    python tools/model_profile_summary.py <kernel_trace.csv> <out.csv> [steps=4]
"""
import collections
import csv
import sys


def short(name):
    name = name.strip('"')
    if name.startswith("void (anonymous namespace)::") or name.startswith("(anonymous namespace)::"):
        # (a non-template kernel demangles without its return type: these rows used to collapse into an empty name)
        name = name.replace("(anonymous namespace)::", "rk::", 1)
    if "rk::" in name:
        return name.split("(")[0].replace("void ", "")
    for key in ("multi_tensor_apply", "elementwise_kernel", "reduce_kernel", "Cijk_", "igemm", "naive_conv", "batched_transpose",
                "SubTensorOpWithScalar", "MIOpen", "gemm", "vectorized_elementwise"):
        if key in name:
            return key + " (" + name.split("(")[0][-40:] + ")" if key in ("elementwise_kernel", "vectorized_elementwise") else key
    return name.split("(")[0][-70:]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [int(r["Start_Timestamp"]) for r in rows if "multi_tensor_apply" in r["Kernel_Name"]]
    # bursts of Adam kernels = one per step
    bursts = [adam[0]] if adam else []
    for a, b in zip(adam, adam[1:]):
        if b - a > 2_000_000:
            bursts.append(b)
    if len(bursts) <= steps:
        raise SystemExit("not enough steps in the trace")
    t0 = bursts[-steps - 1]
    t1 = bursts[-1]
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[short(r["Kernel_Name"])]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    total = sum(v[0] for v in agg.values())
    with open(sys.argv[2], "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "CallsPerStep", "UsPerStep", "PctOfKernelTime"])
        for k, (ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            w.writerow([k, "%.1f" % (n / steps), "%.1f" % (ns / steps / 1e3), "%.2f" % (100.0 * ns / total)])
        w.writerow(["TOTAL kernel time per step (us); wall per step (us)", "", "%.1f" % (total / steps / 1e3),
                    "%.1f" % ((t1 - t0) / steps / 1e3)])
    print("steady-state steps: %d, kernel time %.2f ms/step, wall %.2f ms/step" % (steps, total / steps / 1e6, (t1 - t0) / steps / 1e6))


if __name__ == "__main__":
    main()
