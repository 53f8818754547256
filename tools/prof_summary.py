#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small text summaries committed under profiles/.

  prof_summary.py stats <dir> <out.md>      kernel_stats.csv  -> per-kernel calls / total / average / share
  prof_summary.py pmc   <dir> <out.md>      counter_collection.csv -> per-kernel, per-counter sum and per-dispatch mean
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits


def short(name):
    name = name.replace("wg::", "")
    return name if len(name) < 90 else name[:87] + "..."


def stats(d, out):
    rows = []
    for f in find(d, "kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            o.write("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |\n" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))


def pmc(d, out):
    agg = defaultdict(lambda: [0, 0.0])
    for f in find(d, "counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            a = agg[(r["Kernel_Name"], r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    with open(out, "w") as o:
        o.write("| kernel | counter | dispatches | sum | mean per dispatch |\n|---|---|---|---|---|\n")
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write("| %s | %s | %d | %.1f | %.3f |\n" % (short(k), c, n, s, s / max(1, n)))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
