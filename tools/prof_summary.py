#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small text summaries committed under profiles/.

  prof_summary.py stats <dir> <out.md>      kernel_stats.csv  -> per-kernel calls / total / average / share
  prof_summary.py pmc   <dir> <out.md>      counter_collection.csv -> per-kernel, per-counter sum and per-dispatch mean
  prof_summary.py phases <dir> <out.md> [period]   kernel_trace.csv of a bench.py run (device-side RunMultipleTimes loop,
                                            runMs(10) chunks) -> per kernel, mean duration by simulated ms modulo the
                                            dissemination period (Handel: every node disseminates in the same ms)
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


def phases(d, out, period=20, chunk=10):
    rows = []
    for f in find(d, "kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ch, j = -1, -1
    agg = defaultdict(lambda: [[0, 0.0] for _ in range(period)])
    tot = defaultdict(float)
    ms_at = [0] * period  # simulated ms seen at each phase (a kernel launched twice per ms must not halve the ALL row)
    for r in rows:
        n = short(r["Kernel_Name"]).split("(")[0]
        if "k_chunk_begin" in n:
            ch, j = ch + 1, -1
        if ch < 0:
            continue
        first = ("k_scan1<" in n or "k_scan<" in n) and "ExpandF" in n  # the first kernel of a simulated ms
        if first:
            j += 1
        t = chunk * ch + max(j, 0)
        if first:
            ms_at[t % period] += 1
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[n][t % period]
        a[0] += 1
        a[1] += dur
        tot[n] += dur
    with open(out, "w") as o:
        o.write("mean us per launch by (simulated ms mod %d); last column: total ms\n\n" % period)
        o.write("| kernel | " + " | ".join(str(p) for p in range(period)) + " | total ms |\n|---|" + "---|" * (period + 1) + "\n")
        for n in sorted(tot, key=lambda k: -tot[k]):
            o.write("| %s | " % n + " | ".join("%.0f" % (a[1] / max(1, a[0]) / 1e3) for a in agg[n]) + " | %.1f |\n" % (tot[n] / 1e6))
        # ALL: device time per simulated ms at this phase = every launch's duration / the ms counted there (the column sum
        # where every kernel launches once per ms; k_scatter / k_col_reserve_end launch twice and count twice)
        o.write("| ALL (us per simulated ms) | " + " | ".join("%.0f" % (sum(agg[n][p][1] for n in agg) / max(1, ms_at[p]) / 1e3)
                                                              for p in range(period)) + " | %.1f |\n" % (sum(tot.values()) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "phases":
        phases(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 20)
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
