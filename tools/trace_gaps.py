#!/usr/bin/env python3
"""Where a small grid's step goes: from a rocprofv3 --kernel-trace CSV, the kernels of the steady loop in time order, their
durations and the idle time between the end of one and the start of the next.

    python tools/trace_gaps.py <..._kernel_trace.csv> [--dump N]      # --dump N: also N consecutive dispatches from the middle
Prints per kernel of a step the median duration and the median gap that FOLLOWS it (gaps above 100 us -- host pauses between
the phases of the bench -- are left out), and the sums."""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
dur, gap = {}, {}
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g = s1 - e0
    dur.setdefault(short(n0), []).append(e0 - s0)
    if g < 100_000:
        gap.setdefault(short(n0), []).append(g)
tot_d = tot_g = 0.0
for n in dur:
    if len(dur[n]) < 20:
        continue
    d, g = statistics.median(dur[n]) / 1e3, statistics.median(gap.get(n, [0])) / 1e3
    print("%-46s x%-5d duration %7.2f us   gap after %6.2f us" % (n, len(dur[n]), d, g))

if "--dump" in sys.argv:
    n = int(sys.argv[sys.argv.index("--dump") + 1])
    mid = len(ks) // 2
    print("\n%d consecutive dispatches from the middle of the trace (start offset us, duration us, grid):" % n)
    for s0, e0, name in ks[mid:mid + n]:
        print("  +%9.2f  %8.2f  %s" % ((s0 - ks[mid][0]) / 1e3, (e0 - s0) / 1e3, short(name)))
