#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run's counter_collection.csv (any program):  python tools/pmc_positions.py <csv> [...]
Several files (one per --pmc pass) are merged by kernel."""
import collections, csv, sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        short = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        out[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(out.items()):
    if max(len(v) for v in d.values()) < 8:
        continue
    print(k, "x%d" % max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        print("    %-24s %14.0f" % (c, sum(v) / len(v)))
