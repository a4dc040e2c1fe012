#!/usr/bin/env python3
"""Print the engine's kernels from a rocprofv3 kernel_stats.csv (name, calls, avg / min / max ms)."""
import csv
import sys

for path in sys.argv[1:]:
    print(path)
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if "gnnvc" not in n:
            continue
        short = n.split("(anonymous namespace)::")[1].split("(")[0]
        print(f"  {short[:64]:64s} calls={r['Calls']:>4} avg={float(r['AverageNs']) / 1e6:8.3f} ms "
              f"min={float(r['MinNs']) / 1e6:8.3f} max={float(r['MaxNs']) / 1e6:8.3f}")
