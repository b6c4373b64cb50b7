#!/usr/bin/env python3
"""Dev tool: top kernels of a rocprofv3 --stats kernel_stats.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls={r['Calls']:>7s} avg={float(r['AverageNs'])/1e3:8.1f}us {r['Name'][:90]}")
