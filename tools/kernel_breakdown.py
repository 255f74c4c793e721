#!/usr/bin/env python3
"""Prints the per-step time of every kernel from a rocprofv3 --stats kernel_stats.csv of bench.py (usage: kernel_breakdown.py csv batches)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nb = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r['Name']
    if 'rocclr' in n or 'Cijk' in n or 'at::native' in n:
        continue
    print(f"{n[:72]:72s} {int(r['Calls']) / nb:5.1f}/step {int(r['TotalDurationNs']) / nb / 1e6:7.3f} ms/step")
