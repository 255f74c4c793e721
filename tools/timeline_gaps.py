#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 kernel trace of bench.py: every interval > argv[2] us (default 30) in which no kernel ran,
with the kernels that end before / start after it; and the busy fraction per 50 ms slice.  usage: timeline_gaps.py trace.csv [us]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
rows.sort()
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 30e3
t0 = rows[0][0]
cur_e, last = rows[0][1], rows[0][2]
gaps = []
for s, e, n in rows[1:]:
    if s > cur_e + thr:
        gaps.append(((cur_e - t0) / 1e6, (s - cur_e) / 1e3, last, n))
    if e > cur_e:
        cur_e, last = e, n
print(f'{len(gaps)} gaps > {thr / 1e3:.0f} us; trace {(cur_e - t0) / 1e6:.1f} ms')
for at, us, a, b in gaps[-60:]:
    print(f'  at {at:9.2f} ms: idle {us:8.1f} us   after {a}   before {b}')
