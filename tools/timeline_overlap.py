#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (p_kernel_trace.csv) of bench.py and prints, for the steady-state part, how much of the wall time
at least one kernel / at least one MFMA-bound kernel (wino_implicit / wino_gemm, conv_mfma) was running, and the per-class sums.
usage: timeline_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
g = [(s, e) for s, e, n in rows if 'wino_gemm' in n or 'wino_implicit' in n]
g0, g1 = g[0][0], max(e for _, e in g)
lo, hi = g0 + 0.3 * (g1 - g0), g0 + 0.9 * (g1 - g0)      # steady state: inside the span of the GEMM launches, past priming / warm-up
rows = [(max(s, lo), min(e, hi), n) for s, e, n in rows if e > lo and s < hi]


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


wall = hi - lo
cls = lambda n: 'mfma' if ('wino_gemm' in n or 'wino_implicit' in n or 'conv_mfma' in n or 'attention_mfma' in n) else ('wino_input' if 'wino_input' in n else 'other')
allu = union([(s, e) for s, e, _ in rows])
mf = union([(s, e) for s, e, n in rows if cls(n) == 'mfma'])
sums = {}
for s, e, n in rows:
    sums[cls(n)] = sums.get(cls(n), 0) + (e - s)
print(f'window {wall / 1e6:.1f} ms: some kernel running {100 * allu / wall:.1f} %, an MFMA-bound kernel running {100 * mf / wall:.1f} %')
print('sum of kernel durations / window:', {k: round(v / wall, 2) for k, v in sums.items()})
