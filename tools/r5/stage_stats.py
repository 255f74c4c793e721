#!/usr/bin/env python3
"""Steady-state kernel statistics of a `rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format csv` run of bench.py
with LANEMAP_ROCTX=1: only kernels that START inside the `timed_steps` roctx range are counted (warm-up, the priming batch, the
post-clock checkers and the instrumented roofline pass are outside it), per kernel name and per stage (the innermost roctx range that
was open on the launching thread when the kernel's hipLaunchKernel / hipModuleLaunchKernel / graph launch was issued, matched through the
correlation id; kernels replayed from a HIP graph carry the graph launch's id).

usage: stage_stats.py <rocprof output dir> <steps> [top N]"""
import bisect
import collections
import csv
import glob
import os
import sys


def load(pattern, d):
    f = glob.glob(os.path.join(d, '**', pattern), recursive=True)
    if not f:
        return None
    with open(f[0]) as fh:
        return list(csv.DictReader(fh))


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '')[:64]


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
    kern = load('*kernel_trace.csv', d)
    mark = load('*marker_api_trace.csv', d)
    hip = load('*hip_api_trace.csv', d)
    if kern is None or mark is None:
        raise SystemExit(f'{d}: kernel trace or marker trace missing (run with --kernel-trace --marker-trace)')
    name_col = 'Function' if 'Function' in mark[0] else next(c for c in mark[0] if 'name' in c.lower() or 'message' in c.lower())
    timed = [m for m in mark if m[name_col].strip('"') == 'timed_steps']
    if not timed:
        raise SystemExit('no timed_steps range in the marker trace (LANEMAP_ROCTX=1?)')
    t0, t1 = int(timed[0]['Start_Timestamp']), int(timed[0]['End_Timestamp'])
    # per thread: sorted list of (start, end, name) of the stage ranges
    per_thread = collections.defaultdict(list)
    for m in mark:
        n = m[name_col].strip('"')
        if n == 'timed_steps':
            continue
        per_thread[m['Thread_Id']].append((int(m['Start_Timestamp']), int(m['End_Timestamp']), n))
    for v in per_thread.values():
        v.sort()
    launch = {}
    if hip is not None:
        for h in hip:
            if 'Launch' in h['Function']:
                launch[h['Correlation_Id']] = (h['Thread_Id'], int(h['Start_Timestamp']))

    def stage_of(k):
        rec = launch.get(k['Correlation_Id'])
        if rec is None:
            return '(no launch record)'
        tid, ts = rec
        best, best_len = '(outside the stage ranges)', None
        rs = per_thread.get(tid, [])
        i = bisect.bisect_right(rs, (ts, float('inf'), ''))
        for s, e, n in rs[max(0, i - 64):i]:
            if s <= ts <= e and (best_len is None or e - s < best_len):
                best, best_len = n, e - s
        return best

    by_name, by_stage = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
    n_in = 0
    for k in kern:
        s, e = int(k['Start_Timestamp']), int(k['End_Timestamp'])
        if not (t0 <= s <= t1):
            continue
        n_in += 1
        a = by_name[short(k['Kernel_Name'])]
        a[0] += 1; a[1] += e - s
        b = by_stage[stage_of(k)]
        b[0] += 1; b[1] += e - s
    tot = sum(v[1] for v in by_name.values())
    print(f'# steady state only: {n_in} kernel dispatches started inside the timed_steps range ({(t1 - t0) / 1e6:.1f} ms, {steps} steps); '
          f'{len(kern) - n_in} dispatches outside it are not counted')
    print(f'total {tot / steps / 1e6:.3f} ms of kernel time per step; wall {(t1 - t0) / steps / 1e6:.3f} ms per step')
    wino = sum(v[1] for n, v in by_name.items() if n.startswith('wino44_kernel'))
    print(f'wino44_kernel {wino / steps / 1e6:.3f} ms/step, every other kernel {(tot - wino) / steps / 1e6:.3f} ms/step')
    copies = sum(v[0] for n, v in by_name.items() if 'copyBuffer' in n or 'Memcpy' in n)
    aten = sum(v[0] for n, v in by_name.items() if n.startswith('at::'))
    print(f'copy kernels {copies / steps:.1f} per step, ATen kernels {aten / steps:.1f} per step')
    print('## per kernel')
    for n, (c, ns) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{n:64s} {c / steps:7.1f}/step {ns / steps / 1e6:8.3f} ms/step avg {ns / c / 1e3:8.1f} us')
    print('## per stage (roctx range open on the launching thread)')
    for n, (c, ns) in sorted(by_stage.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:40s} {c / steps:7.1f} launches/step {ns / steps / 1e6:8.3f} ms/step')


if __name__ == '__main__':
    main()
