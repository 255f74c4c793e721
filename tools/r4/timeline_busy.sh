#!/bin/bash
# GPU busy fraction and kernel concurrency inside the timed region of the default (two-stream) bench command, from a rocprofv3 kernel
# trace.  Run through gpurun: tools/gr.sh <dir> 900 'bash tools/r4/timeline_busy.sh'   ($O = gpurun_out/<dir>)
cd /tmp && export TMPDIR=/tmp
K=6; W=2
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K --warmup $W --cpu-budget-s 1 > $GRAFT_REPO_ROOT/$O/bench_traced.json 2>/dev/null
cd $GRAFT_REPO_ROOT
K=$K W=$W python3 - <<'PY'
import csv, glob, os, collections
K, W = int(os.environ['K']), int(os.environ['W'])
f = glob.glob(os.environ['O'] + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
parts = [r[0] for r in rows if 'raster_partition' in r[2]]
# the W + K back-to-back steps are the densest run of W + K partition launches; the window is [first timed step, last timed step)
n = W + K
i0 = min(range(len(parts) - n + 1), key=lambda i: parts[i + n - 1] - parts[i])
tb, te, NS = parts[i0 + W], parts[i0 + n - 1], K - 1
sel = [(max(s, tb), min(e, te), nme) for s, e, nme in rows if e > tb and s < te]
busy, gaps = 0, []
cs, ce = sel[0][0], sel[0][1]
for s, e, nme in sel[1:]:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce, nme)); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = te - tb
print(f'{NS} timed steps: span {span / 1e6:.2f} ms = {span / NS / 1e6:.2f} ms per step; some kernel running {busy / span:.4f} of it; idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps')
short = lambda nme: nme.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
print('largest gaps (us, next kernel):', sorted(((round(g / 1e3, 1), short(nme)) for g, nme in gaps), reverse=True)[:6])
ev = []
for s, e, nme in sel:
    ev += [(s, 1), (e, -1)]
ev.sort()
c, last, t1, t2 = 0, ev[0][0], 0, 0
for t, d in ev:
    if c >= 2: t2 += t - last
    elif c == 1: t1 += t - last
    last = t; c += d
print(f'one kernel resident {t1 / NS / 1e6:.2f} ms per step, two or more {t2 / NS / 1e6:.2f} ms per step')
dur, cnt = collections.Counter(), collections.Counter()
for s, e, nme in sel:
    dur[short(nme)] += e - s; cnt[short(nme)] += 1
print(f'sum of kernel durations {sum(dur.values()) / NS / 1e6:.2f} ms per step (kernels of the two streams overlap: durations are not additive)')
for k, v in dur.most_common(16):
    print(f'  {k:46s} {v / NS / 1e6:7.3f} ms per step  x{cnt[k] / NS:.1f}')
PY
