cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-budget-s 1 > $GRAFT_REPO_ROOT/$O/bench_traced.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ['O'] + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
# the timed region: the last 6 steps = last 6*... take the last 60% of raster_partition launches
parts = [r for r in rows if 'raster_partition' in r[2]]
t_begin = parts[-6][0]
t_end = max(r[1] for r in rows)
sel = [r for r in rows if r[0] >= t_begin]
# union busy time
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
gaps = []
for s, e, n in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - t_begin
print(f'last 6 steps: span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({busy/span:.3f}), idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps')
big = sorted(gaps, reverse=True)[:12]
print('largest gaps (us):', [(round(g/1e3,1), n.split('(')[0][-30:]) for g, n in big])
# sum of kernel durations per class
dur = collections.Counter()
for s, e, n in sel:
    k = n.split('(')[0].replace('(anonymous namespace)::','').replace('void ','')[:40]
    dur[k] += e - s
tot = sum(dur.values())
print(f'sum of kernel durations {tot/1e6:.2f} ms over 6 steps = {tot/6e6:.2f} ms/step (span/step {span/6e6:.2f})')
for k, v in dur.most_common(14): print(f'   {k:42s} {v/6e6:7.3f} ms/step')
PY
