cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > $GRAFT_REPO_ROOT/$O/bench_s1.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['O'] + '/s1/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total', tot / 6e6, 'ms/step (6 batches)')
for r in rows[:40]:
    print(f"{r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]:60s} {int(r['Calls'])/6:6.1f}/step {float(r['TotalDurationNs'])/6e6:7.3f} ms/step avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
