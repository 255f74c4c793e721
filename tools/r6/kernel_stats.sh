#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (headline workload; --no-second-line: the second line is a child process of its own)
# -> gpurun_out/$1/{default,streams1}_kernel_stats.csv + the bench lines of the profiled runs
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_kstats}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-line > $O/bench_default_profiled.json 2>> $O/prof_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-line --streams 1 --no-graphs > $O/bench_streams1_profiled.json 2>> $O/prof_stderr.log
cp $(find $O/prof_default -name "*kernel_stats.csv" | head -1) $O/default_kernel_stats.csv
cp $(find $O/prof_s1 -name "*kernel_stats.csv" | head -1) $O/streams1_kernel_stats.csv
rm -rf $O/prof_default $O/prof_s1
head -8 $O/streams1_kernel_stats.csv
tail -c 600 $O/bench_streams1_profiled.json
