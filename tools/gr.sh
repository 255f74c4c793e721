#!/bin/bash
# gpurun wrapper of the build sessions: tools/gr.sh OUTDIR TIMEOUT 'command'  (creates gpurun_out/OUTDIR on the GPU box first; O=$that dir)
D=$1; T=$2; shift 2
exec /usr/local/graft/bin/gpurun --timeout $T -- "mkdir -p gpurun_out/$D; export O=gpurun_out/$D; $*"
