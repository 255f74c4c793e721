#!/bin/bash
# tools/gr.sh with retries while the pod's GPU slots are busy (gpurun exit code 3): tools/gr_retry.sh OUTDIR TIMEOUT 'command'
for i in $(seq 1 40); do
  tools/gr.sh "$@" > /tmp/gr_last.log 2>&1
  rc=$?
  if ! grep -q "status=transient" /tmp/gr_last.log; then cat /tmp/gr_last.log; exit $rc; fi
  sleep 45
done
cat /tmp/gr_last.log; exit 3
