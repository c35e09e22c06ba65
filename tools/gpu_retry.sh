#!/bin/bash
# gpurun with a wait for a free slot: exit code 3 means "no box or slot free right now, nothing charged" -- only that case is retried.
# usage: tools/gpu_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
