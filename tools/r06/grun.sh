#!/bin/bash
# usage: tools/r06/grun.sh <timeout_s> <logfile> '<command>'   -- gpurun with retries while the pod's GPU slots are busy (exit code 3)
t=$1; log=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
