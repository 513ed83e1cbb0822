#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  Usage: tools/gpu_retry.sh <timeout_s> <log> <command...>
T=$1; LOG=$2; shift 2
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$*" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3
