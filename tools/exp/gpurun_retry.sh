#!/bin/bash
# usage: gpurun_retry.sh <timeout> <logfile> <command...>   - retries while the pod has no free GPU slot (exit 3)
T=$1; L=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $L 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
