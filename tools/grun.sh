#!/bin/bash
# local helper: run a gpurun call, retrying while the pod has no free GPU slot (exit code 3: nothing charged)
T=${1:-900}; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
