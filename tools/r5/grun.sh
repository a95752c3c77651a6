#!/bin/bash
# tools/r5/grun.sh TIMEOUT 'command': gpurun with retries while no box / slot is free (rc 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
