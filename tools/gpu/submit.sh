#!/bin/bash
# tools/gpu/submit.sh TIMEOUT 'command': gpurun with retries while no GPU slot is free (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
