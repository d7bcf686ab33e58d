#!/bin/bash
# Run one gpurun command; when the pool has no free box (exit 3: nothing ran, nothing charged) wait and ask again.
# usage: scripts/gpu_try.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
