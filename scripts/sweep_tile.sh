#!/bin/bash
# kernel-only time of the f32 sampler under both builds over a range of batch sizes (cfg2 geometry)
set -e
BS=${BS:-"1024 4096 8192 12288 16384 24576 32768 49152 65536"}
POLS=${POLS:-"throughput latency"}
for b in $BS; do
  for pol in $POLS; do
    echo -n "B=$b $pol " 
    timeout -k 10 100 python bench.py --workload cfg2 --batch $b --tile-policy $pol --kernel-only --no-cpu-baseline 2>/dev/null
  done
done
