#!/bin/bash
# Round-end measurement set: smoke, every BASELINE config once (kernel timing + roofline), and the profile of the default run.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/final
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1 || exit 2
timeout -k 10 300 python bench.py > gpurun_out/final/cfg2.json 2> gpurun_out/final/cfg2.err || exit 3
timeout -k 10 200 python bench.py --workload cfg1 --no-cpu-baseline > gpurun_out/final/cfg1.json 2>/dev/null || exit 4
timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/final/cfg3.json 2>/dev/null || exit 5
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/final/cfg4.json 2>/dev/null || exit 6
timeout -k 10 200 python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/final/cfg5_f32.json 2>/dev/null || exit 7
timeout -k 10 200 python bench.py --workload cfg5 --dtype bf16 --no-cpu-baseline > gpurun_out/final/cfg5_bf16.json 2>/dev/null || exit 8
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/final/cfg2_bf16.json 2>/dev/null || exit 9
timeout -k 10 200 python bench.py --rng partitionable --no-cpu-baseline > gpurun_out/final/cfg2_keyed.json 2>/dev/null || exit 11
timeout -k 10 300 python bench.py --workload cfg4 --rng partitionable --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/final/cfg4_keyed.json 2>/dev/null || exit 12
timeout -k 10 300 python bench.py --workload cfg4 --select fused --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/final/cfg4_select_fused.json 2>/dev/null || exit 13
timeout -k 10 300 python bench.py --workload cfg4 --select separate --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/final/cfg4_select_separate.json 2>/dev/null || exit 14
timeout -k 10 300 python bench.py --workload cfg3 --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/final/cfg3_bf16.json 2>/dev/null || exit 16
timeout -k 10 300 python bench.py --workload cfg4 --dtype bf16 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/final/cfg4_bf16.json 2>/dev/null || exit 17
timeout -k 10 200 bench/native_bench > gpurun_out/final/native_cfg2.json 2>/dev/null || exit 15
echo done
