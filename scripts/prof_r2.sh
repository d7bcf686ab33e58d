#!/bin/bash
# Round-2 profiling recipe (run on the GPU box via gpurun): scripts/prof_r2.sh <tag> <bench.py args...>
#   kernel trace + stats, then separate PMC passes (never combined with a trace), then a trimmed summary of the frl_*
#   kernels under gpurun_out/prof_<tag>/summary/ (kernel_stats.csv, pmc_summary.json) -> copy that into profiles/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/summary
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $*"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/pmc_lds.log 2>&1 || exit 7
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
summ = collections.defaultdict(dict)
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'frl' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in acc.items():
        summ[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(summ, open(out + '/summary/pmc_summary.json', 'w'), indent=1, sort_keys=True)
rows = []
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    rd = list(csv.reader(open(f)))
    rows = [rd[0]] + [r for r in rd[1:] if 'frl' in r[0]]
with open(out + '/summary/kernel_stats.csv', 'w', newline='') as fh:
    csv.writer(fh).writerows(rows)
for r in rows: print([x[:70] for x in r[:4]])
for k in sorted(summ):
    d = summ[k]
    g = lambda c: d.get(c, {}).get("mean", float('nan'))
    print(f"{k}: mfma_busy/busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('SQ_BUSY_CYCLES'), 1):.3f} valu/mfma {g('SQ_INSTS_VALU') / max(g('SQ_INSTS_MFMA'), 1):.2f} "
          f"hbm_rd_MB {2 * g('FETCH_SIZE') / 1024:.1f} hbm_wr_MB {g('WRITE_SIZE') / 1024:.1f} lds_conflict/active {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f}")
PY
grep "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/summary/bench_line.json
