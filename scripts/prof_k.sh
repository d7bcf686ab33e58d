#!/bin/bash
# PMC passes over a kernel-only bench run: scripts/prof_k.sh <tag> <bench args...>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 bench.py --kernel-only --steps 10 $*"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/pmc_lds.log 2>&1 || exit 5
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
summ = {}
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'frl_sampler' in r['Kernel_Name'] or 'frl_bwd' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items():
        summ[k[0] + ' ' + k[1]] = sum(v) / len(v)
for k in sorted(summ): print(f"{k:70s} {summ[k]:.4g}")
json.dump(summ, open(out + '/pmc_summary.json', 'w'), indent=1)
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.reader(open(f)))[:4]: print([x[:50] for x in r[:6]])
PY
