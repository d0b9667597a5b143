#!/bin/bash
# kernel stats + fabric counters + SQ counters of one head stage's grouped DeformConv backward (grad_input, grad_offset,
# grad_weight plane kernels), ONE binary, one run each; run through gpurun:  bash tools/profile_bwd_group.sh <outdir-name> [B] [random|trained]
R=${1:-r02bwd}; B=${2:-2}; MODE=${3:-random}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bwdprof /tmp/pmcf /tmp/pmcw /tmp/pmcs1 /tmp/pmcs2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bwdprof -o bwd -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 30 $MODE > $OUT/bwd.log 2>&1
cp /tmp/bwdprof/bwd_kernel_stats.csv $OUT/
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 5 $MODE > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 5 $MODE > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcs1 -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 5 $MODE > $OUT/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmcs2 -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 5 $MODE > $OUT/pmc_sq2.log 2>&1
for d in pmcf pmcw pmcs1 pmcs2; do cp /tmp/$d/p_counter_collection.csv $OUT/$d.csv 2>/dev/null; done
python3 - $OUT <<'PY'
import csv, collections, sys
out = sys.argv[1]
print('## kernel stats (30 iterations)')
for r in csv.DictReader(open(out + '/bwd_kernel_stats.csv')):
    if 'dcn_' in r['Name'] or 'conv1x1' in r['Name']:
        print('| `%s` | %s | %.1f | %.1f | %.1f |' % (r['Name'][:70].replace('|', '/'), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
for f in ('pmcf', 'pmcw', 'pmcs1', 'pmcs2'):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open(out + '/' + f + '.csv')):
            rows[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
    except FileNotFoundError:
        continue
    for k, d in rows.items():
        if 'dcn_' in k:
            print(f, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
