#!/bin/bash
# kernel averages of one head stage's forward + backward (rocprofv3 --kernel-trace --stats), for the library in $KGDET_LIB
# (default: the product); run through gpurun:  bash tools/kstats_bwd.sh [B]
B=${1:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksb -o k -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 30 > /tmp/ksb.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/ksb/k_kernel_stats.csv')):
    if 'dcn_' in r['Name']:
        print('%-60s %4s  avg %7.1f  min %7.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
