#!/bin/bash
# kernel averages of the large-map DeformConv forward (tools/time_large_fwd.py) under rocprofv3; run through gpurun
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksl -o k -- python3 $GRAFT_REPO_ROOT/tools/time_large_fwd.py ${1:-split} > /tmp/ksl.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/ksl/k_kernel_stats.csv')):
    if 'dcn_' in r['Name'] or 'kgdet' in r['Name']:
        print('%-64s %4s  avg %7.1f  min %7.1f max %7.1f' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
