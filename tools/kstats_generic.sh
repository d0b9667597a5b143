#!/bin/bash
# kernel averages (rocprofv3 --kernel-trace --stats) of `python3 <script> <args...>`, lines matching $PATTERN (default dcn_):
#   PATTERN=psroi bash tools/kstats_generic.sh tools/time_psroi.py 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksg -o k -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/ksg.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/ksg/k_kernel_stats.csv')):
    if '${PATTERN:-dcn_}' in r['Name']:
        print('%-64s %5s  avg %8.1f  min %8.1f' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
