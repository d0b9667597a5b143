#!/bin/bash
# kernel-level timings of tools/bench_conv1x1.py (run on the GPU box through gpurun); prints the summary, keeps no trace
OUT=/tmp/c1prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o c1 -- python3 $GRAFT_REPO_ROOT/tools/bench_conv1x1.py nofind $1 > $OUT/log.txt 2>&1
grep "^C=" $OUT/log.txt | cut -c1-220
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/c1prof/c1_kernel_trace.csv')))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'conv1x1' in n or 'conv_nn' in n:
        agg.setdefault((n.split('(')[0][-24:], r['Grid_Size_X']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print('%-26s grid %8s  n=%4d  median %7.1f us' % (k[0], k[1], len(v), v[len(v) // 2]))
PY
