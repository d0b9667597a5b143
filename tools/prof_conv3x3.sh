cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/k3 -o k3 -- python3 $GRAFT_REPO_ROOT/tools/bench_conv1x1.py k3 nofind > /tmp/k3.log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/k3/k3_kernel_trace.csv')))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'conv_n' not in n and 'wsum' not in n and 'conv1x1_sum' not in n: continue
    key = (n[:40], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    v.sort()
    print(k, len(v), 'median %.1f us' % v[len(v)//2])
PY
tail -5 /tmp/k3.log
