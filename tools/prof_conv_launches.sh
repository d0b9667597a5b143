#!/bin/bash
# per-launch workgroup counts and durations of the dense convolution kernels in one steady training step (run on the GPU box)
OUT=/tmp/convl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/convl/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: after the last-but-one optimizer kernel
opt = [i for i, r in enumerate(rows) if 'multi_clip_adam' in r['Kernel_Name'] or 'FusedOptimizerTensorListMetadata' in r['Kernel_Name']]
starts = [i for j, i in enumerate(opt) if j == 0 or opt[j - 1] != i - 1]
lo, hi = (starts[-2], starts[-1]) if len(starts) >= 2 else (0, len(rows))
agg = collections.OrderedDict()
for r in rows[lo:hi]:
    n = r['Kernel_Name']
    if not any(k in n for k in ('conv_nn', 'conv_nt8', 'conv3x3_patch')): continue
    short = n.split('(')[0].replace('void kgdet::', '')
    wgs = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    key = (short, wgs)
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
print('kernel | workgroups | launches | us each (mean) | us total')
for (k, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-22s %6d  x%2d  %7.1f  %8.1f' % (k, w, len(v), sum(v) / len(v), sum(v)))
PY
