#!/bin/bash
# per-launch durations of the bookkeeping kernels of one training step (rocprofv3 kernel trace); run through gpurun
OUT=/tmp/skh; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --windows 1 --no-inference-leg --no-cpu-baseline --no-roofline > $OUT/log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/skh/t_kernel_trace.csv')))
# last step: between the last two multi_clip_adam launches
adam = [i for i, r in enumerate(rows) if 'multi_clip_adam' in r['Kernel_Name']]
lo, hi = adam[-2], adam[-1]
sel = collections.defaultdict(list)
for r in rows[lo:hi]:
    n = r['Kernel_Name']
    for key in ('conv1x1_sum', 'pad_rows2', 'conv3x3_wsum', 'bn_fold_finish', 'relu_sum_bwd', 'bias_act_kernel'):
        if key in n:
            sel[key].append(((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) if 'Grid_Size_X' in r else 0))
for k, v in sel.items():
    v.sort(reverse=True)
    print(k, 'n=%d total=%.0f us' % (len(v), sum(d for d, _ in v)), 'top:', ' '.join('%.1f' % d for d, _ in v[:12]), '... median %.1f' % v[len(v) // 2][0])
PY
