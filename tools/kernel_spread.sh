#!/bin/bash
# min / median / max duration per kernel over the last steady step of the training bench (config $1: kgdet | serial); run through gpurun
OUT=/tmp/ks
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config ${1:-kgdet} --steps 3 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-inference-leg > $OUT/log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/ks/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
key = 'multi_clip_adam' if any('multi_clip_adam' in n for n in names) else 'moment_bbox_backward'
marks = [i for i, n in enumerate(names) if key in n]
per = max(1, len(marks) // 5)          # markers per step (5 steps traced)
marks = marks[per - 1::per]
sel = rows[marks[-2] + 1:marks[-1] + 1] if len(marks) >= 2 else rows
agg = collections.defaultdict(list)
for r in sel:
    agg[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('launches %d, kernel time %.2f ms' % (len(sel), sum(sum(v) for v in agg.values()) / 1e3))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
    v.sort()
    print('%8.1f us  %3d x  min %7.1f  med %7.1f  max %7.1f  %s' % (sum(v), len(v), v[0], v[len(v) // 2], v[-1], n[:70]))
PY
