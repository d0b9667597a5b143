#!/bin/bash
# min / median / max duration per kernel over the last batch of the bf16 batch-8 inference bench; run through gpurun
OUT=/tmp/ksi
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode infer --dtype bf16 --imgs-per-gpu 8 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline > $OUT/log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/ksi/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'multiclass_nms' in r['Kernel_Name']]
sel = rows[marks[-2] + 1:marks[-1] + 1]
agg = collections.defaultdict(list)
for r in sel:
    agg[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3
print('launches %d, kernel time %.2f ms, span %.2f ms' % (len(sel), sum(sum(v) for v in agg.values()) / 1e3, span / 1e3))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:60]:
    v.sort()
    print('%8.1f us  %3d x  min %7.1f  med %7.1f  max %7.1f  %s' % (sum(v), len(v), v[0], v[len(v) // 2], v[-1], n[:90]))
PY
