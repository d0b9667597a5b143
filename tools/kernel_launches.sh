#!/bin/bash
# per-launch durations of one kernel (name substring $1) over the last steady step of the config-5 training bench; run through gpurun
OUT=/tmp/kl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config ${2:-serial} --steps 3 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-inference-leg > $OUT/log 2>&1
python3 - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/kl/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if sys.argv[1] in r['Kernel_Name']]
n = len(sel) // 5 if len(sel) >= 5 else len(sel)
for r in sel[-n:]:
    print('%8.1f us  grid %s  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', '?'), r['Kernel_Name'][:60]))
PY
