#!/bin/bash
# kernel trace of a long training loop, reduced on the box to per-step sums over time (gpurun: bash tools/steady_trace.sh <label> [steps])
OUT=/tmp/steadytrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o st -- python3 $GRAFT_REPO_ROOT/tools/steady_loop.py ${2:-500} > $OUT/loop.log 2>&1
tail -6 $OUT/loop.log
python3 $GRAFT_REPO_ROOT/tools/steady_trace_stats.py $OUT/st_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/${1:-r04}_steady_trace.md
cat $GRAFT_REPO_ROOT/gpurun_out/${1:-r04}_steady_trace.md | cut -c1-260
