#!/bin/bash
# steady-state kernel table of the training bench (run on the GPU box through gpurun); prints it, keeps no trace
OUT=/tmp/trainprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o train -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --windows 1 --no-inference-leg --no-cpu-baseline --no-roofline --no-exact-leg --graph-train 0 > $OUT/train.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/${1:-r01t}
python3 $GRAFT_REPO_ROOT/tools/trace_steady_stats.py $OUT/train_kernel_trace.csv multi_clip_adam 1 ${2:-70} > $GRAFT_REPO_ROOT/gpurun_out/${1:-r01t}/train_steady.md
head -60 $GRAFT_REPO_ROOT/gpurun_out/${1:-r01t}/train_steady.md | cut -c1-150
