#!/bin/bash
# Kernel-trace stats of the bf16 batch-8 inference bench (run on the GPU box through gpurun).
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --mode infer --dtype bf16 --imgs-per-gpu 8 --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/warm_infer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infer -o infer -- python3 $GRAFT_REPO_ROOT/bench.py --mode infer --dtype bf16 --imgs-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/infer.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/infer_phases.py 8 bf16 > $OUT/infer_phases.log 2>&1
ls -R $OUT | head
