#!/bin/bash
# Steady-state kernel table of the bf16 batch-8 inference bench (run on the GPU box through gpurun; keeps no trace).
# The eager loop is profiled (--graph 0): a graph replay shows up as one opaque launch.
R=${1:-r01i}
OUT=/tmp/inferprof
rm -rf $OUT; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/$R
cd /tmp && export TMPDIR=/tmp
# the 1x1 convolutions' routes are measured once per shape (backbone._gemm_choice); the tracer shifts those timings, so an
# un-profiled run records the choices first and the profiled process reads them
export KGDET_GEMM_CHOICES=$OUT/gemm_choices.json
python3 $GRAFT_REPO_ROOT/bench.py --mode infer --dtype bf16 --imgs-per-gpu 8 --graph 0 --steps 3 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline > $OUT/choices.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT -o infer -- python3 $GRAFT_REPO_ROOT/bench.py --mode infer --dtype bf16 --imgs-per-gpu 8 --graph 0 --steps 10 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline > $OUT/infer.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_steady_stats.py $OUT/infer_kernel_trace.csv multiclass_select 1 ${2:-60} > $GRAFT_REPO_ROOT/gpurun_out/$R/infer_steady.md
head -12 $GRAFT_REPO_ROOT/gpurun_out/$R/infer_steady.md | cut -c1-150
