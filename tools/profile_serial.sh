OUT=/tmp/serprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o train -- python3 $GRAFT_REPO_ROOT/bench.py --config serial --steps 4 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline > $OUT/train.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_steady_stats.py $OUT/train_kernel_trace.csv '~moment_bbox_backward' 10 40; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/miopen_db; cp -r $GRAFT_REPO_ROOT/kgdet_amd/miopen_db/serial_train_fp32_b2 $GRAFT_REPO_ROOT/gpurun_out/miopen_db/ 2>/dev/null; grep -c moment_bbox_backward $OUT/train_kernel_trace.csv
