OUT=/tmp/sinf; rm -rf $OUT; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o infer -- python3 $GRAFT_REPO_ROOT/bench.py --config serial --mode infer --dtype bf16 --imgs-per-gpu 8 --graph 0 --steps 10 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline > $OUT/infer.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_steady_stats.py $OUT/infer_kernel_trace.csv multiclass_soft_select 1 45
