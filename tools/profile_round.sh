#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace stats of the bench command and of the grouped
# head-stage forward, and the HBM traffic counters of the latter (separate --pmc passes, no tracing, as the
# guide prescribes).
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/warm.log 2>&1   # fills MIOpen's find-db
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fwd -o fwd -- python3 $GRAFT_REPO_ROOT/tools/run_group.py 2 30 > $OUT/fwd.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group.py 2 5 > $OUT/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group.py 2 5 > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -40
