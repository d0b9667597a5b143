#!/bin/bash
# kernel stats + HBM counters of the grouped head-stage forward (the bench's roofline kernel); run through gpurun
R=${1:-r01fwd}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fwdprof -o fwd -- python3 $GRAFT_REPO_ROOT/tools/run_group.py 2 30 > $OUT/fwd.log 2>&1
cp /tmp/fwdprof/fwd_kernel_stats.csv $OUT/
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group.py 2 5 > $OUT/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw -o p -- python3 $GRAFT_REPO_ROOT/tools/run_group.py 2 5 > $OUT/pmc_write.log 2>&1
cp /tmp/pmcf/p_counter_collection.csv $OUT/fetch.csv; cp /tmp/pmcw/p_counter_collection.csv $OUT/write.csv
head -6 $OUT/fwd_kernel_stats.csv | cut -c1-200
