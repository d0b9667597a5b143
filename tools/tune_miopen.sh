#!/bin/bash
# Regenerate kgdet_amd/miopen_db (run on the GPU box through gpurun): every bench workload measures MIOpen's solvers
# into its own directory; the records are copied to gpurun_out/miopen_db, from where they are committed.
cd $GRAFT_REPO_ROOT
rm -rf kgdet_amd/miopen_db/*/
run() { (time python3 bench.py "$@" --no-cpu-baseline --no-roofline) 2>&1 | grep -E "^real|value" | cut -c1-200; }
run
run --mode infer --dtype bf16 --imgs-per-gpu 8
run --mode infer --dtype fp32 --imgs-per-gpu 8
find kgdet_amd/miopen_db -name '*.time' -delete -o -name '*.lock' -delete
mkdir -p gpurun_out/miopen_db && cp -rv kgdet_amd/miopen_db/*/ gpurun_out/miopen_db/
echo "second pass (lookups only)"
run
run --mode infer --dtype bf16 --imgs-per-gpu 8
run --mode infer --dtype fp32 --imgs-per-gpu 8
