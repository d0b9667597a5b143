#!/bin/bash
# kernel averages of one head stage's forward + backward for several experiment builds of the library:
#   bash tools/kstats_variants.sh "<variant> <variant> ..." [B] [filter] [random|trained]     ("-" = the product build)
# (variants are built with `make -C kgdet_amd/csrc VARIANT=<v> EXTRA=-D...` -> kgdet_amd/libkgdet_hip_<v>.so)
B=${2:-2}
F=${3:-dcn_bwd}
MODE=${4:-random}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  if [ "$v" = "-" ]; then unset KGDET_LIB; else export KGDET_LIB=$GRAFT_REPO_ROOT/kgdet_amd/libkgdet_hip_$v.so; fi
  rm -rf /tmp/ksv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksv -o k -- python3 $GRAFT_REPO_ROOT/tools/run_group_bwd.py $B 30 $MODE > /tmp/ksv.log 2>&1
  echo "== $v"
  FILTER=$F python3 - <<'PY'
import csv, os
for r in csv.DictReader(open('/tmp/ksv/k_kernel_stats.csv')):
    if os.environ['FILTER'] in r['Name']:
        print('%-60s %4s  avg %7.1f  min %7.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
