#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <counters...> -- <python args...>   (run on the GPU box)
# one rocprofv3 --pmc pass (counters only, no tracing) and a per-kernel average table
name=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $out -o pmc -- python3 "$@" > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in rows.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'n=%d' % len(next(iter(d.values()))))
PY
