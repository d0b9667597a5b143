cd /tmp && export TMPDIR=/tmp
for w in forward grad_offset grad_input; do
rm -rf /tmp/kp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o k -- python3 $GRAFT_REPO_ROOT/tools/run_bwd_phase.py $w 20 > /tmp/kp.log 2>&1
echo "== $w"
python3 - <<'PY'
import csv
tot=0
for r in csv.DictReader(open('/tmp/kp/k_kernel_stats.csv')):
    per=float(r['TotalDurationNs'])/20/1e3
    tot+=per
    if per>2: print('%-70s calls/it %5.1f  us/it %7.1f' % (r['Name'][:70], int(r['Calls'])/20, per))
print('total us/it %.1f' % tot)
PY
done
