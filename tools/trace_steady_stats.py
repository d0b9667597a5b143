"""Per-step kernel statistics of the steady state from a rocprofv3 --kernel-trace csv.

MIOpen's find step (cudnn.benchmark) measures candidate solvers -- including its naive reference convolutions -- inside
the first iteration, which swamps `--stats`.  This keeps only the dispatches after the last `naive_conv*` kernel and
after `skip` further step markers, and reports per-step averages.
    python tools/trace_steady_stats.py <kernel_trace.csv> <marker substring> <markers per step> [top]"""
import collections
import csv
import sys

path, marker, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last_find = max([i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('naive_conv')] or [-1])
rows = rows[last_find + 1:]
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
if marker.startswith('~'):      # '~name': ONE dispatch of a kernel whose name contains `name` closes each step
    marks = [i for i, r in enumerate(rows) if marker[1:] in r['Kernel_Name']]
    per_step = int(sys.argv[3])
steps = len(marks) // per_step - 1
assert steps >= 1, 'not enough steady-state steps in the trace (markers found: %d)' % len(marks)
lo, hi = marks[len(marks) - 1 - steps * per_step], marks[-1]
sel = rows[lo + 1:hi + 1]
span = (int(rows[hi]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])) / steps / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r['Kernel_Name']]
    a[0] += 1
    a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(a[1] for a in agg.values())
print('steady-state steps: %d   wall %.2f ms/step (under the profiler)   kernel time %.2f ms/step   %d launches/step'
      % (steps, span, tot / steps / 1e6, len(sel) / steps))
print('| kernel | launches/step | us/step | % |')
print('|---|---|---|---|')
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('| `%s` | %.1f | %.1f | %.1f |' % (n[:120].replace('|', '/'), c / steps, t / steps / 1e3, 100 * t / tot))
