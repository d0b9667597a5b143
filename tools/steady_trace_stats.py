"""Per-STEP kernel time over a long traced run: does the GPU side of the step slow down over the first seconds?
    python tools/steady_trace_stats.py <kernel_trace.csv>
A step ends at `multi_clip_adam`.  Prints, for steps in groups of 25: wall per step (end to end), sum of kernel
durations per step, launches per step, and the durations of a few fixed kernels (per-launch average)."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
last_find = max([i for i, r in enumerate(rows) if r[2].startswith('naive_conv')] or [-1])
rows = rows[last_find + 1:]
marks = [i for i, r in enumerate(rows) if 'multi_clip_adam' in r[2]]
watch = ['conv_nn<1, 4, true>', 'conv_nt8<9>', 'dcn_fwd_plane', 'relu_sum_bwd', 'conv1x1_sum(', 'vectorized_elementwise', 'head_loss_rows<true>']
print('steps traced: %d' % (len(marks) - 1))
G = 25
print('| steps | t since first (s) | wall ms/step | kernel ms/step | gaps ms/step | launches | ' + ' | '.join(w + ' us' for w in watch) + ' |')
for g0 in range(0, len(marks) - 1 - G, G):
    lo, hi = marks[g0], marks[g0 + G]
    sel = rows[lo + 1:hi + 1]
    wall = (rows[hi][1] - rows[lo][1]) / G / 1e6
    ker = sum(e - s for s, e, _ in sel) / G / 1e6
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        for w in watch:
            if w in n:
                per[w][0] += 1
                per[w][1] += e - s
    print('| %d-%d | %.1f | %.2f | %.2f | %.2f | %d | ' % (g0, g0 + G, (rows[lo][1] - rows[marks[0]][1]) / 1e9, wall, ker, wall - ker, len(sel) / G)
          + ' | '.join('%.1f' % (per[w][1] / max(per[w][0], 1) / 1e3) for w in watch) + ' |')

# which kernels changed: first 25 steps vs last 25 steps
def table(lo_m, hi_m):
    sel = rows[marks[lo_m] + 1:marks[hi_m] + 1]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        agg[n][0] += 1
        agg[n][1] += e - s
    return agg
n_steps = len(marks) - 1
A, B = table(0, G), table(n_steps - G, n_steps)
print()
print('| kernel | launches/step early | us/step early | launches/step late | us/step late | delta us/step |')
keys = sorted(set(A) | set(B), key=lambda k: -abs(B.get(k, [0, 0])[1] - A.get(k, [0, 0])[1]))
for k in keys[:25]:
    a, b = A.get(k, [0, 0]), B.get(k, [0, 0])
    print('| `%s` | %.1f | %.1f | %.1f | %.1f | %+.1f |' % (k[:110].replace('|', '/'), a[0] / G, a[1] / G / 1e3, b[0] / G, b[1] / G / 1e3, (b[1] - a[1]) / G / 1e3))
