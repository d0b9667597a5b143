"""per-kernel totals of a rocprofv3 results .db (sqlite): python tools/kernel_stats_db.py <results.db> [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute('select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start) from %s d '
                  'join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc' % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
print('| kernel | calls | total us | avg us | min us | % |\n|---|---|---|---|---|---|')
for name, n, t, a, m in rows[:top]:
    print('| `%s` | %d | %.1f | %.2f | %.2f | %.1f |' % (name[:110], n, t / 1e3, a / 1e3, m / 1e3, 100.0 * t / tot))
