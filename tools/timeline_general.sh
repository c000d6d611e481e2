export TMPDIR=/tmp
rm -rf /tmp/p4; rocprofv3 --kernel-trace -d /tmp/p4 -o t -- python3 tools/bench_general_sizes.py $1 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3
cur = sqlite3.connect('/tmp/p4/t_results.db').cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
i0 = len(rows) - 1
while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 150000: i0 -= 1
t0 = rows[i0][1]; prev = t0
for n, a, b in rows[i0:]:
    print(f"{(a - t0) / 1e3:8.1f} dur {(b - a) / 1e3:6.1f} gap {(a - prev) / 1e3:6.1f}  {n.replace('symgpu::','').replace('void ','')[:60]}")
    prev = max(prev, b)
print('span', (rows[-1][2] - t0) / 1e3, 'kernels', len(rows) - i0)
PY
