# timeline of the last cfg3 step (GPU box): start offset, duration, gap to the previous kernel's end, name
export TMPDIR=/tmp
rm -rf /tmp/p2; rocprofv3 --kernel-trace -d /tmp/p2 -o t -- python3 bench.py --workload mul_cleanup --steps 3 --warmup 1 --no-cpu --no-api > /dev/null 2>&1
python3 - <<'PY'
import sqlite3
cur = sqlite3.connect('/tmp/p2/t_results.db').cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# last step: from the last k_hash_rows / k_mul_coeff backwards
starts = [i for i, r in enumerate(rows) if 'k_mul_coeff' in r[0]]
i0 = starts[-1]
while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 50000 and 'k_emit' not in rows[i0 - 1][0]:
    i0 -= 1
t0 = rows[i0][1]
prev_end = t0
for n, a, b in rows[i0:]:
    nm = n.replace('symgpu::', '').replace('void ', '')[:48]
    print(f"{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:8.1f}  gap {(a - prev_end) / 1e3:7.1f}  {nm}")
    prev_end = max(prev_end, b)
print('step span us', (rows[-1][2] - t0) / 1e3)
PY
