# timeline of the last step of a bench workload (GPU box): bash tools/timeline_step.sh gf2|rotation|mul_cleanup|adjacency [min_gap_us]
# start offset, duration, gap to the previous kernel's end, name — a step is taken to start after a gap of more than min_gap_us (default 200)
export TMPDIR=/tmp
wl=${1:-gf2}; gap=${2:-200}
rm -rf /tmp/p3; rocprofv3 --kernel-trace -d /tmp/p3 -o t -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-api --no-extras > /dev/null 2>&1
python3 - $gap <<'PY'
import sqlite3, sys
gap = float(sys.argv[1]) * 1e3
cur = sqlite3.connect('/tmp/p3/t_results.db').cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
i0 = len(rows) - 1
while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < gap:
    i0 -= 1
t0 = rows[i0][1]; prev_end = t0
agg = {}
for n, a, b in rows[i0:]:
    nm = n.replace('symgpu::', '').replace('void ', '')[:44]
    agg.setdefault(nm, [0, 0.0]); agg[nm][0] += 1; agg[nm][1] += (b - a) / 1e3
    if len(rows) - i0 <= 80:
        print(f"{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:8.1f}  gap {(a - prev_end) / 1e3:7.1f}  {nm}")
    prev_end = max(prev_end, b)
span = (rows[-1][2] - t0) / 1e3
busy = sum(v[1] for v in agg.values())
print('kernels', len(rows) - i0, 'step span us', round(span, 1), 'sum of kernel time us', round(busy, 1), 'idle us', round(span - busy, 1))
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"   {c:5d} x {t / c:8.1f} us = {t:9.1f}  {nm}")
PY
