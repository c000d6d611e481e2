# timeline of the LAST burst of kernels of any command (GPU box): bash tools/timeline_cmd.sh [min_gap_us] -- python3 tools/bench_adj_rows.py 25000
# start offset, duration, gap to the previous kernel's end, name — the burst is taken to start after an idle gap of more than min_gap_us (default 200)
export TMPDIR=/tmp
gap=200; if [ "$1" != "--" ]; then gap=$1; shift; fi; shift
rm -rf /tmp/p3; rocprofv3 --kernel-trace -d /tmp/p3 -o t -- "$@" > /dev/null 2>&1
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
for n, a, b in rows[i0:][-60:]:
    nm = n.replace('symgpu::', '').replace('void ', '')[:60]
    print(f"{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:8.1f}  gap {(a - prev_end) / 1e3:7.1f}  {nm}")
    prev_end = max(prev_end, b)
PY
