# GPU box: durations of the launches of ONE cfg4 reduction, in launch order
export TMPDIR=/tmp
rm -rf /tmp/p3; rocprofv3 --kernel-trace -d /tmp/p3 -o t -- python3 bench.py --workload gf2 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import sqlite3
cur = sqlite3.connect('/tmp/p3/t_results.db').cursor()
rows = cur.execute("select name, start, duration from kernels where name like '%k_sweep_m4r%' or name like '%k_gf2_spec%' order by start").fetchall()
last = rows[-140:]
print(' '.join(('S' if 'spec' in r[0] else ('A' if '<3>' in r[0] else 'B')) + f'{r[2]/1e3:.1f}' for r in last))
print('span us', (last[-1][1] + last[-1][2] - last[0][1]) / 1e3)
PY
