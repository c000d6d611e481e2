#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (``*_results.db``) into a small text summary for ``profiles/``.

  python profiles/summarize_rocpd.py <trace.db> [--pmc <pmc.db> ...] > profiles/rNN_<what>.txt

Kernel table = the `--kernel-trace --stats` view (calls, total, avg, min, max, %).  PMC tables = per-kernel sums of each
collected counter (one `--pmc` pass per database, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not
fit in one pass).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x.
"""
import sqlite3, sys


def kernels(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f'# kernel trace: {db}')
    print(f'{"calls":>7} {"total_ms":>11} {"avg_us":>11} {"min_us":>10} {"max_us":>10} {"pct":>6}  name')
    for name, n, s, a, mn, mx in rows:
        print(f'{n:7d} {s / 1e6:11.3f} {a / 1e3:11.3f} {mn / 1e3:10.3f} {mx / 1e3:10.3f} {100 * s / tot:6.2f}  {name[:150]}')
    print()


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    print(f'# pmc: {db}')
    print(f'{"counter":>12} {"dispatches":>10} {"sum":>16} {"avg_per_dispatch":>18} {"avg_dur_us":>11}  kernel')
    for k, c, n, s, a, d in rows:
        print(f'{c:>12} {n:10d} {s:16.1f} {a:18.2f} {d / 1e3:11.3f}  {k[:140]}')
    print()


if __name__ == '__main__':
    args = sys.argv[1:]
    i = 0
    while i < len(args):
        if args[i] == '--pmc':
            pmc(args[i + 1]); i += 2
        else:
            kernels(args[i]); i += 1
