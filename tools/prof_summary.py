"""Turn a rocprofv3 rocpd database (…_results.db, from `rocprofv3 --kernel-trace --stats`) into the
text summary committed under profiles/.   Usage: python tools/prof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    """argv: <results.db> [title] [--last-ms N]   (--last-ms: only the kernels that started in the last N ms of the trace,
    e.g. the steady-state steps of a run whose first steps include MIOpen's one-off solver search)"""
    args = sys.argv[1:]
    last_ms = None
    if '--last-ms' in args:
        i = args.index('--last-ms')
        last_ms = float(args[i + 1])
        del args[i:i + 2]
    db = args[0]
    title = args[1] if len(args) > 1 else db
    c = sqlite3.connect(db)
    if last_ms is not None:
        t_end = c.execute('select max(end) from kernels').fetchone()[0]
        c.execute('create temp view k2 as select * from kernels where start >= ?'.replace('?', str(int(t_end - last_ms * 1e6))))
        src = 'k2'
    else:
        src = 'kernels'
    print(f'# {title}')
    print('# per-kernel statistics (rocprofv3 --kernel-trace --stats; durations in microseconds)')
    print(f'{"kernel":72s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                     f'from {src} group by name order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows)
    for n, k, s, a, mn, mx in rows:
        print(f'{n[:72]:72s} {k:6d} {s / 1e3:12.1f} {a / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f} {100 * s / tot:6.2f}')
    print('\n# launch geometry / resources')
    print(f'{"kernel":60s} {"grid":>9s} {"wg":>5s} {"lds_B":>7s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"scratch":>7s}')
    for r in c.execute('select distinct name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, '
                       f'scratch_size from {src} order by name'):
        print(f'{r[0][:60]:60s} {r[1]:9d} {r[2]:5d} {r[3]:7d} {r[4]:5d} {r[5]:5d} {r[6]:5d} {r[7]:7d}')


if __name__ == '__main__':
    main()
