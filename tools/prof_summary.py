"""Turn a rocprofv3 rocpd database (…_results.db, from `rocprofv3 --kernel-trace --stats`) into the
text summary committed under profiles/.   Usage: python tools/prof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    print(f'# {title}')
    print('# per-kernel statistics (rocprofv3 --kernel-trace --stats; durations in microseconds)')
    print(f'{"kernel":72s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                     'from kernels group by name order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows)
    for n, k, s, a, mn, mx in rows:
        print(f'{n[:72]:72s} {k:6d} {s / 1e3:12.1f} {a / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f} {100 * s / tot:6.2f}')
    print('\n# launch geometry / resources')
    print(f'{"kernel":60s} {"grid":>9s} {"wg":>5s} {"lds_B":>7s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"scratch":>7s}')
    for r in c.execute('select distinct name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, '
                       'scratch_size from kernels order by name'):
        print(f'{r[0][:60]:60s} {r[1]:9d} {r[2]:5d} {r[3]:7d} {r[4]:5d} {r[5]:5d} {r[6]:5d} {r[7]:7d}')


if __name__ == '__main__':
    main()
