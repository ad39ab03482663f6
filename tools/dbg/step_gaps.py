"""GPU idle time between the kernels of the forward step, from a rocprofv3 --kernel-trace database of tools/run_hot.py:
python tools/dbg/step_gaps.py <results.db>   -> for the last 3 steps (a step starts at k_view_setup): wall time from the first kernel's start to
the last kernel's end, the sum of kernel durations, and the gaps grouped by the kernel that FOLLOWS the gap."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute('select name, start, end from kernels order by start').fetchall()
starts = [i for i, r in enumerate(rows) if 'k_view_setup' in r[0]]
for si in range(len(starts) - 4, len(starts) - 1):
    seg = rows[starts[si]:starts[si + 1]]
    wall = (seg[-1][2] - seg[0][1]) / 1e3
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    gaps = collections.Counter()
    for a, b in zip(seg, seg[1:]):
        g = (b[1] - a[2]) / 1e3
        gaps[b[0].split('(')[0][-40:]] += max(g, 0.0)
    nxt = (rows[starts[si + 1]][1] - seg[-1][2]) / 1e3
    print(f'step: {len(seg)} kernels, wall {wall:.1f} us, kernels {busy:.1f} us, gaps {wall - busy:.1f} us, to next step {nxt:.1f} us')
    for k, v in gaps.most_common(8):
        print(f'    {v:7.1f} us before {k}')
