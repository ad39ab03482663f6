"""Per-dispatch durations of the gnr_head kernels of the last step in a rocprofv3 --kernel-trace database."""
import sqlite3, sys
db, steps = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = list(c.execute("select name, duration, grid_x, grid_y, workgroup_x from kernels order by start"))
sel = rows[-len(rows) // steps:]
for r in sel:
    if 'gnr_head' in r[0] and 'pack' not in r[0]:
        print(f"{r[0].split('gnr_head::')[1][:28]:30s} {r[1] / 1e3:8.1f} us grid {r[2] // r[4]}x{r[3]}")
