"""Sum of kernel durations and launch count of a rocprofv3 --kernel-trace database (per step, given the number of steps)."""
import sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2])
c = sqlite3.connect(db)
rows = list(c.execute("select name, duration from kernels"))
tot = sum(r[1] for r in rows) / 1e6
print(f"{db}: {len(rows) / steps:.0f} launches / step, {tot / steps:.3f} ms of kernel time / step")
