"""Kernel timeline of the last complete step in a rocprofv3 rocpd database: per launch start offset, duration and the
gap to the previous kernel's end.  usage: python tools/timeline.py x_results.db [anchor-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "philox_fill_jobs"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# steps start at the first anchor kernel after a non-anchor kernel
starts = [i for i, r in enumerate(rows) if anchor in r[0] and (i == 0 or anchor not in rows[i - 1][0])]
if len(starts) < 3:
    raise SystemExit("not enough steps")
a, b = starts[-3], starts[-2]
t0 = rows[a][1]
prev_end = None
busy = 0
for n, s, e in rows[a:b]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[:70]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {n}")
    prev_end = e
print(f"step span {(rows[b][1] - t0) / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, launches {b - a}")
per = [(rows[starts[i + 1]][1] - rows[starts[i]][1]) / 1e3 for i in range(max(0, len(starts) - 10), len(starts) - 1)]
print("start-to-start periods of the last steps (us): " + " ".join(f"{p:.0f}" for p in per))
