"""Summarise a rocprofv3 rocpd database (kernel-trace): per-kernel calls / total / average, as CSV on stdout.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("kernel,calls,total_us,avg_us,min_us,max_us,percent" + (",us_per_step" if steps else ""))
for n, c, t, a, mn, mx in rows:
    n = n.replace("(anonymous namespace)::", "").replace(",", ";")
    if len(n) > 110:
        n = n[:110] + "..."
    line = f'"{n}",{c},{t / 1e3:.1f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * t / tot:.2f}'
    if steps:
        line += f",{t / 1e3 / steps:.1f}"
    print(line)
print(f'"TOTAL",,{tot / 1e3:.1f},,,,100' + (f",{tot / 1e3 / steps:.1f}" if steps else ""))
