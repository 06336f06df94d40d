"""launch-ordered kernel timeline of the LAST repetition of a profiled command (rocprofv3 rocpd database):
   python tools/prof_timeline.py <results.db> <first-kernel-substring> [max rows]
prints name, duration and the gap to the previous kernel from the last dispatch whose name contains the substring on"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sy = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({sy})")]
name_col = "kernel_name" if "kernel_name" in cols else "display_name"
rows = list(cur.execute(f"select s.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {sy} s "
                        f"on d.kernel_id = s.id order by d.start"))
first = max(i for i, r in enumerate(rows) if sys.argv[2] in r[0])
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 200
prev_end = None
total = 0.0
for name, st, en, gx, wx in rows[first:first + limit]:
    short = name.split("(")[0].replace("void ", "").replace("wise::", "")
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    total += (en - st) / 1e3
    print(f"{(en - st) / 1e3:9.2f} us  gap {gap:7.2f}  blocks {gx // max(wx, 1):7d}  {short[:80]}")
    prev_end = en
print(f"sum of durations {total:.1f} us")
