"""Per-dispatch durations in launch order from a rocprofv3 rocpd database (the last `n` dispatches):
python tools/prof_seq.py <results.db> [n]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(db.execute(f"select {name}, start, end, grid_x, workgroup_x from kernels order by start"))
    rows = rows[-n:]
    t0 = rows[0][1]
    for r in rows:
        print(f"{(r[1] - t0) / 1e3:10.1f} us  +{(r[2] - r[1]) / 1e3:8.2f} us  grid {r[3]:8d} wg {r[4]:4d}  {r[0][:100]}")
    print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
