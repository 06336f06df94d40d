"""Print the per-kernel summary of a rocprofv3 rocpd database: python tools/prof_top.py <results.db> [n] [--csv out.csv]"""
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    csv_out = None
    if "--csv" in args:
        i = args.index("--csv")
        csv_out = args[i + 1]
        del args[i:i + 2]
    db = sqlite3.connect(args[0])
    n = int(args[1]) if len(args) > 1 else 25
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    if csv_out:
        with open(csv_out, "w") as f:
            f.write('"Name","Calls","TotalDurationUs","AverageUs","Percentage"\n')
            for r in rows:
                f.write('"%s",%d,%.3f,%.3f,%.4f\n' % (r[0].replace('"', "'"), r[1], r[2], r[3], r[4]))
    for r in rows[:n]:
        print(f"{r[4]:6.2f}%  calls {r[1]:5d}  avg {r[3]:10.2f} us  total {r[2] / 1e3:9.3f} ms  {r[0][:110]}")


if __name__ == "__main__":
    main()
