"""Summarise an SQ-counter rocprofv3 pass per kernel: mean of each counter over dispatches."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("wise::", "") + f" grid={r['Grid_Size']}"
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[k][1] += 1
for k in acc:
    if not any(s in k for s in sys.argv[2:]) and len(sys.argv) > 2:
        continue
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    print(f"{k}  n={dur[k][1]} avg_us={dur[k][0]/max(dur[k][1],1):.1f}")
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    for n, v in sorted(c.items()):
        print(f"    {n:28s} {v:16.0f}  ({v/wc*100:6.1f}% of WAVE_CYCLES)")
