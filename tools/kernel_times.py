#!/usr/bin/env python3
"""Per-kernel totals and the launch timeline of a rocprofv3 --kernel-trace CSV.  Usage: kernel_times.py trace.csv [--timeline]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"pt::\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n
tot = {}
for r in rows:
    k = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += d
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} launches over {span / 1e3:.3f} ms")
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} {n:5d} launches {d / 1e3:9.3f} ms  avg {d / n:9.1f} us")
if "--timeline" in sys.argv:
    t0 = int(rows[0]["Start_Timestamp"]); prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {((s - prev) / 1e3 if prev else 0):6.1f}  {short(r['Kernel_Name'])}")
        prev = e
