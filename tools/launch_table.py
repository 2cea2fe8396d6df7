#!/usr/bin/env python3
"""Per-launch comparison of two kernel_times.py --timeline dumps of the same frame (whole frame vs one rank's 1/N share):
launch_table.py whole.txt share.txt N"""
import sys
def frames(path):
    rows = [l.split() for l in open(path) if ' us  dur ' in l]
    out = []; cur = None
    for r in rows:
        name = ' '.join(r[6:]); dur = float(r[3]); start = float(r[0])
        if name.startswith('k_generate'):
            cur = []; out.append(cur)
        if cur is not None and name.startswith('k_'): cur.append((name, dur, start))
    return out
f1 = frames(sys.argv[1])[-1]; f8 = frames(sys.argv[2])[-1]; N = float(sys.argv[3])
tot1 = tot8 = 0
for (n1, d1, s1), (n8, d8, s8) in zip(f1, f8):
    assert n1.split('<')[0] == n8.split('<')[0], (n1, n8)
    if 'k_closest<true, 1' in n1:
        print(f"{n1:32s} {d1:9.1f} {d8:8.1f}  (side stream)"); continue
    tot1 += d1; tot8 += d8
    print(f"{n1:32s} {d1:9.1f} {d8:8.1f}  ideal {d1 / N:8.1f}  eff {d1 / N / d8:5.2f}  lost {d8 - d1 / N:7.1f}")
print(f"sum {tot1:.0f} {tot8:.0f} eff {tot1 / N / tot8:.3f}")
print("frame span", f1[-1][2] + f1[-1][1] - f1[0][2], f8[-1][2] + f8[-1][1] - f8[0][2])
