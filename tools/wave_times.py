#!/usr/bin/env python3
"""Life of the waves of each k_closest launch (needs a -DPT_WAVE_TIMES=1 build: PTMI_LIB=build/variants/wt.so): when they start, first
have rays, find the queue empty, and end.   wave_times.py [world] [spp] [scene[:level]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import api, scenes
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
name = sys.argv[3] if len(sys.argv) > 3 else "cornell_box"
kw = {}
if ":" in name:
    name, lv = name.split(":"); kw["level"] = int(lv)
W, H = 1920, 1080
r = api.Renderer(getattr(scenes, name)(W, H, **kw), W, H, max_bounces=8, pipelines=1, rank=0, world_size=world, strip_rows=4)
r.render_device(0, spp); r.synchronize()
r.reset_accumulation(); r.render_device(0, spp); r.synchronize()
st = r.last_batch_step_stats().astype(np.int64)
ctr = r.last_batch_counters().astype(np.float64)
T = 0.01  # us per tick (100 MHz)
print(f"# wave times of k_closest, {name} {kw}, rank 0 of {world}, {spp} spp: microseconds from the first wave's start")
print("| bounce | rays | waves | launch span | waves start by (avg) | first rays (avg after start) | queue empty: first wave / last wave | wave ends: avg life | span - last 'queue empty' |")
print("|---|---|---|---|---|---|---|---|---|")
for b in range(len(st)):
    nmin_start, max_end, sum_life, sum_dr, waves, nmin_dr, max_dr, sum_first = [int(x) for x in st[b]]
    if waves == 0: continue
    t0 = (~nmin_start) & 0xffffffff
    d = lambda t: ((t - t0) & 0xffffffff) * T
    first_dr = (~nmin_dr) & 0xffffffff
    print(f"| {b} | {ctr[b][13] / 1e6:.2f} M | {waves} | {d(max_end):.0f} | - | {sum_first / waves * T:.1f} | {d(first_dr):.0f} / {d(max_dr):.0f} | {sum_life / waves * T:.0f} (queue empty after {sum_dr / waves * T:.0f}) | {d(max_end) - d(max_dr):.0f} |")
