#!/usr/bin/env python3
"""Where the traversal steps of the world closest-hit kernel go (needs a -DPT_STEP_STATS=1 build: PTMI_LIB=build/variants/stats.so).
Per bounce: wave-steps, average lanes active, and for the instance / branch / triangle-leaf sections of a step how many lanes take the
section per wave-step that executes it.   step_stats.py [scene[:level]] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import api, scenes
if len(sys.argv) > 1 and sys.argv[1] == "any":      # step_stats.py any [scene] [spp]: the shadow-ray kernel (k_any) instead
    os.environ["PTMI_STEP_STATS_QUEUE"] = "shadow"
    del sys.argv[1]
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
kw = {}
if ":" in name:
    name, lv = name.split(":"); kw["level"] = int(lv)
W, H = 1920, 1080
r = api.Renderer(getattr(scenes, name)(W, H, **kw), W, H, max_bounces=8, pipelines=1)
r.render_device(0, spp); r.synchronize()
st = r.last_batch_step_stats().astype(np.float64)
ctr = r.last_batch_counters().astype(np.float64)
print(f"# step statistics of {'k_any (shadow rays cast by the bounce)' if os.environ.get('PTMI_STEP_STATS_QUEUE') else 'k_closest'}, scene {name} {kw}, 1920x1080, {spp} spp, depth 8 (one batch)")
print("| bounce | rays | wave-steps | steps per ray (lane-steps / rays) | lanes active per wave-step | instance: lanes per executing wave-step (share of wave-steps) | branch | triangle leaf |")
print("|---|---|---|---|---|---|---|---|")
tot = np.zeros(8); rays_tot = 0
for b in range(len(st)):
    it, act, li, lb, ll, wi, wb, wl = st[b]
    rays = ctr[b][14] if os.environ.get("PTMI_STEP_STATS_QUEUE") else ctr[b][13]
    if it == 0: continue
    tot += st[b]; rays_tot += rays
    f = lambda l, w: f"{l / max(w, 1):.1f} ({w / it:.2f})"
    print(f"| {b} | {rays / 1e6:.2f} M | {it / 1e6:.2f} M | {act / max(rays, 1):.1f} | {act / it:.1f} | {f(li, wi)} | {f(lb, wb)} | {f(ll, wl)} |")
it, act, li, lb, ll, wi, wb, wl = tot
print(f"| all | {rays_tot / 1e6:.2f} M | {it / 1e6:.2f} M | {act / max(rays_tot, 1):.1f} | {act / it:.1f} | {li / max(wi, 1):.1f} ({wi / it:.2f}) | {lb / max(wb, 1):.1f} ({wb / it:.2f}) | {ll / max(wl, 1):.1f} ({wl / it:.2f}) |")
print()
print(f"Per ray: {li / rays_tot:.2f} instance steps, {lb / rays_tot:.2f} branch steps, {ll / rays_tot:.2f} leaf steps; a wave-step executes the instance section "
      f"{wi / it:.2f}, the branch section {wb / it:.2f} and the leaf section {wl / it:.2f} of the time.")
