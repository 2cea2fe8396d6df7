#!/usr/bin/env python3
"""Wave-steps of the closest-hit kernel with deferred leaves (k_closest2; needs a -DPT_STEP_STATS=1 build: PTMI_LIB=build/variants/stats.so).
Per bounce: wave-steps, how many were branch steps / leaf rounds and with how many lanes, lanes that could do neither (parked or
finished and waiting for the service).   step_stats2.py [scene[:level]] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import api, scenes
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
print(f"# wave-steps of k_closest2, scene {name} {kw}, 1920x1080, {spp} spp, depth 8 (one batch)")
print("| bounce | rays | wave-steps | per ray | lanes holding a ray | branch steps (share) | lanes per branch step | leaf rounds (share) | lanes per leaf round | of them testing triangles | idle lanes with a ray per step |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
tot = np.zeros(8); rays_tot = 0
def line(tag, rays, row):
    it, act, lb, ll, lt, wb, wl, park = row
    print(f"| {tag} | {rays / 1e6:.2f} M | {it / 1e6:.2f} M | {it / max(rays, 1):.3f} | {act / it:.1f} | {wb / 1e6:.2f} M ({wb / it:.2f}) | {lb / max(wb, 1):.1f} | {wl / 1e6:.2f} M ({wl / it:.2f}) | {ll / max(wl, 1):.1f} | {lt / max(wl, 1):.1f} | {park / it:.1f} |")
for b in range(len(st)):
    if st[b][0] == 0: continue
    rays = ctr[b][13]
    tot += st[b]; rays_tot += rays
    line(b, rays, st[b])
line("all", rays_tot, tot)
it, act, lb, ll, lt, wb, wl, park = tot
print(f"\nPer ray: {lb / rays_tot:.2f} branch steps, {ll / rays_tot:.2f} candidates taken, {lt / rays_tot:.2f} leaves tested; lanes doing work per wave-step: {(lb + ll) / it:.1f} of 64.")

os.environ["PTMI_STEP_STATS_BASE"] = "16"
tt = r.last_batch_step_stats().astype(np.float64).sum(axis=0)
if tt[4] > 0 and tt[4] > tt[0] * 100:   # k_closest2, PT_STEP_STATS=2 build: section times
    n_service, t_service, t_branch, t_leaf, t_total = tt[:5]
    print(f"\nWave time by section (PT_STEP_STATS=2 build): service {t_service / t_total:.3f} ({n_service / 1e6:.2f} M services, {64 * t_service / max(n_service, 1):.0f} ticks each), "
          f"branch steps {t_branch / t_total:.3f} ({64 * t_branch / max(wb, 1):.0f} ticks each), leaf rounds {t_leaf / t_total:.3f} ({64 * t_leaf / max(wl, 1):.0f} ticks each), "
          f"rest {1 - (t_service + t_branch + t_leaf) / t_total:.3f}; rays per service {rays_tot / max(n_service, 1):.1f}")
elif tt[4] > 0:                          # k_closest3: batches
    n_out, lanes_out, n_in, lanes_in, bound = tt[:5]
    print(f"\nStreamed lanes (k_closest3): {bound / 1e6:.2f} M looks at the buffers ({it / bound:.2f} wave-steps between two), {n_out / 1e6:.3f} M retirement batches of {lanes_out / max(n_out, 1):.1f} lanes, "
          f"{n_in / 1e6:.3f} M set-up batches of {lanes_in / max(n_in, 1):.1f} rays")
