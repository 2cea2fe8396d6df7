#!/usr/bin/env python3
"""Shadow-ray tallies of the two routes (GPU box): the Lambertian shading pass walking its own shadow rays (BVH in LDS) against the
queue + k_any route (PT_FLAG_NO_LDS_SCENE), per bounce row and in pt_stats, at several sample counts on one and two pipelines."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from path_tracer_amd import api, scenes
W, H = 1920, 1080
res = {}
for flags in (0, 2):
    for spp, pipes in ((64, 1), (128, 1), (256, 1), (256, 2)):
        r = api.Renderer(scenes.cornell_box(W, H), W, H, max_bounces=8, flags=flags, pipelines=pipes)
        r.render_device(0, spp); r.synchronize()
        st = r.stats()
        rows = r.last_batch_counters().astype(np.int64)
        res[(flags, spp, pipes)] = (st.rays_any, [int(x) for x in rows[:10, 14]])
        r.close()
for spp, pipes in ((64, 1), (128, 1), (256, 1), (256, 2)):
    a, b = res[(0, spp, pipes)], res[(2, spp, pipes)]
    print(spp, pipes, "rays_any inline", a[0], "queued", b[0], "diff", a[0] - b[0])
    print("   rows inline", a[1]); print("   rows queued", b[1])
