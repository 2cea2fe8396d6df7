#!/usr/bin/env python3
"""Per-bounce ray counts and per-launch kernel times of one wavefront batch (run under rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 43
r = api.Renderer(scenes.cornell_box(1920, 1080), 1920, 1080, max_bounces=8, batch_spp=spp)
r.render_device(0, spp)
r.render_device(0, spp)
c = r.last_batch_counters()
print("bounce slots_closest valid_closest slots_shadow valid_shadow slots_lchain valid_lchain lchain_hit slots_term slots_lambert")
for b, row in enumerate(c):
    print(b, row[0], row[13], row[2], row[14], row[4], row[6], row[7], row[8], row[9])
