#!/usr/bin/env python3
"""The same Cornell frame with the BVH staged in LDS and (PT_FLAG_NO_LDS_SCENE) read from global memory through L1/L2: how much of the
traversal time is memory latency?  1080p, 64 spp, depth 8."""
import sys, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from path_tracer_amd import api, scenes
W, H, SPP = 1920, 1080, 64
for flags in (0, api.FLAG_NO_LDS_SCENE):
    r = api.Renderer(scenes.cornell_box(W, H), W, H, max_bounces=8, flags=flags | api.FLAG_TIMING_ALL, pipelines=1)
    r.render_device(0, SPP); r.synchronize(); r.reset_stats(); r.reset_accumulation()
    t0 = time.perf_counter(); r.render_device(0, SPP); r.synchronize(); dt = time.perf_counter() - t0
    st = r.stats()
    print(f"lds_scene={st.lds_scene}: {dt * 1e3:.2f} ms; closest {st.ms_trace_closest:.2f} ms ({st.rays_closest / st.ms_trace_closest / 1e6:.2f} Gray/s), any {st.ms_trace_any:.2f} ms, "
          f"light {st.ms_trace_light:.2f} ms, shade {st.ms_shade:.2f} ms")
    r.close()
