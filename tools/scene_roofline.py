#!/usr/bin/env python3
"""Roofline of the world closest-hit kernel on a scene whose BVH does NOT fit LDS: algorithmic bytes per launch =
sum over the rays of (48 + nodes_visited * 32 + triangles_tested * 48), nodes / triangles from the oracle's counters 6 and 7 on the
SAME rays (same scene, samples, seed), divided by the kernel's launch time measured with HIP events on the launch stream.
    scene_roofline.py <scene[:level]> <spp> [depth]      prints one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
from oracle import oracle as O
name = sys.argv[1]; spp = int(sys.argv[2]); depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kw = {}
if ":" in name:
    name, lv = name.split(":"); kw["level"] = int(lv)
W, H = 1920, 1080
sc = getattr(scenes, name)(W, H, **kw)
r = api.Renderer(sc, W, H, max_bounces=depth, flags=api.FLAG_TIMING, pipelines=1)  # one batch at a time: launch times are not inflated by a second pipeline
r.render_device(0, spp); r.synchronize(); r.reset_stats(); r.reset_accumulation()
t0 = time.perf_counter(); r.render_device(0, spp); r.synchronize(); wall = time.perf_counter() - t0
st = r.stats()
t0 = time.perf_counter()
_, _, _, ctr = O.Oracle(sc).render(W, H, spp, max_bounces=depth)
cpu_s = time.perf_counter() - t0
assert int(ctr[0]) == st.rays_closest, (int(ctr[0]), st.rays_closest)
alg = 48 * int(ctr[0]) + 32 * int(ctr[6]) + 48 * int(ctr[7])
ms = st.ms_trace_closest
print(json.dumps({"scene": sc.name, "triangles": sc.n_triangles(), "spp": spp, "depth": depth, "scene_bytes": st.scene_bytes, "lds_scene": st.lds_scene,
                  "stack_entries": st.stack_entries, "frame_ms": wall * 1e3, "Mray_per_s": st.rays / wall / 1e6,
                  "rays_closest": st.rays_closest, "rays_any": st.rays_any, "rays_light_closest": st.rays_light_closest,
                  "nodes_visited_per_closest_ray": int(ctr[6]) / max(int(ctr[0]), 1), "triangles_tested_per_closest_ray": int(ctr[7]) / max(int(ctr[0]), 1),
                  "algorithmic_bytes_per_closest_ray": alg / max(int(ctr[0]), 1), "k_closest_ms": ms, "k_closest_launches": st.launches_trace_closest,
                  "k_closest_Mray_per_s": st.rays_closest / max(ms, 1e-9) / 1e3,
                  "roofline": {"bound": "hbm", "achieved": alg / max(ms, 1e-9) / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": alg / max(ms, 1e-9) / 1e6 / 8000.0},
                  "oracle_s": cpu_s, "oracle_threads": max(1, (os.cpu_count() or 2) - 1)}))
