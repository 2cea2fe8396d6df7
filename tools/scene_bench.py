#!/usr/bin/env python3
"""Throughput of the other BASELINE.json config classes (parity-test scenes, not the headline bench):
   python tools/scene_bench.py <scene> [spp] [level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_mesh"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
kw = {"level": int(sys.argv[3])} if len(sys.argv) > 3 else {}
W, H = 1920, 1080
t0 = time.time(); sc = getattr(scenes, name)(W, H, **kw); t1 = time.time()
r = api.Renderer(sc, W, H, max_bounces=8, flags=api.FLAG_TIMING_ALL); t2 = time.time()
r.render_device(0, spp); r.reset_stats(); r.reset_accumulation()  # same batch size as the timed run: no reallocation inside it
t3 = time.time(); r.render_device(0, spp); r.synchronize(); t4 = time.time()
st = r.stats()
print(f"{name} {kw} tris={sc.n_triangles()} gen={t1-t0:.2f}s build={t2-t1:.2f}s render={1e3*(t4-t3):.1f}ms rays={st.rays} -> {st.rays/(t4-t3)/1e6:.0f} Mray/s "
      f"paths/s={st.paths/(t4-t3)/1e6:.0f}M lds_scene={st.lds_scene} scene_bytes={st.scene_bytes} stack={st.stack_entries}")
print({k: round(getattr(st, k), 1) for k in ("ms_trace_closest", "ms_trace_any", "ms_trace_light", "ms_shade", "ms_generate", "ms_accumulate")})
