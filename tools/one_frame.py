#!/usr/bin/env python3
"""One 1080p x 256 spp frame (or rank 0's share of a WORLD-way split) for profiling: one_frame.py [world] [batch_spp] [scene] [spp] [pipelines]"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from path_tracer_amd import api, scenes
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
name = sys.argv[3] if len(sys.argv) > 3 else "cornell_box"
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 256
kw = {}
pipes = int(sys.argv[5]) if len(sys.argv) > 5 else 0
stack_lds = int(sys.argv[6]) if len(sys.argv) > 6 else 0
if ":" in name:
    name, lv = name.split(":"); kw["level"] = int(lv)
W, H = 1920, 1080
r = api.Renderer(getattr(scenes, name)(W, H, **kw), W, H, max_bounces=8, rank=0, world_size=world, strip_rows=4, batch_spp=batch, pipelines=pipes, stack_lds_levels=stack_lds)
r.render_device(0, spp); r.synchronize()
r.reset_accumulation(); r.reset_stats()
import time
t0 = time.perf_counter(); r.render_device(0, spp); r.synchronize(); dt = time.perf_counter() - t0
st = r.stats()
print(f"{name} world {world} batch {batch} pipes {pipes} spp {spp}: {dt * 1e3:.2f} ms, {st.rays / dt / 1e9:.2f} Gray/s, state {st.state_bytes / 2**30:.1f} GiB")
for row in r.last_batch_counters():
    print("B", *[int(x) for x in row])
