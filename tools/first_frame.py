#!/usr/bin/env python3
"""Cold-start times of a bench.py configuration: Scene::new on the host, the first render (scene upload + wavefront pool allocation +
the frame), the second render (the frame alone).  Usage: tools/first_frame.py [config] [--spp N]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="cornell")
ap.add_argument("--spp", type=int, default=0)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
w, h, spp, depth = cfg[2], cfg[3], a.spp or cfg[5], cfg[6]
sd = bench.make_scene(cfg)
api.lib()
t0 = time.perf_counter()
r = api.Renderer(sd, w, h, max_bounces=depth)
t1 = time.perf_counter()
r.render_device(0, spp); r.synchronize()
t2 = time.perf_counter()
r.reset_accumulation()
r.render_device(0, spp); r.synchronize()
t3 = time.perf_counter()
st = r.stats()
print(f"{a.config} {w}x{h} {spp} spp depth {depth}: Scene::new {1e3 * (t1 - t0):.1f} ms, first render {1e3 * (t2 - t1):.1f} ms, "
      f"second render {1e3 * (t3 - t2):.1f} ms, wavefront state {st.state_bytes / 2**30:.1f} GiB")
