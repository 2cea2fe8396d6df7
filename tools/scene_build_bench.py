#!/usr/bin/env python3
"""Host-side Scene::new time (pt_add_model + pt_build: triangle set-up, SAH sweep BLAS, TLAS, light tables, flattening) of a bench.py
configuration, for a list of builder thread counts.  No GPU work: pt_build only fills host tables (the upload happens at the first render).
Usage: tools/scene_build_bench.py [config ...] [--threads 1,2,4,8,16]      (default: mesh82k mesh328k)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["mesh82k", "mesh328k"])
ap.add_argument("--threads", default="1,2,4,8,16")
ap.add_argument("--repeat", type=int, default=3)
a = ap.parse_args()
os.environ["PTMI_DEBUG_BUILD"] = "1"
api.lib()
for name in a.configs:
    cfg = bench.CONFIGS[name]
    sd = bench.make_scene(cfg)
    tris = sum(m.positions.shape[0] for m in sd.models)
    for t in a.threads.split(","):
        os.environ["PTMI_BUILD_THREADS"] = t
        best = 1e9
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            r = api.Renderer(sd, 64, 64)
            best = min(best, time.perf_counter() - t0)
            r.close()
        print(f"{name}: {tris} triangles, PTMI_BUILD_THREADS={t}: Scene::new {best * 1e3:.1f} ms (best of {a.repeat})", flush=True)
