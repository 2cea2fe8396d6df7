#!/usr/bin/env python3
"""Pure traversal throughput through the C-ABI hook (k_closest<HOOK>: no binning, no staging): random rays inside the box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
r = api.Renderer(scenes.cornell_box(64, 64), 64, 64)
rng = np.random.default_rng(0)
O = rng.uniform(-270, 270, (n, 3)).astype(np.float32); O[:, 1] += 50
D = rng.normal(size=(n, 3)); D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
for rep in range(3):
    t0 = time.time(); h = r.trace_closest(O, D); dt = time.time() - t0
print("rays", n, "hit fraction", float((h["inst"] != 0xFFFFFFFF).mean()), "hook wall", dt)
tm = np.full(n, 300.0, np.float32)
for rep in range(2):
    a = r.trace_any(O, D, tm)
print("any hit fraction", float(a.mean()))
