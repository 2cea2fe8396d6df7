#!/usr/bin/env python3
"""Kernel duration vs launch size through the C-ABI hooks (k_closest<HOOK>, k_any<HOOK>): random rays inside the Cornell room.
Run under rocprofv3 --kernel-trace and read the durations with tools/kernel_times.py --timeline."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
sizes = [int(a) for a in sys.argv[1:]] or [64, 4096, 65536, 1 << 20, 1 << 22, 1 << 24]
r = api.Renderer(scenes.cornell_box(64, 64), 64, 64)
rng = np.random.default_rng(0)
nmax = max(sizes)
O = rng.uniform(-270, 270, (nmax, 3)).astype(np.float32); O[:, 1] += 50
D = rng.normal(size=(nmax, 3)); D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
tm = np.full(nmax, 300.0, np.float32)
for n in sizes:
    for rep in range(3):
        r.trace_closest(O[:n], D[:n])
    for rep in range(3):
        r.trace_any(O[:n], D[:n], tm[:n])
    print("done", n, flush=True)
