#!/usr/bin/env python3
"""What could ray binning buy the global-BVH traversal?  Bounce-like rays on a mesh scene (origins = first hits of the camera rays,
directions random) traced through the C-ABI hook in three orders: as generated (pixel order), shuffled, sorted by direction octant x
origin cell.  Run under rocprofv3 --kernel-trace (tools/ktrace-style) and read the k_closest<.., 2, ..> launches in order:
3 x pixel order, 3 x shuffled, 3 x sorted (the first call warms up).   sort_probe.py [level] [cells per axis]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import api, scenes
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
sc = scenes.cornell_mesh(W, H, level=level)
r = api.Renderer(sc, W, H)
rect, box = r.active_pixels()
x0, w, y0, h = rect
xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + h))
n = xs.size
O = np.zeros((n, 3), np.float32); D = np.zeros((n, 3), np.float32)
# camera rays through pixel centres (host evaluation, camera.rs:94-105)
eye, _ = r.create_ray(0.5, 0.5)
m, ip = r.camera_matrices()
u = (xs.ravel() + 0.5) / W; v = (ys.ravel() + 0.5) / H
# direction through create_ray for a coarse grid would be slow in Python: use the pinhole directly from two corner rays
o00, d00 = r.create_ray(0.0, 0.0); _, d10 = r.create_ray(1.0, 0.0); _, d01 = r.create_ray(0.0, 1.0)
# rays are linear in (u, v) before normalisation up to the perspective divide; good enough for a probe
dd = (np.asarray(d00)[None] / d00[2] * 1.0) + 0
dirs = (np.asarray(d00) / -d00[2])[None] + u[:, None] * ((np.asarray(d10) / -d10[2]) - (np.asarray(d00) / -d00[2]))[None] + v[:, None] * ((np.asarray(d01) / -d01[2]) - (np.asarray(d00) / -d00[2]))[None]
dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
O[:] = np.asarray(o00, np.float32)
hit = r.trace_closest(O, dirs)
ok = hit["inst"] != 0xFFFFFFFF
P = (O + dirs * hit["t"][:, None])[ok]
rng = np.random.default_rng(0)
n = P.shape[0]
Dn = rng.normal(size=(n, 3)); Dn = (Dn / np.linalg.norm(Dn, axis=1, keepdims=True)).astype(np.float32)
P = (P + Dn * np.float32(0.01)).astype(np.float32)   # off the surface
print("bounce rays", n)
def run(o, d, tag):
    for _ in range(3):
        t0 = time.time(); hh = r.trace_closest(o, d); dt = time.time() - t0
    print(tag, "hit fraction", float((hh["inst"] != 0xFFFFFFFF).mean()), "hook wall", round(dt, 3))
run(P, Dn, "pixel order")
perm = rng.permutation(n)
run(P[perm], Dn[perm], "shuffled")
lo = np.asarray(box[:3]); ext = np.asarray(box[3:]) - lo
cell = np.clip(((P - lo) / ext * cells).astype(np.int64), 0, cells - 1)
octant = (Dn[:, 0] < 0).astype(np.int64) | ((Dn[:, 1] < 0).astype(np.int64) << 1) | ((Dn[:, 2] < 0).astype(np.int64) << 2)
key = ((octant * cells + cell[:, 0]) * cells + cell[:, 1]) * cells + cell[:, 2]
order = np.argsort(key, kind="stable")
run(P[order], Dn[order], f"sorted by octant x {cells}^3 cells")
# groups of 64 consecutive rays kept together, the groups shuffled / dealt by a multiplicative permutation
G = n // 64
gperm = rng.permutation(G)
idx = (gperm[:, None] * 64 + np.arange(64)[None]).ravel()
run(P[idx], Dn[idx], "64-ray groups shuffled")
Pm = 1000003
gm = (np.arange(G, dtype=np.int64) * Pm) % G
idx = (gm[:, None] * 64 + np.arange(64)[None]).ravel()
run(P[idx], Dn[idx], "64-ray groups dealt (g * 1000003 mod G)")
G4 = n // 1024
g4 = rng.permutation(G4)
idx = (g4[:, None] * 1024 + np.arange(1024)[None]).ravel()
run(P[idx], Dn[idx], "1024-ray groups shuffled")
