"""Extended randomized parity run (not part of the test suite): random scenes, per-sample radiance + frame + ray tallies, GPU vs oracle.
Usage: python tools/fuzz_parity.py [first_seed] [n_seeds] [width height]   (default: tiny frames, 40x24 / 64x20; a few hundred pixels a side
exercise the striped tails, tapered chunks and dynamic claims that tiny queues never reach)"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import oracle as O
from path_tracer_amd import api, scenes

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
size = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
every = 500 if size is None else 20
O.build()


def bits(a):
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


bad = 0
n_general = 0   # instance matrices that are rotations about a general axis (nine non-zero entries), summed over the scenes
with_general = 0
t0 = time.time()
for seed in range(first, first + count):
    w, h = size if size else ((40, 24) if seed % 3 else (64, 20))
    sc = scenes.random_scene(seed, w, h, with_media=(seed % 2 == 0))
    g = sum(int(np.count_nonzero(m[:, :3]) > 3) for mo in sc.models for m in mo.matrices)
    n_general += g
    with_general += 1 if g else 0
    o = O.Oracle(sc)
    # every fourth scene keeps its BVH in global memory (the kernels the mesh configurations run)
    r = api.Renderer(sc, w, h, max_bounces=6 + seed % 9, flags=api.FLAG_NO_LDS_SCENE if seed % 4 == 3 else 0)
    mb = 6 + seed % 9
    ok = np.array_equal(bits(r.render_samples(0, 3)), bits(o.render_samples(w, h, 3, max_bounces=mb)))
    r.reset_stats(); r.reset_accumulation()
    acc, pos, idb = r.render(3, 2)
    oacc, opos, oid, octr = o.render(w, h, 2, first_sample=3, max_bounces=mb)
    st = r.stats()
    ok = ok and np.array_equal(bits(acc), bits(oacc)) and np.array_equal(bits(pos), bits(opos)) and np.array_equal(idb, oid)
    ok = ok and (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, flush=True)
    r.close()
    if (seed - first + 1) % every == 0:
        print(f"{seed - first + 1} scenes so far, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
print(f"{count} scenes ({with_general} with at least one general rigid instance, {n_general} such instances in all), {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
