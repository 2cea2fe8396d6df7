"""How well does ONE rank's share of the frame run?  Renders rank 0's rows of an N-way sharded 1080p x 256 spp frame on this GPU
and compares with 1/N of the whole-frame time (strong-scaling efficiency before any collective).  Usage: shard_probe.py [N ...]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from path_tracer_amd import api, scenes

W, H, SPP = 1920, 1080, 256
sc = scenes.cornell_box(W, H)


def timed(world):
    r = api.Renderer(sc, W, H, max_bounces=8, rank=0, world_size=world, strip_rows=4)
    r.render_device(0, SPP)
    best = 1e9
    for _ in range(3):
        r.reset_accumulation()
        t0 = time.perf_counter()
        r.render_device(0, SPP)
        r.synchronize()
        best = min(best, time.perf_counter() - t0)
    r.close()
    return best * 1e3


whole = timed(1)
print(f"whole frame: {whole:.2f} ms")
for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    t = timed(n)
    print(f"rank 0 of {n}: {t:.2f} ms  (ideal {whole / n:.2f} ms, efficiency {whole / n / t:.2f})")
