#!/usr/bin/env python3
"""k_shade_surface's access pattern, measured before anything is built.  For bounces 1..3 of the Cornell frame: take the REAL
order of path ids in the Lambertian shade queue (a render that stops at that bounce; pt_last_batch_shade_pids), then time a kernel that moves the
shading pass's bytes per hit (48 in, 64-byte path record in and out, 32 + 0.7 x 32 out) with the record addressed (i) by path id, as today, and
(ii) by queue slot, as a state that travels with the ray would be.  tools/microbench/shade_access.hip; (iii), what the traversal kernels would
pay to carry the state, is the -DPT_PROBE_STATE_COPY=1 build timed by tools/ab_any.sh.     shade_access_bench.py [spp] [alu]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from path_tracer_amd import api, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
alus = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 400, 1600]
so = os.path.join(ROOT, "build", "microbench", "shade_access.so")
src = os.path.join(ROOT, "tools", "microbench", "shade_access.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", so, src], check=True)
api.lib()
M = C.CDLL(so)
M.shade_access_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
W, H = 1920, 1080
print(f"# shade access pattern, Cornell {W}x{H}, {spp} spp, Lambertian queue; ms per launch (median of 5), GB/s of the bytes a hit moves")
print("| bounce | slots | hits | holes | alu | (i) record by path id: ms, GB/s | (ii) record by queue slot: ms, GB/s | (i) - (ii) ms | x 256/spp, per frame |")
print("|---|---|---|---|---|---|---|---|---|")
tot = {}
for b in (1, 2, 3, 4):
    r = api.Renderer(scenes.cornell_box(W, H), W, H, max_bounces=b, pipelines=1)
    r.render_device(0, spp); r.synchronize()
    n = int(r.last_batch_counters()[b][9])                  # slots of the Lambertian shade queue at bounce b
    n_rec = int(r.stats().paths)
    pids = r.last_batch_shade_pids(api_qclass := 1, 0, n)
    r.close()
    live = pids != 0xFFFFFFFF
    assert pids[live].max() < n_rec
    for alu in alus:
        ms = (C.c_float * 2)(); by = C.c_double()
        rc = M.shade_access_run(pids.ctypes.data, n, n_rec, alu, 5, ms, C.byref(by))
        assert rc == 0, rc
        d = ms[0] - ms[1]
        tot[alu] = tot.get(alu, 0.0) + d * 256.0 / spp
        print(f"| {b} | {n / 1e6:.2f} M | {live.sum() / 1e6:.2f} M | {(~live).sum() / 1e6:.2f} M | {alu} | {ms[0]:.3f}, {by.value / ms[0] / 1e6:.0f} | {ms[1]:.3f}, {by.value / ms[1] / 1e6:.0f} | {d:.3f} | {d * 256.0 / spp:.2f} |")
print()
for alu, v in tot.items():
    print(f"bounces 1-4, alu {alu}: addressing the record by queue slot instead of by path id would save {v:.2f} ms per 256-spp frame of the shading pass's ~23 ms")
