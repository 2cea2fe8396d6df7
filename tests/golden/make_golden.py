"""Generates the committed fixtures from the CPU oracle (oracle/).  The reference cannot be run here (no rustc) and holds
no fixtures of its own, so these vectors pin the ORACLE against regressions and give the GPU tests inputs/outputs that
need no oracle at run time; they do not pin the oracle to the Rust binary (parity unpinned).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402


def main():
    # BASELINE.json configs[0]: Cornell box 256x256, 16 spp, depth 4
    sc = scenes.cornell_box(256, 256)
    o = O.Oracle(sc)
    acc, pos, idb, ctr = o.render(256, 256, 16, max_bounces=4)
    crop = acc[96:160, 96:160].copy()
    digest = hashlib.sha256(acc.tobytes()).hexdigest()
    # recorded rays: camera rays + one diffuse-ish bounce, with the oracle's hits
    rng = np.random.default_rng(7)
    px = rng.integers(0, 256 * 256, 2048)
    ro = np.zeros((4096, 3), np.float32); rd = np.zeros((4096, 3), np.float32)
    for i, p in enumerate(px):
        ro[i], rd[i] = o.primary_ray(256, 256, int(p), i % 16)
    h = o.trace_closest(ro[:2048], rd[:2048])
    t = np.where(np.isfinite(h["t"]), h["t"], 0).astype(np.float32)
    ro[2048:] = ro[:2048] + rd[:2048] * t[:, None]
    d2 = rng.normal(size=(2048, 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    rd[2048:] = d2.astype(np.float32)
    # renormalise in float32 the way Vec3A::normalize would leave them (|d| within 1 ulp of 1)
    hits = o.trace_closest(ro, rd)
    tmax = rng.uniform(10, 900, 4096).astype(np.float32)
    occl = o.trace_any(ro, rd, tmax)
    np.savez_compressed(os.path.join(HERE, "cornell_c1.npz"), crop=crop, sha256=np.frombuffer(bytes.fromhex(digest), np.uint8),
                        counters=ctr, pos_crop=pos[96:160, 96:160], id_crop=idb[96:160, 96:160],
                        ray_o=ro, ray_d=rd, hit_t=hits["t"], hit_u=hits["u"], hit_v=hits["v"], hit_inst=hits["inst"], hit_prim=hits["prim"],
                        any_tmax=tmax, any_hit=occl)
    # per-sample radiance of a small frame (bit-exact target for the GPU per-sample hook)
    sc64 = scenes.cornell_box(64, 64)
    o64 = O.Oracle(sc64)
    samples = o64.render_samples(64, 64, 4, max_bounces=8)
    mixed = O.Oracle(scenes.cornell_mixed(48, 48)).render_samples(48, 48, 4, max_bounces=8)
    np.savez_compressed(os.path.join(HERE, "samples_small.npz"), cornell64=samples, mixed48=mixed)
    instanced()
    print("wrote fixtures; C1 sha256", digest, "counters", ctr)


def instanced():
    """General rigid instance matrices (scenes.cornell_instanced: glam-built rotations about arbitrary axes + the reference's own
    from_rotation_translation(from_rotation_y(PI), (0, 200, 0)) of main.rs:97-113): the leaves' matrix / inv_matrix, TLAS boxes, camera
    and bounce rays with closest / any hits on both TLASes, per-sample radiance."""
    W, H = 48, 32
    sc = scenes.cornell_instanced(W, H)
    o = O.Oracle(sc)
    rng = np.random.default_rng(11)
    n = 3072
    ro = np.zeros((3 * n, 3), np.float32); rd = np.zeros((3 * n, 3), np.float32)
    px = rng.integers(0, W * H, n)
    for i, p in enumerate(px):
        ro[i], rd[i] = o.primary_ray(W, H, int(p), i % 8)
    h = o.trace_closest(ro[:n], rd[:n])
    t = np.where(np.isfinite(h["t"]), h["t"], 0).astype(np.float32)
    ro[n:2 * n] = ro[:n] + rd[:n] * t[:, None]
    d2 = rng.normal(size=(n, 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    rd[n:2 * n] = d2.astype(np.float32)
    # rays from inside the room aimed at the instances' translations (most camera and bounce rays see walls)
    centres = np.concatenate([m.matrices[:, :, 3] for m in sc.models[4:]]).astype(np.float64)
    o3 = rng.uniform(-260, 260, (n, 3)) + np.array([0.0, 50.0, 0.0])
    d3 = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 45.0, (n, 3)) - o3
    d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    ro[2 * n:] = o3.astype(np.float32); rd[2 * n:] = d3.astype(np.float32)
    out = dict(ray_o=ro, ray_d=rd)
    for which in (0, 1):
        hits = o.trace_closest(ro, rd, which=which)
        for k in ("t", "u", "v", "inst", "prim", "normal", "front"):
            out[f"hit{which}_{k}"] = hits[k]
        ti = o.tlas_instances(which)
        out[f"matrix{which}"] = ti["matrix"]; out[f"inv_matrix{which}"] = ti["inv_matrix"]
        td = o.tlas_dump(which)
        out[f"tlas{which}_boxes"] = td["boxes"]
    tmax = rng.uniform(10, 900, 3 * n).astype(np.float32)
    out["any_tmax"] = tmax
    out["any_hit"] = o.trace_any(ro, rd, tmax)
    out["samples"] = o.render_samples(W, H, 3, max_bounces=7)
    acc, pos, idb, ctr = o.render(W, H, 3, max_bounces=7)
    out["counters"] = ctr; out["position"] = pos; out["id"] = idb
    np.savez_compressed(os.path.join(HERE, "instanced.npz"), **out)
    print("instanced.npz: hits on a general instance:", int((np.isin(out["hit0_inst"], [4, 5, 6, 7, 8, 9, 10, 11, 12]) & np.isfinite(out["hit0_t"])).sum()), "of", 3 * n)


if __name__ == "__main__":
    main()
