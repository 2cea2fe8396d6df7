"""GPU (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same seeded inputs and against
the committed golden fixtures.  Bar: bit-exact (float results compared as u32 bit patterns)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def api():
    from path_tracer_amd import api
    api.lib()
    return api


@pytest.fixture(scope="module")
def r256(api, cornell256):
    return api.Renderer(cornell256, 256, 256, max_bounces=4)


def test_device_arithmetic_matches_host(api, oracle_mod, cornell64):
    """division, sqrt, the deterministic sin/cos/exp/ln/hypot and the WyRand stream are the same function on both sides"""
    r = api.Renderer(cornell64, 64, 64)
    rng = np.random.default_rng(0)
    n = 1 << 16
    x = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    gs, gc = r.math_batch(0, x)
    cs, cc = oracle_mod.math_batch(0, x)
    assert_bit_equal(gs, cs, "sin"); assert_bit_equal(gc, cc, "cos")
    x = np.concatenate([rng.uniform(-100, 100, n - 4), [0, -0.0, 88.8, -104]]).astype(np.float32)
    assert_bit_equal(r.math_batch(1, x)[0], oracle_mod.math_batch(1, x)[0], "exp")
    x = np.concatenate([np.exp(rng.uniform(-80, 80, n - 4)), [0, 1, 1e-42, 3e38]]).astype(np.float32)
    assert_bit_equal(r.math_batch(2, x)[0], oracle_mod.math_batch(2, x)[0], "ln")
    a = (rng.normal(size=n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    b = (rng.normal(size=n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    a[:4] = [0, 1, -1, 3]; b[:4] = [0, 0, 3, -0.0]
    assert_bit_equal(r.math_batch(3, a, b)[0], oracle_mod.math_batch(3, a, b)[0], "hypot")
    with np.errstate(all="ignore"):
        assert_bit_equal(r.math_batch(4, a, b)[0], oracle_mod.math_batch(4, a, b)[0], "divide")
        assert_bit_equal(r.math_batch(5, np.abs(a))[0], oracle_mod.math_batch(5, np.abs(a))[0], "sqrt")
    # denormal operands must not be flushed
    d = (rng.uniform(1, 100, n) * 1e-41).astype(np.float32)
    assert_bit_equal(r.math_batch(4, d, np.full(n, 3.0, np.float32))[0], (d / np.float32(3.0)).astype(np.float32), "denormal divide")
    # RNG: first u32 and second f32 draw of stream (pixel, sample)
    px = rng.integers(0, 1 << 31, n).astype(np.uint32); sm = rng.integers(0, 4096, n).astype(np.uint32)
    g0, g1 = r.math_batch(7, px.view(np.float32), sm.view(np.float32))
    L = oracle_mod.lib()
    for i in range(0, n, 997):
        s0 = L.pto_stream_state0(api.DEFAULT_SEED, int(px[i]), int(sm[i]))
        assert int(g0[i:i + 1].view(np.uint32)[0]) == L.pto_wyrand(s0, 0) & 0xFFFFFFFF
        assert g1[i] == np.float32(np.float32(L.pto_wyrand(s0, 1) & 0xFFFFFFFF) / np.float32(4294967296.0))


def test_device_sobol_matches_known_answers(api, oracle_mod, cornell64):
    r = api.Renderer(cornell64, 64, 64)
    kat = [(0, 0, 0xA30AE09A, 0x231C175E), (1, 0, 0x19984A78, 0x9ECD1C95), (0, 1, 0x6B17BD49, 0x0EE62FF7),
           (5, 0xDEADBEEF, 0x646F0B04, 0x9E5AF2A8), (255, 12345, 0xE2891023, 0x053CCA92)]
    out = r.ss_sobol(512, [k[0] for k in kat], [k[1] for k in kat])
    for o, k in zip(out, kat):
        assert o[0] == np.float32(np.float32(k[2]) / np.float32(4294967296.0)) and o[1] == np.float32(np.float32(k[3]) / np.float32(4294967296.0))
    rng = np.random.default_rng(1)
    for n_points in (512, 2, 100000, 1 << 31):
        idx = rng.integers(0, 1 << 32, 4096, dtype=np.uint64).astype(np.uint32); seed = rng.integers(0, 1 << 32, 4096, dtype=np.uint64).astype(np.uint32)
        g = r.ss_sobol(n_points, idx, seed)
        c = np.stack([oracle_mod.ss_sobol(n_points, int(i), int(s)) for i, s in zip(idx[:512], seed[:512])])
        assert_bit_equal(g[:512], c, f"ss_sobol N={n_points}")


def test_trace_closest_golden_rays(r256):
    g = np.load(os.path.join(GOLD, "cornell_c1.npz"))
    h = r256.trace_closest(g["ray_o"], g["ray_d"])
    for k in ("t", "u", "v", "inst", "prim"):
        assert_bit_equal(h[k], g["hit_" + k], f"closest.{k}")


def test_trace_any_golden_rays(r256):
    g = np.load(os.path.join(GOLD, "cornell_c1.npz"))
    assert np.array_equal(r256.trace_any(g["ray_o"], g["ray_d"], g["any_tmax"]), g["any_hit"])


@pytest.mark.parametrize("scene_name,n", [("cornell_box", 20000), ("cornell_mixed", 5000)])
def test_trace_matches_oracle_on_random_rays(api, oracle_mod, scene_name, n):
    from path_tracer_amd import scenes
    sc = getattr(scenes, scene_name)(64, 64)
    r = api.Renderer(sc, 64, 64)
    o = oracle_mod.Oracle(sc)
    rng = np.random.default_rng(3)
    O = rng.uniform(-270, 270, (n, 3)).astype(np.float32)
    O[:, 1] += 50
    D = rng.normal(size=(n, 3))
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    # axis-aligned directions (zero components -> infinite inverse direction -> NaN slabs) and degenerate cases
    D[:300] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)] * rng.choice([-1.0, 1.0], (300, 1)).astype(np.float32)
    O[300:400] = np.round(O[300:400] / 139) * 139           # origins on box planes
    D[400:420, 2] = -0.0
    D[400:420] /= np.linalg.norm(D[400:420], axis=1, keepdims=True)
    # 0 * inf: origin exactly on a wall / box plane AND a (signed) zero direction component on that axis
    planes = np.array([-278.0, 278.0, -228.0, 328.0, 0.0, 139.0, -139.0], np.float32)
    for i in range(420, 720):
        ax = int(rng.integers(0, 3))
        O[i, ax] = planes[rng.integers(0, len(planes))]
        D[i, ax] = rng.choice(np.array([0.0, -0.0], np.float32))
        if i % 3 == 0:                                      # two zero components: an axis-aligned ray sliding along a plane
            ax2 = (ax + 1) % 3
            D[i, ax2] = rng.choice(np.array([0.0, -0.0], np.float32))
        nrm = np.linalg.norm(D[i].astype(np.float64))
        D[i] = (D[i] / nrm).astype(np.float32)
    for which in (0, 1):
        g = r.trace_closest(O, D, which=which)
        c = o.trace_closest(O, D, which=which)
        for k in ("inst", "prim", "t", "u", "v"):
            assert_bit_equal(g[k], c[k], f"{scene_name} tlas{which} closest.{k}")
    tm = rng.uniform(0, 800, n).astype(np.float32)
    tm[:50] = np.nan
    tm[50:100] = np.inf
    tm[100:120] = 0.0
    assert np.array_equal(r.trace_any(O, D, tm), o.trace_any(O, D, tm))
    assert np.array_equal(r.trace_any(O, D, tm, which=1), o.trace_any(O, D, tm, which=1))
    # closest hit with a finite initial t_max (integrate always passes INFINITY, the hook accepts any)
    g = r.trace_closest(O, D, tm)
    c = o.trace_closest(O, D, tm)
    for k in ("inst", "prim", "t"):
        assert_bit_equal(g[k], c[k], f"closest with t_max .{k}")


def test_materials_match_oracle(api, oracle_mod):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Dielectric, GGX, Lambertian, Model, SceneDesc, Specular, Emissive
    sc = scenes.cornell_box(32, 32)
    kinds = [Lambertian.new((0.7, 0.6, 0.5)), Specular.new((0.9, 0.9, 0.8)), GGX.new_metal((0.1, 0.1, 0.45), 0.4),
             GGX.new_dielectric((0.95, 0.95, 0.95), 0.2, 1.5, None), GGX.new_dielectric((1, 1, 1), 0.0, 1.5, None), Dielectric.new((0.95, 0.95, 0.95), 1.5, None),
             GGX.new_metal((0.9, 0.5, 0.2), 1.0)]
    tri = sc.models[0]
    models = [Model.new(tri.positions, tri.normals, Emissive.new((1, 1, 1)))] + [Model.new(tri.positions + np.float32(10 * (i + 1)), tri.normals, m) for i, m in enumerate(kinds)]
    scene = SceneDesc.new(models, sc.camera)
    r = api.Renderer(scene, 32, 32)
    o = oracle_mod.Oracle(scene)
    mats = scene.materials()
    rng = np.random.default_rng(11)
    n = 3000
    nrm = rng.normal(size=(n, 3)); nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    inc = rng.normal(size=(n, 3)); inc = (inc / np.linalg.norm(inc, axis=1, keepdims=True)).astype(np.float32)
    flip = (inc * nrm).sum(1) > 0          # incoming direction points into the surface: dot(incoming, normal) < 0 after face-forwarding
    inc[flip] = -inc[flip]
    inc[:50] = -nrm[:50]                   # normal incidence
    nrm[50:60] = [0, 0, 1]; nrm[60:70] = [0, 0, -1]
    front = rng.integers(0, 2, n).astype(np.uint8)
    px = rng.integers(0, 1 << 20, n).astype(np.uint32); sm = rng.integers(0, 256, n).astype(np.uint32)
    for m in kinds:
        mi = mats.index(m)
        g = r.material_eval(mi, inc, nrm, front, px, sm, 3)
        c = np.stack([o.material_eval(mi, inc[i], nrm[i], front[i], int(px[i]), int(sm[i]), 3) for i in range(n)])
        assert_bit_equal(g, c, f"material kind {m.kind} roughness {m.roughness}")


def test_per_sample_radiance_bit_exact_vs_golden(api):
    from path_tracer_amd import scenes
    g = np.load(os.path.join(GOLD, "samples_small.npz"))
    r = api.Renderer(scenes.cornell_box(64, 64), 64, 64, max_bounces=8)
    assert_bit_equal(r.render_samples(0, 4), g["cornell64"], "cornell 64x64 per-sample radiance")
    r = api.Renderer(scenes.cornell_mixed(48, 48), 48, 48, max_bounces=8)
    assert_bit_equal(r.render_samples(0, 4), g["mixed48"], "mixed-material 48x48 per-sample radiance")


def test_config1_image_bit_exact(api, oracle_mod, cornell256):
    """BASELINE.json configs[0]: Cornell 256x256, 16 spp, depth 4 — checksum + crop from the fixture, then the live oracle."""
    g = np.load(os.path.join(GOLD, "cornell_c1.npz"))
    r = api.Renderer(cornell256, 256, 256, max_bounces=4, flags=api.FLAG_TIMING)
    acc, pos, idb = r.render(0, 16)
    assert hashlib.sha256(acc.tobytes()).digest() == g["sha256"].tobytes()
    assert_bit_equal(acc[96:160, 96:160], g["crop"], "C1 crop")
    assert_bit_equal(pos[96:160, 96:160], g["pos_crop"], "C1 first-hit position")
    assert np.array_equal(idb[96:160, 96:160], g["id_crop"])
    st = r.stats()
    assert (st.rays_closest, st.rays_any, st.rays_light_closest, st.paths) == tuple(int(x) for x in g["counters"][[0, 1, 2, 5]])
    o = oracle_mod.Oracle(cornell256)
    oacc, opos, oid, _ = o.render(256, 256, 16, max_bounces=4)
    assert_bit_equal(acc, oacc, "C1 image"); assert_bit_equal(pos, opos, "C1 position"); assert np.array_equal(idb, oid)


@pytest.mark.parametrize("kw", [dict(max_bounces=0), dict(max_bounces=1), dict(max_bounces=12), dict(enable_nee=False, max_bounces=6),
                                dict(max_bounces=40), dict(max_bounces=1024), dict(n_sobol=2, max_bounces=3), dict(seed=12345, max_bounces=5)])
def test_integrator_variants_bit_exact(api, oracle_mod, kw):
    from path_tracer_amd import scenes
    sc = scenes.cornell_mixed(40, 24)
    r = api.Renderer(sc, 40, 24, **kw)
    o = oracle_mod.Oracle(sc)
    okw = dict(kw)
    if "enable_nee" in okw:
        okw["enable_nee"] = int(okw["enable_nee"])
    want = o.render_samples(40, 24, 3, first_sample=3, **okw)
    assert_bit_equal(r.render_samples(3, 3), want, str(kw))
    # again in the same buffers: the per-bounce bookkeeping rows (cleared eagerly for the first 18 bounces, lazily beyond) must come up clean
    assert_bit_equal(r.render_samples(3, 3), want, str(kw) + " (second render)")


def test_batching_resume_and_sharding_do_not_change_the_image(api, oracle_mod):
    from path_tracer_amd import scenes
    from path_tracer_amd.dist import rows_of_rank
    sc = scenes.cornell_box(48, 30)
    ref = api.Renderer(sc, 48, 30, max_bounces=5).render(0, 6)
    small = api.Renderer(sc, 48, 30, max_bounces=5, batch_spp=1)
    a1 = small.render(0, 6)
    for x, y, w in zip(ref, a1, ("acc", "pos", "id")):
        assert_bit_equal(x, y, f"batch_spp=1 {w}")
    res = api.Renderer(sc, 48, 30, max_bounces=5, batch_spp=4)
    res.render(0, 2)
    idb = np.zeros((30, 48), np.uint32)
    res.reset_accumulation()
    res.render(0, 4, ident=idb)
    a2 = res.render(4, 2, ident=idb)
    assert_bit_equal(ref[0], a2[0], "resumed accumulation"); assert np.array_equal(ref[2], a2[2])
    # three ranks' row strips assembled == single-GPU frame (counter RNG is keyed by the global pixel)
    full = np.zeros_like(ref[0])
    for rank in range(3):
        rr = api.Renderer(sc, 48, 30, max_bounces=5, rank=rank, world_size=3, strip_rows=4)
        full[rows_of_rank(30, rank, 3, 4)] = rr.render(0, 6)[0]
    assert_bit_equal(ref[0], full, "row-sharded render")
    o = oracle_mod.Oracle(sc)
    assert_bit_equal(ref[0], o.render(48, 30, 6, max_bounces=5)[0], "vs oracle")


def test_pipelines_give_back_an_oversized_pool(api, oracle_mod):
    """One resident batch on pipeline 0, then the same context asked for small batches on two pipelines: pipeline 0 must not keep its
    large pool beside pipeline 1's new one (the budget is shared), and the image is still the oracle's."""
    from path_tracer_amd import scenes
    sc = scenes.cornell_box(64, 48)
    r = api.Renderer(sc, 64, 48, max_bounces=5)
    r.render(0, 16)
    r.set_config(batch_spp=2, pipelines=2)
    r.reset_accumulation()
    got = r.render(0, 16)
    fresh = api.Renderer(sc, 64, 48, max_bounces=5, batch_spp=2, pipelines=2)
    fresh.render(0, 16)
    assert r.stats().state_bytes == fresh.stats().state_bytes, (r.stats().state_bytes, fresh.stats().state_bytes)
    o = oracle_mod.Oracle(sc)
    assert_bit_equal(got[0], o.render(64, 48, 16, max_bounces=5)[0], "two small pipelines after one large batch")


def test_checkpoint_restores_into_a_fresh_context(api, oracle_mod):
    """SURVEY 5 checkpoint / resume: context A renders samples [0, 2) and is read back; a NEW context gets that state through
    pt_write_accumulation and renders sample 2: accumulation, position and the id history (which still holds sample 1) equal the
    oracle's samples [0, 3) bit for bit."""
    from path_tracer_amd import scenes
    sc = scenes.cornell_mixed(40, 26)
    a = api.Renderer(sc, 40, 26, max_bounces=6)
    acc, pos, idb = a.render(0, 2)
    a.close()
    b = api.Renderer(sc, 40, 26, max_bounces=6)
    b.write_accumulation(acc, pos, idb)
    b.render_device(2, 1)                      # (render() would upload its own id buffer: the restored one must do)
    got = b.read_frame()
    o = oracle_mod.Oracle(sc)
    oa, op, oi, _ = o.render(40, 26, 2, max_bounces=6)
    oa, op, oi, _ = o.render(40, 26, 1, first_sample=2, max_bounces=6, accum=oa, ident=oi)
    assert_bit_equal(got[0], oa, "restored accumulation + sample 2")
    assert_bit_equal(got[1], op, "position of sample 2")
    assert np.array_equal(got[2], oi), "id history across the restore"
    assert (got[0][..., 3] == 3).all()
    # only the accumulation restored: the frame still continues (id history then starts from zero)
    c = api.Renderer(sc, 40, 26, max_bounces=6)
    c.write_accumulation(acc)
    c.render_device(2, 1)
    assert_bit_equal(c.read_frame()[0], oa, "accumulation alone")


def test_empty_and_edge_inputs(api, cornell64):
    r = api.Renderer(cornell64, 64, 64)
    z = np.zeros((0, 3), np.float32)
    assert r.trace_closest(z, z)["t"].shape == (0,)
    acc, _, _ = r.render(0, 0)
    assert (acc == 0).all()
    one = api.Renderer(cornell64, 1, 1, max_bounces=2)
    assert one.render(0, 3)[0][0, 0, 3] == 3
    empty_rank = api.Renderer(cornell64, 8, 4, rank=1, world_size=2, strip_rows=4)   # a rank that owns no row
    assert empty_rank.render(0, 2)[0].shape == (0, 8, 4)


def test_large_frame_properties(api):
    """BASELINE-size frame (1920x1080): properties that need no oracle — every pixel got its samples, radiance finite and
    non-negative, escaped pixels carry exactly the ambient term, rendering twice gives the same bits."""
    from path_tracer_amd import scenes
    r = api.Renderer(scenes.cornell_box(1920, 1080), 1920, 1080, max_bounces=8)
    a, pos, idb = r.render(0, 2)
    assert (a[..., 3] == 2).all() and np.isfinite(a).all() and (a[..., :3] >= 0).all()
    miss = idb == ((255 << 16) | 255)
    assert miss.any() and (~miss).any()
    assert_bit_equal(a[miss][:, :3], np.full((miss.sum(), 3), np.float32(0.006) + np.float32(0.006), np.float32), "ambient")
    r.reset_accumulation()
    b, _, _ = r.render(0, 2)
    assert_bit_equal(a, b, "re-render")


def test_full_size_frame_bit_exact_vs_oracle(api, oracle_mod):
    """BASELINE.json configs[1] geometry at its full 1920x1080 and depth 8, a few samples per pixel: every word of the accumulated frame,
    first-hit position and id history against the oracle (multi-threaded; a second or so of host time on the GPU box), and the
    ray tallies of the reference's call sites"""
    import hashlib
    from path_tracer_amd import scenes
    W, H, SPP = 1920, 1080, 3
    sc = scenes.cornell_box(W, H)
    r = api.Renderer(sc, W, H, max_bounces=8)
    o = oracle_mod.Oracle(sc)
    acc, pos, idb = r.render(0, SPP)
    oacc, opos, oid, octr = o.render(W, H, SPP, max_bounces=8)
    assert_bit_equal(acc, oacc, "1080p accumulation"); assert_bit_equal(pos, opos, "1080p position"); assert np.array_equal(idb, oid)
    st = r.stats()
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))
    # sharded the way bench.py --gpus 8 shards it: the strips of all ranks reassemble to the same frame (checksum of checksums)
    from path_tracer_amd.dist import rows_of_rank
    whole = hashlib.sha256(acc.tobytes()).hexdigest()
    again = np.zeros_like(acc)
    for rank in range(8):
        rr = api.Renderer(sc, W, H, max_bounces=8, rank=rank, world_size=8, strip_rows=4)
        part, _, _ = rr.render(0, SPP)
        again[rows_of_rank(H, rank, 8, 4)] = part
        rr.close()
    assert hashlib.sha256(again.tobytes()).hexdigest() == whole


def test_bench_configuration_bit_exact_vs_oracle(api, oracle_mod):
    """BASELINE.json configs[1] exactly as bench.py times it: 1920x1080, 256 spp, depth 8, the default (auto) wavefront batch —
    every word of accumulation, first-hit position and id history, and the three ray tallies, against the oracle (about 20 s of
    host time on the GPU box's cores)."""
    from path_tracer_amd import scenes
    W, H, SPP = 1920, 1080, 256
    sc = scenes.cornell_box(W, H)
    r = api.Renderer(sc, W, H, max_bounces=8)
    acc, pos, idb = r.render(0, SPP)
    st = r.stats()
    r.close()
    oacc, opos, oid, octr = oracle_mod.Oracle(sc).render(W, H, SPP, max_bounces=8)
    assert_bit_equal(acc, oacc, "1080p x 256 spp accumulation"); assert_bit_equal(pos, opos, "position"); assert np.array_equal(idb, oid)
    assert (st.rays_closest, st.rays_any, st.rays_light_closest, st.paths) == (int(octr[0]), int(octr[1]), int(octr[2]), W * H * SPP)


@pytest.mark.parametrize("name,kw", [("cornell_mixed", {}), ("cornell_spheres", dict(level=4))])
def test_config5_parameters_bit_exact(api, oracle_mod, name, kw):
    """BASELINE.json configs[4] at its real parameters: 4096x4096, depth 16, a 4096-spp Sobol table (N = 2 * spp = 8192), the LAST two
    samples [4094, 4096), rendered in two wavefront batches; mixed materials (diffuse + dielectric + GGX metal [+ mirror, instanced])."""
    from path_tracer_amd import scenes
    W = H = 4096
    sc = getattr(scenes, name)(W, H, **kw)
    r = api.Renderer(sc, W, H, max_bounces=16, n_sobol=8192, batch_spp=1)
    acc, pos, idb = r.render(4094, 2)
    st = r.stats()
    r.close()
    oacc, opos, oid, octr = oracle_mod.Oracle(sc).render(W, H, 2, first_sample=4094, max_bounces=16, n_sobol=8192)
    assert_bit_equal(acc, oacc, f"{name} 4096^2 depth 16 accumulation"); assert_bit_equal(pos, opos, "position"); assert np.array_equal(idb, oid)
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))


@pytest.mark.parametrize("level", [6, 7])
def test_mesh_scenes_full_frame_bit_exact(api, oracle_mod, level):
    """BASELINE.json configs[2]/[3] class (82 k / 328 k triangles, BVH in HBM / L2) at the full 1920x1080, depth 8, one sample"""
    from path_tracer_amd import scenes
    W, H = 1920, 1080
    sc = scenes.cornell_mesh(W, H, level=level)
    r = api.Renderer(sc, W, H, max_bounces=8)
    acc, pos, idb = r.render(0, 1)
    st = r.stats()
    r.close()
    oacc, opos, oid, octr = oracle_mod.Oracle(sc).render(W, H, 1, max_bounces=8)
    assert_bit_equal(acc, oacc, f"mesh level {level} 1080p accumulation"); assert_bit_equal(pos, opos, "position"); assert np.array_equal(idb, oid)
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))


def test_atrium_full_frame_bit_exact(api, oracle_mod):
    """BASELINE.json configs[3] as SURVEY 8(d) fixes it — the instanced atrium (scenes.atrium: 293 BLASes, ~590 TLAS leaves, quarter turns,
    glam-built general rotations, ~250 k instanced triangles) — at the full 1920x1080, depth 8: accumulation, first-hit position, id
    history, ray tallies.  The first-hit id is the BLAS arena index `as u8` (integrator.rs:184, main.rs:206): with more than 256 models it
    WRAPS, and the statues nearest the camera are models 256..292."""
    from path_tracer_amd import scenes
    W, H = 1920, 1080
    sc = scenes.atrium(W, H)
    assert len(sc.models) > 256 and sum(len(m.matrices) for m in sc.models) > 500
    r = api.Renderer(sc, W, H, max_bounces=8)
    o = oracle_mod.Oracle(sc)
    for which in (0, 1):
        a, b = r.tlas_dump(which), o.tlas_dump(which)
        for k in a:
            assert_bit_equal(np.asarray(a[k]), np.asarray(b[k]), f"tlas{which}.{k}")
    acc, pos, idb = r.render(0, 1)
    st = r.stats()
    assert st.stack_entries > 14 and not st.lds_scene
    oacc, opos, oid, octr = o.render(W, H, 1, max_bounces=8)
    assert_bit_equal(acc, oacc, "atrium 1080p accumulation"); assert_bit_equal(pos, opos, "position"); assert np.array_equal(idb, oid)
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))
    # the wrap is really exercised: camera rays whose first hit is a model of index >= 256 (their id is index - 256)
    rng = np.random.default_rng(5)
    px = rng.integers(0, W * H, 20000)
    ro = np.zeros((len(px), 3), np.float32); rd = np.zeros((len(px), 3), np.float32)
    for i, p in enumerate(px):
        ro[i], rd[i] = o.primary_ray(W, H, int(p), 0)
    h = r.trace_closest(ro, rd)
    c = o.trace_closest(ro, rd)
    for k in ("inst", "prim", "t", "u", "v"):
        assert_bit_equal(h[k], c[k], f"atrium camera rays closest.{k}")
    model_of_instance = np.concatenate([np.full(len(m.matrices), i) for i, m in enumerate(sc.models)])   # TLAS leaves in model order (tlas_bvh.rs:90-103)
    hit = h["inst"] != 0xFFFFFFFF
    first_model = model_of_instance[h["inst"][hit]]
    assert (first_model >= 256).sum() > 50
    wrapped = (idb.reshape(-1)[px[hit]] & 0xFFFF) == (first_model & 0xFF)
    assert wrapped.all()


def test_spilling_stacks_do_not_race_between_concurrent_launches(api, oracle_mod):
    """With only two stack levels in LDS every deeper level of every traversal lives in global memory; the shadow-ray launch and
    the BSDF-sampled NEE launch run side by side on two streams and must not share those slots (large lights so that the NEE
    launch is not empty; enough rays that the two kernels really overlap)."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Emissive, Model, SceneDesc
    W, H = 640, 360
    base = scenes.cornell_mesh(W, H, level=5)
    # a second, huge emitter: the right wall's twin, slightly inside it, so that most BSDF-sampled NEE rays pass the lights' root box
    wall = scenes.cornell_models()[2]
    big = Model.new(wall.positions * np.float32(0.98), wall.normals, Emissive.new((2.0, 1.5, 1.0)), None, "big_light")
    sc = SceneDesc.new(list(base.models) + [big], base.camera, "spill_race")
    r = api.Renderer(sc, W, H, max_bounces=6, stack_lds_levels=2)
    g = r.render_samples(0, 2)
    assert r.stats().stack_entries > 2
    c = oracle_mod.Oracle(sc).render_samples(W, H, 2, max_bounces=6)
    assert_bit_equal(g, c, "spill-everything traversal with concurrent NEE launches")


def test_full_queue_is_reported_not_overrun(api, oracle_mod):
    """pt_config.queue_slack in test mode: the ray / shade queues hold only half as many slots as the batch has paths, in a
    scene where two thirds of the camera rays hit a surface and go on, so reservations find their queue full.  The producers
    divert to the queue's dump area and raise the overflow flag; the render returns PT_ERR_LIMIT; a context next to it (whose
    buffers are neighbours in HBM) still renders the right image; and the same context works again once the slack is restored."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Camera, SceneDesc
    sc = SceneDesc.new(scenes.cornell_models(), Camera.new((0.0, 50.0, 700.0), (0.0, 50.0, 0.0), 60.0, 1.0), "close")
    good = api.Renderer(sc, 256, 256, max_bounces=6)
    want = good.render_samples(0, 4)
    tight = api.Renderer(sc, 256, 256, max_bounces=6, queue_slack=0x80000000 | 512)
    with pytest.raises(api.PtError) as e:
        tight.render(0, 16)
    assert e.value.code == -5 and "queue" in str(e.value)
    assert not tight.read_accumulation().any(), "a failed render leaves the accumulation reset, not polluted with incomplete samples"
    assert_bit_equal(good.render_samples(0, 4), want, "neighbouring context after the overflow")
    tight.set_config(queue_slack=0)
    assert_bit_equal(tight.render_samples(0, 4), want, "same context with the default slack")


@pytest.mark.parametrize("flags", [0, 2])   # 2 = PT_FLAG_NO_LDS_SCENE: the same scene with its BVH in global memory
def test_ray_queues_are_dense_and_shade_queues_nearly(api, flags):
    """A shading workgroup reserves exactly what it appends: a bounce's ray-queue extents (slots) equal the rays traced from
    them (the claim-cursor tallies), nothing is a hole.  The shade queues are written in regions by the traversal waves over up to 64
    striped tails: their extents may exceed the hits by the waves' last regions and the stripes' gaps, a few per cent on a queue this
    size (holes were measured to cost more than reservations: DESIGN.md section 4).
    Shadow rays: with the BVH in LDS the Lambertian shading pass answers them itself (nothing is queued, the tally of traced rays is
    the pass's own); with the BVH in global memory they are queued for k_any and the extent equals the tally.  Both runs cast the same rays."""
    from path_tracer_amd import scenes
    r = api.Renderer(scenes.cornell_box(960, 540), 960, 540, max_bounces=6, flags=flags)
    r.render_device(0, 8)
    r.synchronize()
    rows = r.last_batch_counters().astype(np.int64)
    assert rows[0][0] == rows[0][13] > 500_000          # camera rays: extent == traced
    for b in range(1, 7):
        n_closest, n_shadow, n_lchain = rows[b][0], rows[b - 1][2], rows[b - 1][4]
        assert n_closest == rows[b][13], (b, n_closest, rows[b][13])              # continuation rays of bounce b
        assert rows[b - 1][14] > 0
        assert n_shadow == (rows[b - 1][14] if flags else 0), (b, n_shadow, rows[b - 1][14])   # shadow rays cast by bounce b-1's shading
        assert n_lchain == rows[b - 1][6], (b, n_lchain, rows[b - 1][6])          # BSDF-sampled NEE rays that passed the lights' root box
        lambert_slots = rows[b][9]
        assert n_closest * 0.5 < lambert_slots <= n_closest * 1.06 + 4096 * 64, (b, lambert_slots, n_closest)
    if flags:
        return
    # the two routes cast the same shadow rays
    r2 = api.Renderer(scenes.cornell_box(960, 540), 960, 540, max_bounces=6, flags=2)
    r2.render_device(0, 8)
    r2.synchronize()
    rows2 = r2.last_batch_counters().astype(np.int64)
    assert [int(x) for x in rows[:7, 14]] == [int(x) for x in rows2[:7, 14]]


def test_scene_edit_adds_a_material_class(api, oracle_mod):
    """render, add a model of a material class the context has not seen (its shade queue does not exist yet), rebuild, render again"""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import GGX, Model, SceneDesc, Specular
    base = scenes.cornell_box(64, 48)
    r = api.Renderer(base, 64, 48, max_bounces=6)
    r.render(0, 2)
    t, n = scenes.sphere_mesh(2, (0.0, 120.0, 60.0), 70.0)
    extra = [Model.new(t.astype(np.float32), n.astype(np.float32), Specular.new((0.9, 0.9, 0.9)), None, "mirror"),
             Model.new((t + np.float64(1.0)).astype(np.float32) * np.float32(0.5), n.astype(np.float32), GGX.new_metal((0.8, 0.6, 0.2), 0.3), None, "metal")]
    for m in extra:
        r.add_model(m)
    r.rebuild()
    g = r.render_samples(0, 3)
    sc2 = SceneDesc.new(list(base.models) + extra, base.camera, "edited")
    assert_bit_equal(g, oracle_mod.Oracle(sc2).render_samples(64, 48, 3, max_bounces=6), "scene after adding specular + GGX models")


@pytest.mark.parametrize("name,kw,flags", [
    ("cornell_box", {}, 2),                      # PT_FLAG_NO_LDS_SCENE: same scene, BVH read from global memory
    ("cornell_spheres", dict(level=3), 0),       # 4.5 k triangles, five material kinds, rotated instance
    ("cornell_mesh", dict(level=4), 0),          # 5 k triangles
    ("cornell_mesh", dict(level=6), 0),          # 82 k triangles (configs[2] class): deep BLAS, BVH in HBM/L2
    ("cornell_mesh", dict(level=7), 0),          # 328 k triangles (configs[3] class): traversal stack spills past its LDS levels
])
def test_larger_scenes_bit_exact(api, oracle_mod, name, kw, flags):
    from path_tracer_amd import scenes
    sc = getattr(scenes, name)(64, 36, **kw)
    r = api.Renderer(sc, 64, 36, max_bounces=6, flags=flags)
    o = oracle_mod.Oracle(sc)
    assert_bit_equal(r.render_samples(0, 3), o.render_samples(64, 36, 3, max_bounces=6), f"{name} per-sample radiance")
    st = r.stats()
    assert st.lds_scene == (1 if (name == "cornell_box" and not flags) else 0) or st.scene_bytes <= 48 * 1024
    rng = np.random.default_rng(9)
    n = 4000
    O = rng.uniform(-250, 250, (n, 3)).astype(np.float32)
    D = rng.normal(size=(n, 3)); D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    g, c = r.trace_closest(O, D), o.trace_closest(O, D)
    for k in ("inst", "prim", "t", "u", "v"):
        assert_bit_equal(g[k], c[k], f"{name} closest.{k}")
    tm = rng.uniform(0, 600, n).astype(np.float32)
    assert np.array_equal(r.trace_any(O, D, tm), o.trace_any(O, D, tm))


def test_participating_media_bit_exact(api, oracle_mod):
    """material/volume.rs + integrator.rs:189-205,217-227: Henyey-Greenstein free-flight scattering, Beer-Lambert absorption,
    nested volumes, emissive hits shaded after the media."""
    from path_tracer_amd import scenes
    sc = scenes.cornell_media(56, 40, level=3)
    o = oracle_mod.Oracle(sc)
    for kw in (dict(max_bounces=12), dict(max_bounces=5, enable_nee=False)):
        r = api.Renderer(sc, 56, 40, **kw)
        okw = dict(kw)
        if "enable_nee" in okw:
            okw["enable_nee"] = int(okw["enable_nee"])
        g = r.render_samples(0, 6)
        c = o.render_samples(56, 40, 6, **okw)
        assert_bit_equal(g, c, f"media {kw}")
        assert np.isfinite(g).all() and (g[..., :3].max() > 0.01)
    acc, pos, idb = api.Renderer(sc, 56, 40, max_bounces=12).render(0, 3)
    oacc, opos, oid, _ = o.render(56, 40, 3, max_bounces=12)
    assert_bit_equal(acc, oacc, "media frame"); assert_bit_equal(pos, opos, "media position"); assert np.array_equal(idb, oid)


def test_environment_map_bit_exact(api, oracle_mod):
    """integrator.rs:256-262 + image_helper.rs:61-88: equirect bilinear lookup on a miss (deterministic atan2/asin)."""
    from path_tracer_amd import scenes
    rng = np.random.default_rng(21)
    env = (rng.uniform(0, 1, (17, 33, 3)) ** 3 * 4).astype(np.float32)
    sc = scenes.cornell_mixed(48, 32)
    r = api.Renderer(sc, 48, 32, max_bounces=6)
    o = oracle_mod.Oracle(sc)
    base = r.render_samples(0, 2)
    r.set_environment(env); o.set_environment(env)
    g = r.render_samples(0, 3)
    assert_bit_equal(g, o.render_samples(48, 32, 3, max_bounces=6), "environment-lit samples")
    assert not np.array_equal(g[:2], base)
    acc, pos, idb = r.render(0, 2)
    oacc, opos, oid, _ = o.render(48, 32, 2, max_bounces=6)
    assert_bit_equal(acc, oacc, "env frame"); assert_bit_equal(pos, opos, "env position"); assert np.array_equal(idb, oid)
    r.set_environment(None)
    assert_bit_equal(r.render_samples(0, 2), base, "environment removed")
    # device atan2 / asin agree with the host routines on the whole plane
    y = rng.normal(size=20000).astype(np.float32); x = rng.normal(size=20000).astype(np.float32)
    y[:4] = [0, -0.0, 1, -1]; x[:4] = [-1, -1, 0, -0.0]
    assert_bit_equal(r.math_batch(8, y, x)[0], oracle_mod.math_batch(8, y, x)[0], "atan2")
    v = rng.uniform(-1.001, 1.001, 20000).astype(np.float32)
    assert_bit_equal(r.math_batch(9, v)[0], oracle_mod.math_batch(9, v)[0], "asin")


def test_obj_model_renders_like_its_triangle_soup(api, oracle_mod, tmp_path):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import GGX, Model, SceneDesc
    from test_host import OBJ_TEXT
    path = tmp_path / "thing.obj"
    path.write_text(OBJ_TEXT)
    room = scenes.cornell_models()[:4]
    sc = SceneDesc.new(room + [Model.from_obj(str(path), GGX.new_metal((0.8, 0.6, 0.2), 0.3))], scenes.reference_camera(1.5))
    r = api.Renderer(sc, 48, 32, max_bounces=5)
    o = oracle_mod.Oracle(sc)
    assert_bit_equal(r.render_samples(0, 3), o.render_samples(48, 32, 3, max_bounces=5), "scene with an OBJ-loaded model")


def test_oversized_leaf_bit_exact(api, oracle_mod):
    """100 coincident triangles end up in ONE leaf (splitting them never pays, blas_bvh.rs:112-122): more than a leaf link can
    count, so the device BVH refers to it through the big-leaf table.  Hits, leaf order and the image must not care."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Lambertian, Model, SceneDesc
    tri = np.array([[[-60.0, 40.0, 30.0], [70.0, 45.0, 20.0], [0.0, 150.0, -10.0]]], np.float32)
    pos = np.repeat(tri, 100, axis=0)
    nrm = np.repeat(np.array([[[0.0, 0.0, 1.0]] * 3], np.float32), 100, axis=0)
    sc = SceneDesc.new(scenes.cornell_models() + [Model.new(pos, nrm, Lambertian.new((0.2, 0.3, 0.8)))], scenes.reference_camera(1.0))
    r = api.Renderer(sc, 32, 32, max_bounces=4); o = oracle_mod.Oracle(sc)
    d = r.blas_dump(r.blas_count() - 1)
    assert len(d["kind"]) == 1 and d["kind"][0] == 1 and d["b"][0] == 100        # one leaf holding all hundred
    rng = np.random.default_rng(11)
    n = 4000
    O = np.tile(np.array([0.0, 50.0, 600.0], np.float32), (n, 1)) + rng.normal(size=(n, 3)).astype(np.float32) * 40
    T = np.stack([rng.uniform(-80, 90, n), rng.uniform(20, 170, n), rng.uniform(-20, 40, n)], 1).astype(np.float32)
    D = T - O
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    g = r.trace_closest(O, D); c = o.trace_closest(O, D)
    for k in ("inst", "prim", "t", "u", "v"):
        assert_bit_equal(g[k], c[k], f"big leaf closest.{k}")
    assert (g["inst"] == g["inst"].max()).sum() > 500                            # plenty of rays end on the stack of triangles
    tm = rng.uniform(100, 900, n).astype(np.float32)
    assert np.array_equal(r.trace_any(O, D, tm), o.trace_any(O, D, tm))
    assert_bit_equal(r.render_samples(0, 2), o.render_samples(32, 32, 2, max_bounces=4), "scene with a 100-triangle leaf")


def test_framebuffer_is_visible_to_torch_in_place(api, cornell64):
    """the multi-GPU gather reads libptmi's accumulation buffer through __cuda_array_interface__ (no host round trip)"""
    import torch
    from path_tracer_amd.dist import gather_framebuffer, wrap_device_framebuffer
    r = api.Renderer(cornell64, 64, 64, max_bounces=3)
    stream = torch.cuda.current_stream()
    r.set_stream(stream.cuda_stream)
    acc, _, _ = r.render(0, 2)
    ptr, n = r.accum_device_ptr()
    assert n == 64 * 64
    t = wrap_device_framebuffer(ptr, 64, 64, torch.device("cuda", 0))
    torch.cuda.synchronize()
    assert_bit_equal(t.cpu().numpy(), acc, "device view of the framebuffer")
    assert gather_framebuffer(t, 64, 64, 0, 1, 4) is t
    r.set_stream(None)


@pytest.mark.parametrize("eye,target,fov,culls", [
    ((0.0, 50.0, 1000.0), (0.0, 50.0, 0.0), 60.0, True),      # the reference camera: the room fills a third of the frame
    ((900.0, 700.0, 1100.0), (0.0, 0.0, 0.0), 35.0, True),    # from a corner: the box's image is a hexagon
    ((0.0, 50.0, 1000.0), (900.0, 50.0, 0.0), 40.0, True),    # the room half out of the frame
    ((0.0, 50.0, 1000.0), (0.0, 420.0, 0.0), 20.0, True),     # the room entirely out of the (narrow) frame: no path at all
    ((0.0, 50.0, 1000.0), (0.0, 3000.0, 0.0), 30.0, False),   # looking past it steeply: part of the box is behind the image plane, no rectangle bounds it
    ((0.0, 50.0, 100.0), (0.0, 50.0, 0.0), 60.0, False)])     # inside the room: nothing can be culled
def test_camera_rays_outside_the_scene_bounds_are_not_generated(api, oracle_mod, eye, target, fov, culls):
    """Pixels whose every camera ray misses the world's root box (the host projects the box onto the image plane) get the miss result
    without a path.  Same frame, positions, ids and ray tallies as with PT_FLAG_NO_PRIMARY_CULL and as the oracle, whatever the view;
    sharded rows included."""
    from path_tracer_amd import scenes
    from path_tracer_amd.dist import rows_of_rank
    from path_tracer_amd.scene_desc import Camera, SceneDesc
    W, H = 160, 90
    sc = SceneDesc.new(scenes.cornell_models(), Camera.new(eye, target, fov, W / H), "view")
    a = api.Renderer(sc, W, H, max_bounces=5)
    b = api.Renderer(sc, W, H, max_bounces=5, flags=api.FLAG_NO_PRIMARY_CULL)
    ra, rb = a.render(2, 3), b.render(2, 3)
    for x, y, what in zip(ra, rb, ("accumulation", "position", "id")):
        assert_bit_equal(x, y, f"culled vs not: {what}")
    sa, sb = a.stats(), b.stats()
    assert (sa.rays_closest, sa.rays_any, sa.rays_light_closest, sa.paths) == (sb.rays_closest, sb.rays_any, sb.rays_light_closest, sb.paths)
    assert sb.rays_primary_culled == 0 and (sa.rays_primary_culled > 0) == culls
    o = oracle_mod.Oracle(sc).render(W, H, 3, first_sample=2, max_bounces=5)
    assert_bit_equal(ra[0], o[0], "vs oracle"); assert_bit_equal(ra[1], o[1], "position vs oracle"); assert np.array_equal(ra[2], o[2])
    assert_bit_equal(a.render_samples(0, 2), b.render_samples(0, 2), "per-sample radiance")
    full = np.zeros_like(rb[1])
    for rank in range(3):
        rr = api.Renderer(sc, W, H, max_bounces=5, rank=rank, world_size=3, strip_rows=4)
        full[rows_of_rank(H, rank, 3, 4)] = rr.render(2, 3)[1]
    assert_bit_equal(full, rb[1], "first-hit positions of three ranks' strips")


def test_multi_device_context_through_the_c_abi(api, oracle_mod):
    """pt_multi: one process driving N contexts.  (a) one device through RCCL (a one-rank communicator; ncclGather to itself),
    (b) three contexts SHARING the device: strips of every rank gathered by device copies and de-interleaved on the GPU.  Both must
    reproduce the single-context frame bit for bit (the RNG is keyed by the global pixel), odd height so that ranks own unequal rows."""
    from path_tracer_amd import scenes
    W, H = 96, 70
    sc = scenes.cornell_mixed(W, H)
    want = oracle_mod.Oracle(sc).render(W, H, 5, max_bounces=6)[0]
    one = api.MultiRenderer(sc, W, H, [0], max_bounces=6)
    got = one.render(0, 5)
    assert one.used_rccl(), "a one-device pt_multi gathers through RCCL"
    assert_bit_equal(got, want, "pt_multi, 1 device, RCCL gather")
    st = one.stats()
    one.close()
    three = api.MultiRenderer(sc, W, H, [0, 0, 0], max_bounces=6, strip_rows=4)
    three.render(0, 2, download=False)
    got3 = three.render(2, 3)
    assert not three.used_rccl()
    assert_bit_equal(got3, want, "pt_multi, three contexts on one device, resumed accumulation")
    st3 = three.stats()
    assert (st3.rays_closest, st3.rays_any, st3.rays_light_closest, st3.paths) == (st.rays_closest, st.rays_any, st.rays_light_closest, st.paths)
    three.close()


def test_gather_pad_and_deinterleave_run_on_device_tensors(api, cornell64):
    """path_tracer_amd.dist: the padding of unequal strip sets and the de-interleave of gathered strips, on DEVICE tensors that view
    the framebuffers of three contexts (world_size 3 emulated in one process; the collective itself is torch.distributed's)."""
    import torch
    from path_tracer_amd import dist as ptdist
    W, H, world, strip = 64, 50, 3, 4
    ref = api.Renderer(cornell64, W, H, max_bounces=4).render(0, 3)[0]
    dev = torch.device("cuda", 0)
    parts = []
    keep = []
    for rank in range(world):
        r = api.Renderer(cornell64, W, H, max_bounces=4, rank=rank, world_size=world, strip_rows=strip)
        r.render_device(0, 3); r.synchronize()
        ptr, n = r.accum_device_ptr()
        rows = len(r.local_rows())
        assert n == rows * W
        local = ptdist.wrap_device_framebuffer(ptr, rows, W, dev)
        parts.append(ptdist.pad_strips(local, H, W, world, strip).clone())
        keep.append(r)
    assert len({tuple(p.shape) for p in parts}) == 1 and parts[0].is_cuda
    full = ptdist.assemble_strips(parts, H, W, world, strip)
    assert full.is_cuda
    assert_bit_equal(full.cpu().numpy(), ref, "strips of three ranks padded and de-interleaved on the device")
    # what gather_framebuffer does with them on rank 0: ONE receive buffer whose chunks are the gather list, one index_select
    plan = ptdist._plan(H, W, world, strip, parts[0].dtype, dev, True)
    for dst_chunk, p in zip(plan.parts, parts):
        dst_chunk.copy_(p)  # stands in for the collective
    assert_bit_equal(plan.recv.index_select(0, plan.perm).cpu().numpy(), ref, "one receive buffer + row permutation")


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_scenes_bit_exact(api, oracle_mod, seed):
    """fuzz: several lights, instanced transforms (quarter turns, glam-built rotations about general axes, the reference's own
    from_rotation_y(PI)), every material kind, nested media"""
    from path_tracer_amd import scenes
    sc = scenes.random_scene(seed, 48, 32, with_media=(seed % 2 == 0))
    assert any(np.count_nonzero(m[:, :3]) > 3 for mo in sc.models for m in mo.matrices)
    o = oracle_mod.Oracle(sc)
    r = api.Renderer(sc, 48, 32, max_bounces=10)
    _cmp_host = lambda a, b: all(np.array_equal(np.atleast_1d(np.asarray(a[k], np.float32 if isinstance(a[k], float) else None)).view(np.uint32),
                                                np.atleast_1d(np.asarray(b[k], np.float32 if isinstance(b[k], float) else None)).view(np.uint32)) for k in a)
    assert _cmp_host(r.light_cdf(), o.light_cdf()) and _cmp_host(r.tlas_dump(0), o.tlas_dump(0)) and _cmp_host(r.tlas_dump(1), o.tlas_dump(1))
    assert _cmp_host(r.tlas_instances(0), o.tlas_instances(0)) and _cmp_host(r.tlas_instances(1), o.tlas_instances(1))
    g = r.render_samples(0, 4)
    c = o.render_samples(48, 32, 4, max_bounces=10)
    assert_bit_equal(g, c, f"random scene {seed}")
    acc, pos, idb = r.render(4, 3)
    oacc, opos, oid, octr = o.render(48, 32, 3, first_sample=4, max_bounces=10)
    assert_bit_equal(acc, oacc, "frame"); assert_bit_equal(pos, opos, "position"); assert np.array_equal(idb, oid)
    st = r.stats()
    # ray tallies of the second call only are not separable; compare after a reset
    r.reset_stats(); r.reset_accumulation(); r.render(4, 3)
    st = r.stats()
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))


def test_general_rigid_instances_bit_exact(api, oracle_mod):
    """Every instance matrix of rounds 1-3 was a signed permutation: M3 * v had no inexact product, so no test could see a wrong operation
    order in Ray::transform (ray.rs:22-28 -> to_object), Affine3A::inverse (tlas_bvh.rs:99), the normal transform (tlas.rs:105) or the
    corner-only AABB::transform (boundingbox.rs:51-57).  scenes.cornell_instanced places its models by glam-built rotations about general
    axes and by the reference's own from_rotation_translation(from_rotation_y(PI), (0, 200, 0)) (main.rs:97-113: +-8.742278e-8 off the
    diagonal).  Kernels against the committed fixture (rays aimed at the instances, both TLASes, any-hit, per-sample radiance, position,
    id, tallies), then against the live oracle on a larger frame."""
    from path_tracer_amd import scenes
    g = np.load(os.path.join(GOLD, "instanced.npz"))
    W, H = 48, 32
    sc = scenes.cornell_instanced(W, H)
    r = api.Renderer(sc, W, H, max_bounces=7)
    for which in (0, 1):
        ti = r.tlas_instances(which)
        assert_bit_equal(ti["matrix"], g[f"matrix{which}"], "matrix"); assert_bit_equal(ti["inv_matrix"], g[f"inv_matrix{which}"], "inv_matrix")
        assert_bit_equal(r.tlas_dump(which)["boxes"], g[f"tlas{which}_boxes"], "TLAS boxes (corner-only AABB::transform)")
        h = r.trace_closest(g["ray_o"], g["ray_d"], which=which)
        for k in ("t", "u", "v", "inst", "prim"):
            assert_bit_equal(h[k], g[f"hit{which}_{k}"], f"tlas{which} closest.{k}")
    assert (np.isin(g["hit0_inst"], [6, 7, 8, 9, 10, 12]) & np.isfinite(g["hit0_t"])).sum() > 1000   # hits seen through general rotations
    assert np.array_equal(r.trace_any(g["ray_o"], g["ray_d"], g["any_tmax"]), g["any_hit"])
    assert_bit_equal(r.render_samples(0, 3), g["samples"], "per-sample radiance")
    r.reset_stats(); r.reset_accumulation()
    acc, pos, idb = r.render(0, 3)
    assert_bit_equal(pos, g["position"], "first-hit position"); assert np.array_equal(idb, g["id"])
    st = r.stats()
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == tuple(int(x) for x in g["counters"][:3])
    # live oracle: a frame large enough for dynamic claims and striped tails, with the BVH read from global memory as well
    W, H = 320, 200
    sc = scenes.cornell_instanced(W, H, level=3)
    o = oracle_mod.Oracle(sc)
    want = o.render_samples(W, H, 2, max_bounces=9)
    oacc, opos, oid, octr = o.render(W, H, 2, max_bounces=9)
    for flags in (0, api.FLAG_NO_LDS_SCENE):
        r = api.Renderer(sc, W, H, max_bounces=9, flags=flags)
        assert_bit_equal(r.render_samples(0, 2), want, f"320x200 per-sample radiance (flags {flags})")
        r.reset_stats(); r.reset_accumulation()
        acc, pos, idb = r.render(0, 2)
        assert_bit_equal(acc, oacc, "frame"); assert_bit_equal(pos, opos, "position"); assert np.array_equal(idb, oid)
        st = r.stats()
        assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))
        rng = np.random.default_rng(21)
        n = 20000
        O = rng.uniform(-270, 270, (n, 3)).astype(np.float32); O[:, 1] += 50
        D = rng.normal(size=(n, 3)); D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
        D[:200] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 200)] * rng.choice([-1.0, 1.0], (200, 1)).astype(np.float32)   # zero components: rotated, they stop being zero
        for which in (0, 1):
            a, b = r.trace_closest(O, D, which=which), o.trace_closest(O, D, which=which)
            for k in ("inst", "prim", "t", "u", "v"):
                assert_bit_equal(a[k], b[k], f"random rays tlas{which} closest.{k}")
        tm = rng.uniform(0, 800, n).astype(np.float32)
        assert np.array_equal(r.trace_any(O, D, tm), o.trace_any(O, D, tm))


def test_atrium_with_every_stack_level_spilled(api, oracle_mod):
    """The atrium's traversal stack is deeper than the 14 levels kept in LDS (TLAS depth 15 + BLAS depth): with only TWO levels in LDS nearly every
    push and pop of both traversal kernels goes through the global spill area, TLAS entries lying under a BLAS's among them; per-sample radiance
    of a small frame (all four surface classes are present: the statues)."""
    from path_tracer_amd import scenes
    W, H = 240, 136
    sc = scenes.atrium(W, H)
    r = api.Renderer(sc, W, H, max_bounces=8, stack_lds_levels=2)
    g = r.render_samples(0, 3)
    assert r.stats().stack_entries > 14
    c = oracle_mod.Oracle(sc).render_samples(W, H, 3, max_bounces=8)
    assert_bit_equal(g, c, "atrium, spill-everything traversal")


@pytest.mark.parametrize("scale", [1e-18, 1e-3, 1.0, 1e9, 1e12, 1e13, 1e17])
def test_degenerate_triangles_and_extreme_scales(api, oracle_mod, scale):
    """Triangles the OBJ files of the world contain and the reference neither filters nor guards against: zero-area ones (two equal vertices, three
    collinear ones: n0 = 0, so the Havel-Herout planes n1, n2 are 0 / 0 = NaN, primitive.rs:31-54), slivers, and — through the scale — coordinates whose
    products leave the binary32 range at either end (denormal plane terms at 1e-18, 1e34-sized dot products at 1e17: infinities and 0 * inf in the
    slab tests).  Whatever the reference's arithmetic makes of them, both sides must make the same of them: hit records and any-hit answers as bit patterns."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Emissive, Lambertian, Model, SceneDesc
    rng = np.random.default_rng(23)
    n = 400
    c = rng.uniform(-100, 100, (n, 1, 3))
    p = c + rng.normal(0, 12.0, (n, 3, 3))
    p[0:40, 1] = p[0:40, 0]                                                       # two equal vertices
    p[40:80, 2] = p[40:80, 0] + 2.5 * (p[40:80, 1] - p[40:80, 0])                 # collinear
    p[80:120, 2] = p[80:120, 0] + (p[80:120, 1] - p[80:120, 0]) * 0.5 + rng.normal(0, 1e-4, (40, 3))   # slivers
    p[120:130] = p[120:121]                                                       # ten coincident copies of one triangle
    nr = rng.normal(size=(n, 3, 3)); nr /= np.linalg.norm(nr, axis=2, keepdims=True)
    s = np.float32(scale)
    pos = (p.astype(np.float32) * s).astype(np.float32)
    assert np.isfinite(pos).all()
    light = (np.array([[[-30, 150, -30], [30, 150, -30], [30, 150, 30]]], np.float32) * s).astype(np.float32)
    ln = np.tile(np.array([0, -1, 0], np.float32), (1, 3, 1))
    sc = SceneDesc.new([Model.new(pos, nr.astype(np.float32), Lambertian.new((0.6, 0.6, 0.6)), None, "soup"),
                        Model.new(light, ln, Emissive.new((5, 5, 5)), None, "light")], scenes.reference_camera(1.0))
    r = api.Renderer(sc, 32, 32)
    o = oracle_mod.Oracle(sc)
    a, b = r.blas_dump(0), o.blas_dump(0)
    for k in a:
        assert_bit_equal(np.asarray(a[k]), np.asarray(b[k]), f"blas.{k} at scale {scale}")
    m = 8000
    O = (rng.uniform(-160, 160, (m, 3)) * scale).astype(np.float32)
    D = rng.normal(size=(m, 3)); D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    tgt = (p[rng.integers(0, n, m // 2)].mean(1) * scale)                         # half of the rays aimed at triangle centroids (degenerate ones included)
    d2 = tgt - O[: m // 2].astype(np.float64)
    D[: m // 2] = (d2 / np.maximum(np.linalg.norm(d2, axis=1, keepdims=True), 1e-300)).astype(np.float32)
    D[~np.isfinite(D).all(1)] = np.array([0, 0, 1], np.float32)
    # rays that are not numbers: infinite / NaN origin components, zero and NaN direction components (Ray::new divides by them, ray.rs:16)
    special = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0], np.float32)
    for i in range(m - 60, m):
        if i % 2: O[i, rng.integers(0, 3)] = special[rng.integers(0, 3)]
        else: D[i, rng.integers(0, 3)] = special[rng.integers(2, 5)]
    g, c2 = r.trace_closest(O, D), o.trace_closest(O, D)
    for k in ("inst", "prim", "t", "u", "v"):
        assert_bit_equal(g[k], c2[k], f"closest.{k} at scale {scale}")
    # (hits exist from 1e-3 to 1e13; their number changes at 1e9 and again at 1e12, where |n0|^2 overflows and the planes n1, n2 collapse to 0:
    # below 1e-3 everything lies nearer than EPSILON, at 1e17 every dot product is infinite)
    if 1e-3 <= scale <= 1e13:
        assert np.isfinite(c2["t"]).sum() > m // 16
    tm = (rng.uniform(0, 400, m) * scale).astype(np.float32)
    assert np.array_equal(r.trace_any(O, D, tm), o.trace_any(O, D, tm))
    if scale == 1.0:
        # and a whole (tiny) frame through the integrator with these triangles in the scene: NaN planes reach the shading normals
        sc2 = SceneDesc.new(scenes.cornell_models()[:4] + [Model.new(pos * np.float32(0.8), nr.astype(np.float32), Lambertian.new((0.6, 0.6, 0.6)), None, "soup")],
                            scenes.reference_camera(1.5))
        want = oracle_mod.Oracle(sc2).render_samples(48, 32, 3, max_bounces=6)
        assert_bit_equal(api.Renderer(sc2, 48, 32, max_bounces=6).render_samples(0, 3), want, "frame with degenerate triangles")


@pytest.mark.parametrize("spp,batch", [(1, 0), (2, 0), (3, 0), (5, 0), (15, 0), (16, 0), (17, 0), (31, 0), (33, 0), (70, 0), (37, 8), (40, 16), (50, 17)])
def test_sample_counts_around_the_path_id_block(api, oracle_mod, spp, batch):
    """Path ids are dealt in blocks of 16 samples per pixel (RenderParams::blk_log) and a batch's last block may be short: sample counts below, at and
    around the block size, and requests cut into batches whose size is not a multiple of it — per-sample radiance, accumulated frame, first-hit
    position (the LAST sample's), id history (the last TWO samples') and tallies against the oracle."""
    from path_tracer_amd import scenes
    W, H = 56, 40
    sc = scenes.cornell_mixed(W, H)
    o = oracle_mod.Oracle(sc)
    r = api.Renderer(sc, W, H, max_bounces=5, batch_spp=batch)
    assert_bit_equal(r.render_samples(2, spp), o.render_samples(W, H, spp, first_sample=2, max_bounces=5), f"per-sample radiance, {spp} spp")
    r.reset_stats(); r.reset_accumulation()
    acc, pos, idb = r.render(2, spp)
    oacc, opos, oid, octr = o.render(W, H, spp, first_sample=2, max_bounces=5)
    assert_bit_equal(acc, oacc, "frame"); assert_bit_equal(pos, opos, "position of the last sample"); assert np.array_equal(idb, oid)
    st = r.stats()
    assert (st.rays_closest, st.rays_any, st.rays_light_closest) == (int(octr[0]), int(octr[1]), int(octr[2]))
