"""CPU: libptmi's host logic (no compute calls).  The library must load, export every symbol of include/pt_api.h, build
the same BLAS/TLAS/light tables as the oracle's independently written builders, and fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal


@pytest.fixture(scope="module")
def api():
    from path_tracer_amd import api
    api.lib()
    return api


def test_library_exports_every_declared_symbol(api):
    hdr = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    declared = sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", hdr)))
    declared = [d for d in declared if d not in ("pt_status",)]
    assert sorted(api.EXPORTS) == declared, set(declared) ^ set(api.EXPORTS)
    L = C.CDLL(api._build.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def _cmp(a, b, what):
    assert a.keys() == b.keys()
    for k in a:
        assert_bit_equal(np.asarray(a[k]), np.asarray(b[k]), f"{what}.{k}")


@pytest.mark.parametrize("scene_name", ["cornell_box", "cornell_mixed", "random_soup"])
def test_host_builders_match_oracle(api, oracle_mod, scene_name):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Emissive, Lambertian, Model, SceneDesc
    if scene_name == "random_soup":
        rng = np.random.default_rng(5)
        models = []
        for k, n in enumerate((1, 2, 37, 700)):
            c = rng.uniform(-100, 100, (n, 1, 3))
            p = (c + rng.normal(0, 6.0, (n, 3, 3))).astype(np.float32)
            nr = rng.normal(size=(n, 3, 3))
            nr = (nr / np.linalg.norm(nr, axis=2, keepdims=True)).astype(np.float32)
            th = 0.3 * k
            rot = np.array([[1, 0, 0, 10.0 * k], [0, 0, -1, -3.0], [0, 1, 0, 2.5]], np.float32)  # exact 90-degree rotation + translation
            mats = np.stack([np.eye(3, 4, dtype=np.float32), rot]) if k % 2 else None
            models.append(Model.new(p, nr, Emissive.new((5, 4, 3)) if k == 2 else Lambertian.new((0.5, 0.5, 0.5)), mats))
        sc = SceneDesc.new(models, scenes.reference_camera(1.0))
    else:
        sc = getattr(scenes, scene_name)(64, 64)
    r = api.Renderer(sc, 64, 64)
    o = oracle_mod.Oracle(sc)
    assert r.blas_count() == o.blas_count() == len(sc.models)
    for b in range(r.blas_count()):
        _cmp(r.blas_dump(b), o.blas_dump(b), f"blas{b}")
        for prim in (0, sc.models[b].positions.shape[0] - 1):
            assert_bit_equal(r.triangle(b, prim), o.triangle(b, prim), "triangle precompute")
    for which in (0, 1):
        _cmp(r.tlas_dump(which), o.tlas_dump(which), f"tlas{which}")
    _cmp(r.light_cdf(), o.light_cdf(), "light sampler")
    m, ip = r.camera_matrices()
    mo, ipo, _ = o.camera_matrices()
    assert_bit_equal(m, mo, "camera matrix")
    assert_bit_equal(ip, ipo, "inverse projection")
    for s, t in ((0.5, 0.5), (0.0, 0.0), (1.0, 0.25), (0.123, 0.987)):
        a, b = r.create_ray(s, t), o.create_ray(s, t)
        assert_bit_equal(a[0], b[0], "ray origin")
        assert_bit_equal(a[1], b[1], "ray direction")


def _signed_zero_soup(n, seed):
    """triangles on a coarse lattice that holds +0 and -0: many equal sort keys, many zero bounds of either sign"""
    rng = np.random.default_rng(seed)
    lattice = np.array([-2.0, -1.0, -0.0, 0.0, 1.0, 3.0], np.float32)
    p = lattice[rng.integers(0, lattice.size, (n, 3, 3))]
    p[:, 1] += np.float32(0.5) * (p[:, 1] == p[:, 0])          # no zero-length first edge (a NaN plane is refused by neither side, but is not the point here)
    nr = np.zeros((n, 3, 3), np.float32)
    nr[..., 1] = 1.0
    return p, nr


@pytest.mark.parametrize("threads", ["1", "8"])
@pytest.mark.parametrize("case", ["signed_zero_soup", "displaced_sphere_82k"])
def test_sweep_builder_is_the_reference_tree_whatever_the_thread_count(api, oracle_mod, monkeypatch, case, threads):
    """libptmi reaches the SAH sweep tree of blas_bvh.rs:62-136 with prefix / suffix folds, a keyed sort and forked subtrees; the oracle
    evaluates the reference's loops as written.  Same arena, same leaf order, same boxes to the bit - including the sign of a zero bound."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Lambertian, Model, SceneDesc
    monkeypatch.setenv("PTMI_BUILD_THREADS", threads)
    if case == "signed_zero_soup":
        p, nr = _signed_zero_soup(9000, 11)
        sc = SceneDesc.new([Model.new(p, nr, Lambertian.new((0.5, 0.5, 0.5)), None)], scenes.reference_camera(1.0))
        which = 0
    else:
        sc = scenes.cornell_mesh(64, 64, level=6)
        which = max(range(len(sc.models)), key=lambda i: sc.models[i].positions.shape[0])
    r = api.Renderer(sc, 64, 64)
    o = oracle_mod.Oracle(sc)
    a, b = r.blas_dump(which), o.blas_dump(which)
    assert a["kind"].size > sc.models[which].positions.shape[0] // 8
    if case == "signed_zero_soup":
        boxes = a["boxes"].view(np.uint32)
        assert (boxes == 0x80000000).any() and (boxes == 0).any()   # both zeros made it into node bounds
    _cmp(a, b, case)


def test_non_rigid_instance_is_rejected(api, oracle_mod):
    """model.rs:40-44 asserts scale == 1: PT_ERR_NONRIGID instead of a panic."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import SceneDesc
    sc = scenes.cornell_box(32, 32)
    bad = np.array([[[2, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]], np.float32)
    sc.models[4].matrices = bad
    with pytest.raises(api.PtError) as e:
        api.Renderer(sc, 32, 32)
    assert e.value.code == -4
    with pytest.raises(ValueError):
        oracle_mod.Oracle(SceneDesc.new(sc.models, sc.camera))


def test_no_gpu_means_loud_failure_not_fallback(api, cornell64):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = api.Renderer(cornell64, 64, 64)
    with pytest.raises(api.PtError) as e:
        r.render(0, 1)
    assert e.value.code == -2 and "no CPU path" in str(e.value)
    with pytest.raises(api.PtError):
        r.trace_closest(np.zeros((1, 3), np.float32), np.array([[0, 0, -1]], np.float32))


def test_state_errors(api):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Lambertian, SceneDesc
    sc = scenes.cornell_box(16, 16)
    no_light = SceneDesc.new([m for m in sc.models if m.material.kind != 1], sc.camera)
    r = api.Renderer(no_light, 16, 16, enable_nee=True)
    with pytest.raises(api.PtError) as e:
        r.render(0, 1)
    assert e.value.code == -3            # NEE without an emissive model (reference: panics building the light TLAS)
    assert api.lib().pt_create(None) is None


def test_row_sharding_partitions_the_image(api, cornell64):
    from path_tracer_amd.dist import rows_of_rank
    seen = []
    for rank in range(3):
        r = api.Renderer(cornell64, 64, 50, rank=rank, world_size=3, strip_rows=4)
        rows = r.local_rows()
        assert np.array_equal(rows, rows_of_rank(50, rank, 3, 4))
        seen.append(rows)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(50))


OBJ_TEXT = """# synthetic OBJ exercising load_obj (blas.rs:44-131)
o thing
v -100 -228 0
v 100 -228 0
v 100 -28.5 0.25
v -100 -28.5 1e-1
v 0 120.125 -40
vt 0.5 0.5
vn 0 0 2
vn 0.0 3.0 4.0
usemtl whatever
g quad_with_normals
f 1/1/1 2/1/1 3/1/2 4/1/2
g relative_indices
f -5//-2 -4//-1 -1//1
g face_normal_fallback
f 4/1/0 3/1/0 5/1/0
s off
f 1//1 3//1 5//1 2//2 4//2
"""


def test_obj_loader_matches_oracle(api, oracle_mod, tmp_path):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Lambertian, Model, SceneDesc
    path = tmp_path / "thing.obj"
    path.write_text(OBJ_TEXT)
    room = scenes.cornell_models()[:4]
    rot = np.array([[[0, 0, 1, 30.0], [0, 1, 0, 0], [-1, 0, 0, -20.0]]], np.float32)
    sc = SceneDesc.new(room + [Model.from_obj(str(path), Lambertian.new((0.3, 0.6, 0.9)), rot)], scenes.reference_camera(1.0))
    r = api.Renderer(sc, 32, 32)
    o = oracle_mod.Oracle(sc)
    gp, gn = r.model_vertices(4)
    cp, cn = o.model_vertices(4)
    assert gp.shape[0] == 2 + 1 + 1 + 3        # quad fan, triangle, triangle, pentagon fan
    assert_bit_equal(gp, cp, "obj positions"); assert_bit_equal(gn, cn, "obj normals")
    assert np.allclose(gn[0, 0], [0, 0, 1]) and np.allclose(gn[1, 2], [0, 0.6, 0.8])          # vn normalised on load
    assert_bit_equal(gp[2], np.array([[-100, -228, 0], [100, -228, 0], [0, 120.125, -40]], np.float32), "negative indices")
    e1, e2 = gp[3, 1] - gp[3, 0], gp[3, 2] - gp[3, 0]
    assert np.allclose(gn[3, 0], np.cross(e1, e2)) and np.array_equal(gn[3, 0], gn[3, 2])     # un-normalised face normal
    _cmp(r.blas_dump(4), o.blas_dump(4), "obj blas")
    _cmp(r.tlas_dump(0), o.tlas_dump(0), "obj tlas")


@pytest.mark.parametrize("text,code", [("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", -7), ("v 0 0 zero\n", -7), ("v 0 0 0\nf 1//1 2//1 3//1\n", -7), ("# nothing\n", -7)])
def test_obj_loader_errors(api, tmp_path, text, code):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Lambertian, Model, SceneDesc
    path = tmp_path / "bad.obj"
    path.write_text(text)
    sc = SceneDesc.new(scenes.cornell_models()[:4] + [Model.from_obj(str(path), Lambertian.new((1, 1, 1)))], scenes.reference_camera(1.0))
    with pytest.raises(api.PtError) as e:
        api.Renderer(sc, 8, 8)
    assert e.value.code == code
    sc.models[-1].obj_path = str(tmp_path / "missing.obj")
    with pytest.raises(api.PtError) as e:
        api.Renderer(sc, 8, 8)
    assert e.value.code == -6


# ---------------------------------------------------------------- Camera::input (camera.rs:33-92), host only
def _cam_state(x):
    m, ip = x.camera_matrices()[:2]
    return np.concatenate([m.ravel(), ip.ravel(), x.inv_projection().ravel(), x.camera_angles().ravel()])


def test_camera_new_derives_the_euler_angles(api, oracle_mod, cornell64):
    """camera.rs:23: the reference camera looks down -z, so both angles are zero; an arbitrary camera round-trips through
    update_rotation(0, 0) (angles -> quaternion -> matrix) to the same orientation"""
    from path_tracer_amd.scene_desc import Camera
    r = api.Renderer(cornell64, 64, 64); o = oracle_mod.Oracle(cornell64)
    assert np.all(np.abs(r.camera_angles()) == 0)
    cam = Camera.new((300.0, 220.0, 700.0), (-40.0, 10.0, -90.0), 50.0, 1.5)
    r.set_camera(cam); o.set_camera(cam)
    assert_bit_equal(_cam_state(r), _cam_state(o), "state after Camera::new")
    before, _ = r.camera_matrices()
    assert r.camera_input(api.EV_MOUSE_MOTION, 0.0, 0.0, 0.016)
    after, _ = r.camera_matrices()
    assert np.abs(before - after).max() < 2e-4 * max(1.0, np.abs(before).max())
    rot = after.reshape(3, 4)[:, :3].astype(np.float64)
    assert np.abs(rot @ rot.T - np.eye(3)).max() < 1e-6


def test_camera_input_matches_oracle_over_an_event_sequence(api, oracle_mod, cornell64):
    r = api.Renderer(cornell64, 64, 64); o = oracle_mod.Oracle(cornell64)
    rng = np.random.default_rng(5)
    for k in range(200):
        ev = int(rng.integers(0, 5))
        a, b = (float(np.float32(rng.normal() * 6)), float(np.float32(rng.normal() * 6))) if ev == 0 else (0.0, 0.0)
        dt = float(np.float32(rng.uniform(1e-5, 4e-4)))
        assert r.camera_input(ev, a, b, dt) is True and o.camera_input(ev, a, b, dt) is True
        assert_bit_equal(_cam_state(r), _cam_state(o), f"state after event {k} ({ev})")
        for s, t in ((0.5, 0.5), (0.03, 0.91)):
            go, gd = r.create_ray(s, t); co, cd = o.create_ray(s, t)
            assert_bit_equal(go, co, "ray origin"); assert_bit_equal(gd, cd, "ray direction")
    assert r.camera_input(17) is False and o.camera_input(17) is False                          # any other event: not consumed


def test_camera_keys_move_along_the_view_axes(api, cornell64):
    """update_origin: translation += matrix * (dx, 0, -dz) * dt * 5e5 — W/S along the view direction, A/D along the camera's x"""
    r = api.Renderer(cornell64, 64, 64)
    eye0, fwd = r.create_ray(0.5, 0.5)
    dt = 1e-4
    r.camera_input(api.EV_KEY_W, dt=dt)
    eye1, fwd1 = r.create_ray(0.5, 0.5)
    assert np.allclose(eye1 - eye0, fwd * np.float32(dt) * np.float32(5e5), atol=1e-3) and np.allclose(fwd1, fwd, atol=1e-6)
    r.camera_input(api.EV_KEY_S, dt=dt)
    assert np.allclose(r.create_ray(0.5, 0.5)[0], eye0, atol=1e-3)
    r.camera_input(api.EV_KEY_D, dt=dt)
    eye2, _ = r.create_ray(0.5, 0.5)
    right = (eye2 - eye0) / np.linalg.norm(eye2 - eye0)
    assert abs(float(right @ fwd)) < 1e-5 and right[0] > 0.99                                   # +x for the reference camera
    r.camera_input(api.EV_KEY_A, dt=dt)
    assert np.allclose(r.create_ray(0.5, 0.5)[0], eye0, atol=1e-3)


def test_camera_mouse_turns_about_y_then_x(api, cornell64):
    """update_rotation: `pitch` (the Y angle of EulerRot::YXZ) -= delta.0 * dt * 1e4, `yaw` (the X angle) -= delta.1 * dt * 1e4"""
    r = api.Renderer(cornell64, 64, 64)
    dt = 1e-5
    r.camera_input(api.EV_MOUSE_MOTION, 3.0, 0.0, dt)                                           # 0.3 rad about Y, towards +x (right)
    assert np.allclose(r.camera_angles(), [-0.3, 0.0], atol=1e-6)
    _, d = r.create_ray(0.5, 0.5)
    assert np.allclose(d, [np.sin(0.3), 0.0, -np.cos(0.3)], atol=1e-5)
    r.camera_input(api.EV_MOUSE_MOTION, 0.0, 2.0, dt)                                           # then 0.2 rad about the camera's x: down
    assert np.allclose(r.camera_angles(), [-0.3, -0.2], atol=1e-6)
    _, d = r.create_ray(0.5, 0.5)
    assert np.allclose(d, [np.sin(0.3) * np.cos(0.2), -np.sin(0.2), -np.cos(0.3) * np.cos(0.2)], atol=1e-5)
    eye, _ = r.create_ray(0.5, 0.5)
    assert np.allclose(eye, [0.0, 50.0, 1000.0])                                                # rotation keeps the translation


def test_camera_input_before_set_camera_is_a_state_error(api):
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import SceneDesc
    sc = scenes.cornell_box(64, 64)
    r = api.Renderer(SceneDesc.new(sc.models, None), 64, 64)
    with pytest.raises(api.PtError) as e:
        r.camera_input(api.EV_KEY_W, dt=0.01)
    assert e.value.code == -3


# ---------------------------------------------------------------- C++ host side (include/ptmi.hpp, examples/headless.cpp)
def test_cpp_host_driver_builds_links_and_fails_loudly_without_a_gpu(api):
    import subprocess
    from path_tracer_amd import build as B
    exe = B.build_host_driver()
    out = subprocess.run([exe, "--help"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "usage:" in out.stdout
    bad = subprocess.run([exe, "--models", "no/such/dir", "--width", "32", "--height", "32", "--frames", "1"], capture_output=True, text=True, cwd=ROOT)
    assert bad.returncode == 1 and "libptmi error -6" in bad.stderr                        # PT_ERR_IO from load_obj, as an exception
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([exe, "--width", "32", "--height", "32", "--frames", "1"], capture_output=True, text=True, cwd=ROOT)
        assert run.returncode == 1 and "no HIP device" in run.stderr                       # no CPU path behind the C++ side either


def test_cornell_obj_assets_match_the_generated_scene(api, oracle_mod):
    """models/cornell/*.obj (tools/make_cornell_obj.py) are the triangle soups of scenes.cornell_models(), read back by both loaders"""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Model, SceneDesc
    src = scenes.cornell_models()
    sc = SceneDesc.new([Model.from_obj(os.path.join(ROOT, "models", "cornell", m.name + ".obj"), m.material) for m in src], scenes.reference_camera(1.0))
    r = api.Renderer(sc, 32, 32); o = oracle_mod.Oracle(sc)
    for i, m in enumerate(src):
        gp, gn = r.model_vertices(i); op, on = o.model_vertices(i)
        assert_bit_equal(gp, op, "positions"); assert_bit_equal(gn, on, "normals")
        assert_bit_equal(gp.reshape(-1, 3, 3), m.positions, m.name + " positions survive the text round trip")
        assert np.abs(gn.reshape(-1, 3, 3) - m.normals).max() < 1e-6                          # vn is re-normalised on load (blas.rs:74)


@pytest.mark.parametrize("eye,target,fov,aspect,size", [((0.0, 50.0, 1000.0), (0.0, 50.0, 0.0), 60.0, 16 / 9, (192, 108)), ((900.0, 700.0, 1100.0), (0.0, 0.0, 0.0), 35.0, 1.5, (150, 100)),
                                                        ((0.0, 50.0, 1000.0), (900.0, 50.0, 0.0), 40.0, 1.0, (96, 96)), ((-400.0, 900.0, 600.0), (50.0, 0.0, -50.0), 25.0, 2.0, (200, 100))])
def test_active_pixel_rectangle_is_conservative(api, eye, target, fov, aspect, size):
    """Camera rays are generated only inside a rectangle of pixels (pt_active_pixels: the world root box projected onto the image plane,
    host code, no GPU).  Whatever the jitter, no camera ray of a pixel OUTSIDE it may meet the root box: checked here with the library's own
    Camera::create_ray and the reference's slab formula (boundingbox.rs:97-113) in binary32 on the rim of the rectangle and on random
    outside pixels."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import Camera, SceneDesc
    W, H = size
    sc = SceneDesc.new(scenes.cornell_models(), Camera.new(eye, target, fov, aspect), "view")
    r = api.Renderer(sc, W, H)
    (x0, w, y0, rows), box = r.active_pixels()
    assert 0 < w < W or 0 < rows < H, "this view should leave part of the frame outside the rectangle"
    mn, mx = box[:3].astype(np.float32), box[3:].astype(np.float32)
    EPS, INF = np.float32(5e-4), np.float32(np.inf)

    def hits(o, d):                                           # AABB::intersect with t_max = infinity
        with np.errstate(all="ignore"):
            inv = (np.float32(1.0) / d).astype(np.float32)
            t0, t1 = ((mn - o) * inv).astype(np.float32), ((mx - o) * inv).astype(np.float32)
            sse_max = lambda a, b: np.where(a > b, a, b)      # _mm_max_ps(a, b): b when either is NaN
            sse_min = lambda a, b: np.where(a < b, a, b)
            small = sse_min(sse_max(t0, EPS), sse_max(t1, EPS))
            big = sse_max(sse_min(t0, INF), sse_min(t1, INF))
            return small.max() <= big.min()

    rng = np.random.default_rng(3)
    outside = [(x, y) for x in range(W) for y in range(H) if not (x0 <= x < x0 + w and y0 <= y < y0 + rows)]
    rim = [(x, y) for (x, y) in outside if x0 - 1 <= x <= x0 + w and y0 - 1 <= y <= y0 + rows]
    pick = rim + [outside[i] for i in rng.choice(len(outside), min(300, len(outside)), replace=False)]
    assert pick
    for (x, y) in pick:
        for ox, oy in ((-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5), (0.0, 0.0), tuple(rng.uniform(-0.5, 0.5, 2))):
            o, d = r.create_ray(np.float32((x + ox) / W), np.float32((y + oy) / H))
            assert not hits(np.asarray(o, np.float32), np.asarray(d, np.float32)), (x, y, ox, oy)
    # ... and the rectangle is not needlessly large: some ray of its own rim does meet the box
    k = 6  # three pixels of margin, one of rounding, and the corner of the projected hull need not touch the rectangle's edge midway
    inner = [(x, y) for x in (x0 + k, x0 + w - 1 - k) for y in range(y0 + k, y0 + rows - k)] + [(x, y) for y in (y0 + k, y0 + rows - 1 - k) for x in range(x0 + k, x0 + w - k)]
    assert any(hits(*[np.asarray(v, np.float32) for v in r.create_ray(np.float32((x + 0.5) / W), np.float32((y + 0.5) / H))]) for (x, y) in inner)


def test_python_host_loads_one_hip_runtime():
    # torch ships its own HIP/HSA/RCCL; loading /opt/rocm's first and torch's second leaves torch without a GPU.  api.lib() must
    # end with exactly one libamdhip64 and one libhsa-runtime64 mapped, whichever module a test file touches first.
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from path_tracer_amd import api\n"
        "api.lib()\n"
        "import torch\n"
        "libs = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime64' in l}\n"
        "print(sorted(libs))\n"
        "assert len(libs) == 2, libs\n" % ROOT
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_bench_configurations_name_real_scenes():
    """bench.py --config: every configuration names a scene constructor that exists and takes its keyword arguments (built here for the
    small ones; the meshes are generated and counted), and its own / default sample counts are consistent."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from path_tracer_amd import scenes
    assert set(bench.CONFIGS) == {"cornell", "mesh82k", "atrium", "mesh328k", "mixed", "spheres"}
    for name, cfg in bench.CONFIGS.items():
        fn, kw, w, h, spp_own, spp_default, depth, what, spp_cpu = cfg
        assert hasattr(scenes, fn), name
        assert 1 <= spp_default <= spp_own and depth in (8, 16) and w * h >= 1920 * 1080 and 1 <= spp_cpu <= spp_default
    assert bench.make_scene(bench.CONFIGS["cornell"]).n_triangles() == 36
    assert bench.make_scene(bench.CONFIGS["mesh82k"]).n_triangles() == 81932
    assert bench.make_scene(bench.CONFIGS["mixed"]).n_triangles() == 36
    atrium = bench.make_scene(bench.CONFIGS["atrium"])     # configs[3] as SURVEY 8(d) defines it
    assert 240_000 <= atrium.n_triangles() <= 260_000 and len(atrium.models) > 256 and sum(len(m.matrices) for m in atrium.models) > 500


def _world_triangles_f64(sc):
    """every instance's triangles through its matrix in binary64: (vertices [n, 3, 3], (instance, primitive) tags)"""
    tris, tag, inst = [], [], 0
    for m in sc.models:
        for M in m.matrices.astype(np.float64):
            P = m.positions.astype(np.float64) @ M[:, :3].T + M[:, 3]
            tris.append(P)
            tag += [(inst, j) for j in range(len(P))]
            inst += 1
    return np.concatenate(tris), np.array(tag)


def _closest_f64(ro, rd, T):
    """Moller-Trumbore in binary64 over all triangles: an implementation that shares nothing with the oracle's Havel-Herout planes,
    object-space rays or BVHs"""
    e1, e2 = T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
    best = np.full(len(ro), np.inf); which = np.full(len(ro), -1)
    for s in range(0, len(ro), 512):
        o, d = ro[s:s + 512, None, :], rd[s:s + 512, None, :]
        p = np.cross(d, e2[None]); det = (e1[None] * p).sum(-1)
        inv = 1.0 / np.where(det == 0, 1e-300, det)
        tv = o - T[None, :, 0]; u = (tv * p).sum(-1) * inv
        q = np.cross(tv, e1[None]); v = (d * q).sum(-1) * inv; t = (e2[None] * q).sum(-1) * inv
        t = np.where((u >= 0) & (v >= 0) & (u + v <= 1) & (t > 5e-4) & (np.abs(det) > 1e-12), t, np.inf)
        j = t.argmin(1)
        best[s:s + 512] = t[np.arange(len(j)), j]; which[s:s + 512] = j
    return best, which


def test_general_rigid_instances(api, oracle_mod):
    """Instance matrices built the way glam builds them (unit quaternion -> Mat3A::from_quat in binary32) that pass Model::new's own
    scale == ONE assert, the reference's own from_rotation_translation(from_rotation_y(PI), (0, 200, 0)) among them (main.rs:97-113):
    every product of M3 * v is inexact, so operation order shows.  The library's and the oracle's independently written
    Affine3A::inverse / AABB::transform agree with each other and with the committed fixture; the oracle's hits agree with a binary64
    brute force over world-space triangles that shares none of its machinery; and where they differ it is quirk B-7 (corner-only
    AABB::transform, boundingbox.rs:51-57: a rotated instance's TLAS box does not bound it), which both sides must reproduce."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import affine_from_rotation_translation, is_rigid, quat_from_rotation_y_pi
    ref = affine_from_rotation_translation(quat_from_rotation_y_pi(), (0.0, 200.0, 0.0))
    assert is_rigid(ref) and ref[0, 2] == np.float32(-8.742278e-08) and ref[2, 0] == np.float32(8.742278e-08) and ref[0, 0] == -1.0
    g = np.load(os.path.join(ROOT, "tests", "golden", "instanced.npz"))
    sc = scenes.cornell_instanced(48, 32)
    assert any(np.array_equal(m, ref) for mo in sc.models for m in mo.matrices)
    assert sum(int(np.count_nonzero(m[:, :3]) == 9) for mo in sc.models for m in mo.matrices) >= 6     # rotations about general axes
    r = api.Renderer(sc, 48, 32)
    o = oracle_mod.Oracle(sc)
    for which in (0, 1):
        a, b = r.tlas_instances(which), o.tlas_instances(which)
        for k in ("matrix", "inv_matrix"):
            assert_bit_equal(a[k], b[k], f"tlas{which}.{k} library vs oracle")
            assert_bit_equal(b[k], g[f"{k}{which}"], f"tlas{which}.{k} oracle vs fixture")
        _cmp(r.tlas_dump(which), o.tlas_dump(which), f"tlas{which}")
        assert_bit_equal(o.tlas_dump(which)["boxes"], g[f"tlas{which}_boxes"], "TLAS boxes vs fixture")
        # Affine3A::inverse really inverts (binary64 product within binary32 rounding of the identity)
        M = np.concatenate([b["matrix"].astype(np.float64), np.tile([0, 0, 0, 1.0], (len(b["matrix"]), 1, 1))], axis=1)
        Mi = np.concatenate([b["inv_matrix"].astype(np.float64), np.tile([0, 0, 0, 1.0], (len(b["matrix"]), 1, 1))], axis=1)
        assert np.abs(Mi @ M - np.eye(4)).max() < 1e-4
    # the oracle still reproduces the fixture's hits (regression pin) ...
    ro, rd = g["ray_o"], g["ray_d"]
    h = o.trace_closest(ro, rd)
    for k in ("t", "u", "v", "inst", "prim", "normal", "front"):
        assert_bit_equal(h[k], g["hit0_" + k], f"closest.{k}")
    assert np.array_equal(o.trace_any(ro, rd, g["any_tmax"]), g["any_hit"])
    # ... and they are the hits of the transformed geometry
    T, tag = _world_triangles_f64(sc)
    bt, bi = _closest_f64(ro.astype(np.float64), rd.astype(np.float64), T)
    hit = np.isfinite(h["t"])
    same = hit & (tag[bi, 0] == h["inst"]) & (tag[bi, 1] == h["prim"])
    general = np.isin(h["inst"], [6, 7, 8, 9, 10, 12])
    ht = np.where(hit, h["t"], np.inf).astype(np.float64)
    assert (same & general).sum() > 500 and np.abs(bt[same] - ht[same]).max() < 2e-3
    # world-space normal = matrix * (interpolated object normal)  tlas.rs:105: compare with R * n computed in binary64
    tri_n = np.concatenate([mo.normals.astype(np.float64) @ M[:, :3].astype(np.float64).T for mo in sc.models for M in mo.matrices])
    u, v = h["u"].astype(np.float64)[same], h["v"].astype(np.float64)[same]
    nn = tri_n[bi[same]]
    n64 = nn[:, 0] * (1 - u - v)[:, None] + nn[:, 1] * u[:, None] + nn[:, 2] * v[:, None]
    n64 /= np.linalg.norm(n64, axis=1, keepdims=True)
    n64 *= np.where((n64 * rd[same]).sum(1) < 0, 1.0, -1.0)[:, None]           # face_forward (primitive.rs:166-169)
    assert np.abs(n64 - h["normal"][same]).max() < 1e-4
    # where the oracle's hit is NOT the binary64 closest one it is farther (geometry missed, never invented), the missed triangle
    # belongs to an instance rotated about a general axis, and the ray misses that instance's corner-only TLAS box
    other = np.isfinite(bt) & ~same
    other[other] = ~(np.abs(bt[other] - ht[other]) <= 1e-2)
    assert other.sum() > 100 and (bt[other] < ht[other]).all()
    assert np.isin(tag[bi[other], 0], [6, 7, 8, 9, 10, 12]).all()
    td = o.tlas_dump(0)
    leaf_box = {int(a): td["boxes"][i].astype(np.float64) for i, (k, a) in enumerate(zip(td["kind"], td["a"])) if k == 1}
    o64, d64 = ro.astype(np.float64)[other], rd.astype(np.float64)[other]
    boxes = np.stack([leaf_box[int(i)] for i in tag[bi[other], 0]])
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (boxes[:, :3] - o64) / d64, (boxes[:, 3:] - o64) / d64
    enter, leave = np.fmax(np.fmin(t0, t1).max(1), 5e-4), np.fmax(t0, t1).min(1)
    assert (enter > np.minimum(leave, bt[other] + 1.0)).mean() > 0.97, "the missed triangles lie outside the boxes the reference gives their instances"


def test_write_accumulation_refuses_a_frame_of_another_size(api, cornell64):
    """pt_write_accumulation copies width * height texels from each pointer: the wrappers (api.py, include/ptmi.hpp) check the sizes"""
    r = api.Renderer(cornell64, 64, 64)
    with pytest.raises(api.PtError):
        r.write_accumulation(np.zeros((32, 32, 4), np.float32))
    with pytest.raises(api.PtError):
        r.write_accumulation(np.zeros((64, 64, 4), np.float32), np.zeros((64, 64, 4), np.float32), np.zeros((64, 32), np.uint32))


def test_rigid_assert_agrees_between_python_library_and_oracle(api, oracle_mod):
    """Model::new's `scale == ONE` assert (model.rs:40-44) in three implementations — scene_desc.is_rigid (numpy binary32), libptmi's pt_add_model
    (PT_ERR_NONRIGID) and the oracle's — on glam-built matrices of random unit quaternions, about a third of which pass: the scenes' instance matrices
    are filtered by the first and must be accepted by the other two, and a matrix the reference would refuse must be refused by all."""
    from path_tracer_amd import scenes
    from path_tracer_amd.scene_desc import SceneDesc, affine_from_rotation_translation, is_rigid, quat_unit
    rng = np.random.default_rng(17)
    base = scenes.cornell_box(32, 32)
    seen = {True: 0, False: 0}
    for _ in range(120):
        q = rng.integers(-60, 61, 4)
        if not q.any():
            continue
        m = affine_from_rotation_translation(quat_unit(*q), rng.integers(-50, 51, 3).astype(np.float32))
        want = is_rigid(m)
        seen[want] += 1
        sc = SceneDesc.new(list(base.models), base.camera)
        sc.models[4] = type(sc.models[4])(sc.models[4].positions, sc.models[4].normals, sc.models[4].material, m[None].copy(), "probe")
        try:
            r = api.Renderer(sc, 32, 32)
            got_lib = True
            r.close()
        except api.PtError as e:
            assert e.code == -4
            got_lib = False
        try:
            oracle_mod.Oracle(sc)
            got_or = True
        except ValueError:
            got_or = False
        assert got_lib == want and got_or == want, (q, want, got_lib, got_or)
    assert seen[True] >= 20 and seen[False] >= 20
