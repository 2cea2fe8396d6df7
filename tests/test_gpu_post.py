"""GPU (-m gpu): the post-kernels (State::update / State::render, src/state.rs:505-586,629-667) through the C-ABI against the
oracle's restatement.  Bit-exact; a NaN is a NaN whatever its payload (x86 and gfx950 differ in the default NaN's sign)."""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


assert_same = assert_bit_equal      # conftest.bits() already treats every NaN as the same value


@pytest.fixture(scope="module")
def api():
    from path_tracer_amd import api
    api.lib()
    return api


@pytest.fixture(scope="module")
def rnd(api, cornell64):
    return api.Renderer(cornell64, 64, 64, max_bounces=4)


def _moved(cam, dx):
    from path_tracer_amd.scene_desc import Camera
    return Camera.new((cam.origin[0] + dx, cam.origin[1], cam.origin[2] - 2 * dx), (cam.target[0] + 0.3 * dx, cam.target[1], cam.target[2]),
                      cam.fov, cam.aspect_ratio)


@pytest.mark.parametrize("w,h", [(64, 64), (37, 19), (1, 1), (256, 3)])
def test_tonemap_kernel(rnd, oracle_mod, w, h):
    rng = np.random.default_rng(w * 131 + h)
    acc = (rng.uniform(0, 3, (h, w, 4)) * np.exp(rng.uniform(-6, 4, (h, w, 1)))).astype(np.float32)
    acc[..., 3] = rng.integers(1, 300, (h, w))
    if w > 8:
        acc[0, 0] = [0, 0, 0, 1]; acc[0, 1] = [0, 0, 0, 0]; acc[0, 2] = [1e30, 5, 0.22, 1]; acc[0, 3] = [np.inf, -1, 0.62, 1]
    assert_same(rnd.post_tonemap(acc), oracle_mod.post_tonemap(acc), "tonemap")


@pytest.mark.parametrize("w,h", [(64, 64), (37, 19), (1, 1)])
def test_velocity_kernel(rnd, oracle_mod, w, h):
    rng = np.random.default_rng(w + 7 * h)
    pos = rng.uniform(-600, 600, (h, w, 4)).astype(np.float32)
    pos[..., 3] = rng.uniform(0, 2000, (h, w))
    if w > 8:
        pos[0, 0, :3] = np.inf; pos[0, 1, :3] = 0; pos[0, 2, :3] = [0, 50, 1000]                 # miss sentinel, origin, the eye itself
    M = rnd.inv_projection()
    assert_bit_equal(M, oracle_mod.Oracle(rnd.desc).inv_projection(), "inv_projection")
    assert_same(rnd.post_velocity(pos, M), oracle_mod.post_velocity(pos, M), "velocity")
    M2 = rng.normal(size=16).astype(np.float32)
    assert_same(rnd.post_velocity(pos, M2), oracle_mod.post_velocity(pos, M2), "velocity, arbitrary matrix")


@pytest.mark.parametrize("w,h,seed", [(64, 64, 0), (37, 19, 1), (1, 1, 2), (2, 5, 3), (128, 96, 4)])
def test_reproject_kernel_on_random_images(rnd, oracle_mod, w, h, seed):
    """all branches: history kept (same id, on-screen), id mismatch, off-screen history, velocity read at the closest-depth texel"""
    rng = np.random.default_rng(seed)
    inp = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
    inp[..., 3] = rng.choice(np.array([1.0, 0.5, 2.0], np.float32), (h, w))                       # .a drives the closest-depth pick
    acc = (rng.uniform(0, 2, (h, w, 4)) * rng.integers(1, 50, (h, w, 1))).astype(np.float32)
    acc[..., 3] = rng.choice(np.array([0.25, 1.0, 7.0, 120.0], np.float32), (h, w))               # both sides of max(w, 1)
    vel = (rng.normal(size=(h, w, 2)) * rng.choice([0.0, 0.01, 0.2, 1.5], (h, w, 1))).astype(np.float32)
    new = rng.integers(0, 3, (h, w)).astype(np.uint32); old = rng.integers(0, 3, (h, w)).astype(np.uint32)
    ident = (old << 16) | new
    got = rnd.post_reproject(inp, acc, vel, ident); want = oracle_mod.post_reproject(inp, acc, vel, ident)
    assert_same(got, want, "reproject")
    if w * h > 100:
        assert 0.05 < (want[..., 3] == 1).mean() and (want[..., 3] != 1).any()                    # both branches were taken


def test_reproject_kernel_extreme_velocities(rnd, oracle_mod):
    h, w = 16, 16
    rng = np.random.default_rng(9)
    inp = rng.uniform(0, 1, (h, w, 4)).astype(np.float32); acc = rng.uniform(0, 5, (h, w, 4)).astype(np.float32)
    vel = np.zeros((h, w, 2), np.float32)
    vel[0, :4] = [[np.inf, 0], [-np.inf, 0], [np.nan, 0], [1e30, -1e30]]
    vel[1, :3] = [[-3e9, 0], [0, 3e9], [1e-30, -1e-30]]
    ident = np.full((h, w), (1 << 16) | 1, np.uint32)
    assert_same(rnd.post_reproject(inp, acc, vel, ident), oracle_mod.post_reproject(inp, acc, vel, ident), "extreme velocities")


def test_frame_sequence_static_then_moving_camera(api, oracle_mod, cornell64):
    """the reference's event loop (main.rs:179-218) for six frames: three with the camera at rest (accumulate.wgsl), then three
    with it moving (velocity.wgsl + compute.wgsl), then State::render's tonemap — every texture after every frame"""
    W = H = 64
    r = api.Renderer(cornell64, W, H, max_bounces=4)
    o = oracle_mod.Oracle(cornell64)
    cams = [cornell64.camera] * 3 + [_moved(cornell64.camera, d) for d in (8.0, 20.0, 20.0)]
    acc = np.zeros((H, W, 4), np.float32)
    id_g = np.zeros((H, W), np.uint32); id_o = np.zeros((H, W), np.uint32)
    last_g = r.inv_projection(); last_o = o.inv_projection()
    branches = []
    for k, cam in enumerate(cams):
        r.set_camera(cam); o.set_camera(cam)
        data_g, pos_g, id_g = r.frame(k, last_g, id_g)
        data_o, pos_o, id_o, _ = o.render(W, H, 1, first_sample=k, max_bounces=4, ident=id_o)
        assert_bit_equal(data_g, data_o, f"frame {k} data"); assert_bit_equal(pos_g, pos_o, f"frame {k} position")
        assert_bit_equal(id_g, id_o, f"frame {k} id")
        cur_o = o.inv_projection()
        if np.array_equal(cur_o, last_o):
            acc = oracle_mod.post_accumulate(data_o, acc); branches.append("accumulate")
        else:
            acc = oracle_mod.post_reproject(data_o, acc, oracle_mod.post_velocity(pos_o, last_o), id_o); branches.append("reproject")
        assert_same(r.read_accumulation(), acc, f"frame {k} accumulation ({branches[-1]})")
        last_g = r.inv_projection(); last_o = cur_o
        assert_bit_equal(last_g, last_o, "inv_projection")
    assert branches == ["accumulate"] * 3 + ["reproject"] * 2 + ["accumulate"]
    assert_same(r.present(), oracle_mod.post_tonemap(acc), "present")
    assert np.isfinite(r.present()).all()


def test_interactive_loop_driven_by_camera_input(api, oracle_mod, cornell64):
    """main.rs:141-218 end to end: events move the camera (Camera::input), each MainEventsCleared traces one sample and updates
    the accumulation; compared with the oracle after every frame"""
    W = H = 48
    from path_tracer_amd import scenes
    sc = scenes.cornell_box(W, H)
    r = api.Renderer(sc, W, H, max_bounces=3); o = oracle_mod.Oracle(sc)
    events = [None, None, (api.EV_KEY_W, 0, 0, 2e-4), (api.EV_MOUSE_MOTION, 4.0, -1.5, 1e-5), None, (api.EV_KEY_D, 0, 0, 1e-4), (api.EV_KEY_S, 0, 0, 3e-4),
              (api.EV_MOUSE_MOTION, -9.0, 2.0, 1e-5), (api.EV_KEY_A, 0, 0, 1e-4), None]
    acc = np.zeros((H, W, 4), np.float32)
    id_g = np.zeros((H, W), np.uint32); id_o = np.zeros((H, W), np.uint32)
    last = o.inv_projection()
    moved = 0
    for k, ev in enumerate(events):
        if ev is not None:
            assert r.camera_input(*ev) and o.camera_input(*ev)
        data_g, pos_g, id_g = r.frame(k, last, id_g)
        data_o, pos_o, id_o, _ = o.render(W, H, 1, first_sample=k, max_bounces=3, ident=id_o)
        assert_bit_equal(data_g, data_o, f"frame {k} data"); assert_bit_equal(pos_g, pos_o, f"frame {k} position"); assert_bit_equal(id_g, id_o, f"frame {k} id")
        cur = o.inv_projection()
        if np.array_equal(cur, last):
            acc = oracle_mod.post_accumulate(data_o, acc)
        else:
            acc = oracle_mod.post_reproject(data_o, acc, oracle_mod.post_velocity(pos_o, last), id_o); moved += 1
        assert_same(r.read_accumulation(), acc, f"frame {k} accumulation")
        last = cur
        assert_bit_equal(r.inv_projection(), cur, "inv_projection")
    assert moved == 6
    assert np.array_equal(r.present_rgb8(), oracle_mod.post_rgb8(acc))


def _read_png(path):
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert (depth, ctype) == (8, 2)
    px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert np.all(px[:, 0] == 0)
    return px[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("w,h", [(64, 64), (37, 19), (1, 1), (300, 2)])
def test_rgb8_kernel(rnd, oracle_mod, w, h):
    rng = np.random.default_rng(w * 17 + h)
    acc = (rng.uniform(0, 3, (h, w, 4)) * np.exp(rng.uniform(-8, 3, (h, w, 1)))).astype(np.float32)
    acc[..., 3] = rng.integers(1, 300, (h, w))
    if w > 8:
        acc[0, 0] = [0, -1, np.nan, 1]; acc[0, 1] = [0, 0, 0, 0]; acc[0, 2] = [np.inf, 0.532, 0.22, 1]
        acc[0, 3, :3] = np.float32(0.22) + np.float32(0.78) * np.float32(0.4); acc[0, 3, 3] = 1     # x == m + l0: the 0/0 shoulder weight
    got = rnd.post_rgb8(acc); want = oracle_mod.post_rgb8(acc)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_write_image_round_trips_through_a_png_decoder(api, oracle_mod, cornell64, tmp_path):
    W, H = 96, 64
    from path_tracer_amd import scenes
    sc = scenes.cornell_box(W, H)
    r = api.Renderer(sc, W, H, max_bounces=4)
    r.render(0, 8)
    acc = r.read_accumulation()
    want = oracle_mod.post_rgb8(acc)
    assert np.array_equal(r.present_rgb8(), want)
    path = tmp_path / "out.png"
    r.write_image(path)
    assert np.array_equal(_read_png(path), want)
    assert want.max() > 200 and want.min() < 30                                                  # the light and the shadows are in it
    with pytest.raises(api.PtError) as e:
        r.write_image(tmp_path / "no_such_dir" / "out.png")
    assert e.value.code == -6


def test_cpp_host_driver_renders_what_the_oracle_renders(api, oracle_mod, tmp_path):
    """examples/headless (C++, include/ptmi.hpp) runs main.rs's loop over the OBJ Cornell with a scripted camera; the PNG it writes must
    hold the bytes the oracle computes for the same frames and events"""
    import subprocess
    from conftest import ROOT
    from path_tracer_amd import build as B, scenes
    from path_tracer_amd.scene_desc import Model, SceneDesc
    import os
    W, H, FRAMES, BOUNCES = 96, 64, 8, 4
    exe = B.build_host_driver()
    out_png = tmp_path / "headless.png"
    run = subprocess.run([exe, "--width", str(W), "--height", str(H), "--frames", str(FRAMES), "--bounces", str(BOUNCES), "--move", "--out", str(out_png)],
                         capture_output=True, text=True, cwd=ROOT)
    assert run.returncode == 0, run.stderr
    assert '"frames": 8' in run.stdout
    # the same loop on the oracle
    src = scenes.cornell_models()
    sc = SceneDesc.new([Model.from_obj(os.path.join(ROOT, "models", "cornell", m.name + ".obj"), m.material) for m in src], scenes.reference_camera(W / H))
    o = oracle_mod.Oracle(sc)
    acc = np.zeros((H, W, 4), np.float32); ident = np.zeros((H, W), np.uint32)
    last = o.inv_projection()
    for k in range(FRAMES):
        if k >= FRAMES // 2:
            o.camera_input(api.EV_KEY_W, 0.0, 0.0, 2.0e-6)
            o.camera_input(api.EV_MOUSE_MOTION, 1.0, 0.25, 1.0e-6)
        data, pos, ident, _ = o.render(W, H, 1, first_sample=k, max_bounces=BOUNCES, ident=ident)
        cur = o.inv_projection()
        acc = oracle_mod.post_accumulate(data, acc) if np.array_equal(cur, last) else oracle_mod.post_reproject(data, acc, oracle_mod.post_velocity(pos, last), ident)
        last = cur
    assert np.array_equal(_read_png(out_png), oracle_mod.post_rgb8(acc))


@pytest.mark.parametrize("args,rccl", [(["--gpus", "1"], "true"), (["--devices", "0,0,0"], "false")])
def test_cpp_host_driver_on_several_devices(api, oracle_mod, tmp_path, args, rccl):
    """examples/headless --gpus N: the C++ host side over pt_multi (one process, N contexts, one gather).  One device goes through RCCL;
    three contexts sharing the device exercise the strip assembly.  The PNG must hold the oracle's bytes for the same samples."""
    import os
    import subprocess
    from conftest import ROOT
    from path_tracer_amd import build as B, scenes
    from path_tracer_amd.scene_desc import Model, SceneDesc
    W, H, SPP, BOUNCES = 96, 62, 6, 5
    exe = B.build_host_driver()
    out_png = tmp_path / "multi.png"
    run = subprocess.run([exe, "--width", str(W), "--height", str(H), "--spp", str(SPP), "--bounces", str(BOUNCES), "--out", str(out_png)] + args,
                         capture_output=True, text=True, cwd=ROOT)
    assert run.returncode == 0, run.stderr
    assert f'"rccl": {rccl}' in run.stdout and f'"spp": {SPP}' in run.stdout
    src = scenes.cornell_models()
    sc = SceneDesc.new([Model.from_obj(os.path.join(ROOT, "models", "cornell", m.name + ".obj"), m.material) for m in src], scenes.reference_camera(W / H))
    acc = oracle_mod.Oracle(sc).render(W, H, SPP, max_bounces=BOUNCES)[0]
    assert np.array_equal(_read_png(out_png), oracle_mod.post_rgb8(acc))


def test_cpp_host_driver_resumes_a_render_in_another_process(api, oracle_mod, tmp_path):
    """examples/headless --render: one process renders samples [0, 3) and saves the frame state, a second process loads it and renders
    [3, 5): its PNG holds the bytes of the oracle's samples [0, 5) — a long render stopped and continued (pt_read_frame / pt_write_accumulation)."""
    import os
    import subprocess
    from conftest import ROOT
    from path_tracer_amd import build as B, scenes
    from path_tracer_amd.scene_desc import Model, SceneDesc
    W, H, BOUNCES = 80, 50, 5
    exe = B.build_host_driver()
    state, out_png = tmp_path / "state.bin", tmp_path / "resumed.png"
    common = [exe, "--width", str(W), "--height", str(H), "--bounces", str(BOUNCES)]
    a = subprocess.run(common + ["--render", "0", "3", "--save-state", str(state)], capture_output=True, text=True, cwd=ROOT)
    assert a.returncode == 0, a.stderr
    assert os.path.getsize(state) == 8 + W * H * (16 + 16 + 4)
    b = subprocess.run(common + ["--render", "3", "2", "--load-state", str(state), "--out", str(out_png)], capture_output=True, text=True, cwd=ROOT)
    assert b.returncode == 0, b.stderr
    src = scenes.cornell_models()
    sc = SceneDesc.new([Model.from_obj(os.path.join(ROOT, "models", "cornell", m.name + ".obj"), m.material) for m in src], scenes.reference_camera(W / H))
    acc = oracle_mod.Oracle(sc).render(W, H, 5, max_bounces=BOUNCES)[0]
    assert np.array_equal(_read_png(out_png), oracle_mod.post_rgb8(acc))
    bad = subprocess.run(common + ["--width", "64", "--render", "3", "2", "--load-state", str(state)], capture_output=True, text=True, cwd=ROOT)
    assert bad.returncode == 1 and "cannot read" in bad.stderr      # a state of another frame size is refused


def test_frame_needs_the_whole_image_on_one_rank(api, cornell64):
    r = api.Renderer(cornell64, 64, 64, rank=0, world_size=2)
    with pytest.raises(api.PtError) as e:
        r.frame(0)
    assert e.value.code == -3
