"""CPU: the oracle's restatement of the step after the path — State::update (accumulate.wgsl, velocity.wgsl, compute.wgsl) and
State::render's tonemap (shader.wgsl) — against float64 numpy restatements and against properties of the algorithm."""
import numpy as np
import pytest

from conftest import assert_bit_equal


def _frame(oracle_mod, scene, w, h, sample, ident=None):
    o = oracle_mod.Oracle(scene)
    data, pos, idb, _ = o.render(w, h, 1, first_sample=sample, max_bounces=4, ident=ident)
    return o, data, pos, idb


def test_accumulate_adds_colour_and_counts_frames(oracle_mod):
    rng = np.random.default_rng(0)
    inp = rng.uniform(0, 4, (9, 13, 4)).astype(np.float32); acc = rng.uniform(0, 40, (9, 13, 4)).astype(np.float32)
    out = oracle_mod.post_accumulate(inp, acc)
    assert_bit_equal(out[..., :3], acc[..., :3] + inp[..., :3], "rgb")
    assert_bit_equal(out[..., 3], acc[..., 3] + np.float32(1), "frame count")             # accumulate.wgsl adds 1, not input.w


def test_tonemap_against_float64_gt_curve(oracle_mod):
    """shader.wgsl:3-33 (Uchimura 'GT' curve) with P=1 a=1 m=0.22 l=0.4 c=1.33 b=0, evaluated in float64"""
    x = np.concatenate([np.linspace(1e-4, 0.22, 200), np.linspace(0.22, 0.6, 200), np.linspace(0.6, 20, 400)])
    acc = np.zeros((1, x.size, 4), np.float32); n = np.float32(7)
    acc[0, :, 0] = (x * 7).astype(np.float32); acc[0, :, 1] = acc[0, :, 0]; acc[0, :, 2] = 0; acc[0, :, 3] = n
    out = oracle_mod.post_tonemap(acc)
    xf = (acc[0, :, 0] / n).astype(np.float64)
    P, a, m, l, c, b = 1.0, 1.0, 0.22, 0.4, 1.33, 0.0
    l0 = (P - m) * l / a
    t = np.clip(xf / m, 0, 1); w0 = 1 - t * t * (3 - 2 * t); w2 = (xf >= m + l0).astype(np.float64); w1 = 1 - w0 - w2
    s0, s1 = m + l0, m + a * l0; c2 = a * P / (P - s1)
    ref = (m * (xf / m) ** c + b) * w0 + (m + a * (xf - m)) * w1 + (P - (P - s1) * np.exp(-c2 * (xf - s0) / P)) * w2
    assert np.abs(out[0, :, 0] - ref).max() < 2e-6
    assert np.all(np.diff(out[0, :, 0]) >= -1e-6) and out[0, :, 0].max() <= 1.0                 # monotone, below the white point
    assert np.all(out[0, :, 2] == 0) and np.all(out[0, :, 3] == 1)


def test_velocity_is_subpixel_for_a_static_camera(oracle_mod, cornell64):
    """reprojecting this frame's own first-hit points with this frame's matrix: the ray went through x + [-0.5, 0.5) (main.rs:194-197)
    while velocity.wgsl puts the pixel at x + 0.5, so the residual is within one pixel and half a pixel on average"""
    o, data, pos, idb = _frame(oracle_mod, cornell64, 64, 64, 0)
    v = oracle_mod.post_velocity(pos, o.inv_projection())
    px = v * np.float32(64)
    assert px.min() >= -1e-3 and px.max() <= 1.0 + 1e-3 and abs(px.mean() - 0.5) < 0.02


def test_velocity_matches_float64_projection(oracle_mod, cornell64):
    o, data, pos, idb = _frame(oracle_mod, cornell64, 64, 64, 0)
    M = o.inv_projection().astype(np.float64).reshape(4, 4).T                                   # column-major -> matrix
    v = oracle_mod.post_velocity(pos, o.inv_projection())
    P = np.concatenate([pos[..., :3].astype(np.float64), np.ones((64, 64, 1))], -1)
    clip = P @ M.T
    ndc = clip[..., :2] / np.maximum(clip[..., 3:4], 1.0)
    yy, xx = np.mgrid[0:64, 0:64]
    uv = np.stack([(xx + 0.5) / 64, (yy + 0.5) / 64], -1)
    ref = uv - (ndc * 0.5 + 0.5)
    ok = np.isfinite(ref).all(-1) & (np.abs(clip[..., 3]) < 1e6)
    assert np.abs(v[ok] - ref[ok]).max() < 2e-5


def test_reproject_static_scene_blends_history_with_the_new_frame(oracle_mod):
    """zero velocity, one model id everywhere, flat history: output = mix(history, input, 0.15) wherever the history lies
    inside the neighbourhood's variance box (compute.wgsl:170-210)"""
    h, w = 12, 16
    rng = np.random.default_rng(1)
    inp = np.zeros((h, w, 4), np.float32); inp[..., :3] = rng.uniform(0.2, 0.8, (h, w, 3)); inp[..., 3] = 1
    mean = inp[..., :3].mean((0, 1))
    acc = np.zeros((h, w, 4), np.float32); acc[..., :3] = mean * 5; acc[..., 3] = 5             # 5 accumulated frames of the mean
    ident = np.full((h, w), (3 << 16) | 3, np.uint32)
    out = oracle_mod.post_reproject(inp, acc, np.zeros((h, w, 2), np.float32), ident)
    assert np.all(out[..., 3] == 1)
    inner = out[2:-2, 2:-2, :3]
    lo = np.minimum(mean, inp[2:-2, 2:-2, :3]) - 0.35; hi = np.maximum(mean, inp[2:-2, 2:-2, :3]) + 0.35
    assert np.all(inner >= lo) and np.all(inner <= hi)
    # a pixel whose neighbourhood brackets the history keeps it: out - 0.15*input = 0.85*history
    resid = inner - np.float32(0.15) * inp[2:-2, 2:-2, :3]
    kept = np.abs(resid - 0.85 * mean).max(-1) < 1e-4
    assert kept.sum() > 0


def test_reproject_disocclusion_resets_to_the_filtered_input(oracle_mod):
    """a changed model id or an off-screen history position takes the 2x2 box of the input, alpha included (compute.wgsl:170-181)"""
    h, w = 8, 8
    rng = np.random.default_rng(2)
    inp = rng.uniform(0, 1, (h, w, 4)).astype(np.float32); inp[..., 3] = 1
    acc = rng.uniform(0, 9, (h, w, 4)).astype(np.float32)
    vel = np.zeros((h, w, 2), np.float32)
    ident = np.full((h, w), (1 << 16) | 2, np.uint32)                                            # every pixel changed model
    out = oracle_mod.post_reproject(inp, acc, vel, ident)
    # sampling at texel corners averages the 2x2 block ending at (x, y) and the one starting there -> a 3x3 tent / 16
    pad = np.pad(inp.astype(np.float64), ((1, 1), (1, 1), (0, 0)), mode="edge")
    k = np.array([1, 2, 1], np.float64) / 4
    tent = sum(k[i] * k[j] * pad[i:i + h, j:j + w] for i in range(3) for j in range(3))
    assert np.abs(out - tent).max() < 1e-6
    # off-screen history: same id, velocity pointing one frame-width away
    ident2 = np.full((h, w), (2 << 16) | 2, np.uint32); vel2 = np.full((h, w, 2), 2.0, np.float32)
    assert_bit_equal(oracle_mod.post_reproject(inp, acc, vel2, ident2), out, "oob history")


def test_inv_projection_inverts_the_ray_matrix(oracle_mod, cornell64):
    """main.rs:128: (cam.matrix * cam.inv_projection).inverse(); a point on a primary ray projects back to that ray's NDC"""
    o = oracle_mod.Oracle(cornell64)
    M = o.inv_projection().astype(np.float64).reshape(4, 4).T
    for s, t in [(0.5, 0.5), (0.1, 0.9), (0.77, 0.2)]:
        org, d = o.create_ray(s, t)
        p = org.astype(np.float64) + 300.0 * d.astype(np.float64)
        clip = M @ np.append(p, 1.0)
        ndc = clip[:2] / clip[3]
        assert np.abs(ndc * 0.5 + 0.5 - [s, t]).max() < 1e-4


def test_rgb8_conversion_against_float64(oracle_mod):
    """image_helper.rs:41-48: GT curve (tonemapping.rs, the CPU variant) -> ^(1/2.2) -> x255 -> `as u8` (truncating, saturating)"""
    x = np.concatenate([[0.0, 1e-6, 0.22, 0.532, 0.5320001, 1.0, 50.0, -1.0], np.linspace(0, 3, 500)])
    acc = np.zeros((1, x.size, 4), np.float32)
    acc[0, :, 0] = (x * 3).astype(np.float32); acc[0, :, 1] = np.float32(0.3); acc[0, :, 2] = np.nan; acc[0, :, 3] = 3
    out = oracle_mod.post_rgb8(acc)
    xf = (acc[0, :, 0] / np.float32(3)).astype(np.float64)
    P, a, m, l, c, b = 1.0, 1.0, 0.22, 0.4, 1.33, 0.0
    l0 = (P - m) * l / a
    t = np.clip(xf / m, 0, 1); w0 = 1 - t * t * (3 - 2 * t); w2 = (xf > m + l0).astype(np.float64); w1 = 1 - w0 - w2
    s0, s1 = m + l0, m + a * l0; c2 = a * P / (P - s1)
    with np.errstate(invalid="ignore"):
        curve = (m * np.abs(xf / m) ** c + b) * w0 + (m + a * (xf - m)) * w1 + (P - (P - s1) * np.exp(-c2 * (xf - s0) / P)) * w2
        curve = np.where(xf < 0, b, curve)
        ref = np.clip(curve, 0, None) ** (1 / 2.2) * 255
    got = out[0, :, 0].astype(np.float64)
    near_edge = np.abs(ref - np.round(ref)) < 1e-3                                           # truncation boundary: either side is fine
    quirk = acc[0, :, 0] / np.float32(3) == np.float32(0.22) + np.float32(0.78) * np.float32(0.4)   # x == m + l0: gt_lerp(x, e, e) = 0/0
    assert quirk[3] and quirk.sum() == 1 and out[0, 3, 0] == 0                               # NaN weight -> NaN -> `as u8` 0, as the reference
    assert np.all((got == np.floor(np.clip(ref, 0, 255))) | near_edge | quirk)
    assert out[0, 0, 0] == 0 and out[0, 7, 0] == 0 and out[0, 6, 0] >= 254                    # black, negative -> b = 0, far above white
    assert np.all(out[0, :, 2] == 0)                                                         # NaN `as u8` is 0
    assert len(set(out[0, :, 1].tolist())) == 1
