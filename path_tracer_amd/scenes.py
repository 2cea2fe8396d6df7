"""Synthetic scenes for BASELINE.json's configs (no assets exist in the reference repository:
`models/cornell/*.obj` of src/main.rs:101-114 are absent).  Geometry is generated from integer /
exactly-representable coordinates with float64 numpy arithmetic (IEEE, no libm transcendental), then
rounded once to float32, so the arrays are bit-identical on every host.

Cornell-class scene: 556-unit room centred on the reference camera's look-at point,
open towards +z; materials and emission from src/main.rs:82-92; camera from src/main.rs:122-127.
Model order follows src/main.rs:100-106: light, main (floor/ceiling/back), right (red), left (green), tall box, short box.
"""
from __future__ import annotations

import numpy as np

from .scene_desc import (IDENTITY_3x4, Camera, Dielectric, Emissive, GGX, Lambertian, Model, SceneDesc, Specular, Volume, affine_from_rotation_translation, is_rigid,
                         quat_from_rotation_y_pi, quat_unit)


def _quad(a, b, c, d):
    """Fan triangulation (a,b,c),(a,c,d) as the reference's OBJ loader does (blas.rs:97-119)."""
    a, b, c, d = (np.asarray(v, dtype=np.float64) for v in (a, b, c, d))
    return np.stack([np.stack([a, b, c]), np.stack([a, c, d])])


def _flat_normals(tris64, flip_towards=None):
    """Per-face unit normal replicated on the three vertices; optionally oriented towards a point."""
    e1 = tris64[:, 1] - tris64[:, 0]
    e2 = tris64[:, 2] - tris64[:, 0]
    n = np.cross(e1, e2)
    n = n / np.sqrt((n * n).sum(axis=1, keepdims=True))
    if flip_towards is not None:
        c = tris64.mean(axis=1)
        s = np.where(((np.asarray(flip_towards, dtype=np.float64) - c) * n).sum(axis=1, keepdims=True) < 0, -1.0, 1.0)
        n = n * s
    n = n + 0.0  # canonicalise -0.0
    return np.repeat(n[:, None, :], 3, axis=1)


def _box(top4, y_top, y_bottom):
    """Closed box: top, bottom and four sides from the four (x,z) top corners (12 triangles)."""
    t = [np.array([x, y_top, z], dtype=np.float64) for x, z in top4]
    b = [np.array([x, y_bottom, z], dtype=np.float64) for x, z in top4]
    faces = [_quad(t[0], t[1], t[2], t[3]), _quad(b[0], b[3], b[2], b[1])]
    for i in range(4):
        j = (i + 1) % 4
        faces.append(_quad(b[i], t[i], t[j], b[j]))
    tris = np.concatenate(faces)
    centre = np.mean(np.stack(t + b), axis=0)
    # outward normals: oriented away from the box centre
    n = _flat_normals(tris, flip_towards=centre) * -1.0 + 0.0
    return tris, n


def _model(tris64, normals64, material, name):
    return Model.new(tris64.astype(np.float32), normals64.astype(np.float32), material, None, name)


X0, X1 = -278.0, 278.0
Y0, Y1 = -228.0, 328.0
Z0, Z1 = -278.0, 278.0
ROOM_CENTRE = (0.0, 50.0, 0.0)


def cornell_models(tall_material=None, short_material=None):
    gray = Lambertian.new((0.73, 0.73, 0.73))       # main.rs:82
    green = Lambertian.new((0.12, 0.45, 0.15))      # main.rs:83
    red = Lambertian.new((0.65, 0.05, 0.05))        # main.rs:84
    light = Emissive.new((15.0, 15.0, 15.0))        # main.rs:92

    yl = 327.5
    light_t = _quad((-65.0, yl, -52.5), (65.0, yl, -52.5), (65.0, yl, 52.5), (-65.0, yl, 52.5))
    floor = _quad((X0, Y0, Z1), (X1, Y0, Z1), (X1, Y0, Z0), (X0, Y0, Z0))
    ceil = _quad((X0, Y1, Z0), (X1, Y1, Z0), (X1, Y1, Z1), (X0, Y1, Z1))
    back = _quad((X0, Y0, Z0), (X1, Y0, Z0), (X1, Y1, Z0), (X0, Y1, Z0))
    main_t = np.concatenate([floor, ceil, back])
    right_t = _quad((X1, Y0, Z0), (X1, Y0, Z1), (X1, Y1, Z1), (X1, Y1, Z0))
    left_t = _quad((X0, Y0, Z1), (X0, Y0, Z0), (X0, Y1, Z0), (X0, Y1, Z1))
    # classic Cornell block footprints, recentred (x-278, z -> 278-z) so that the reference camera frames them
    tall_t, tall_n = _box([(145.0, 31.0), (-13.0, -18.0), (36.0, -178.0), (194.0, -128.0)], 102.0, Y0)
    short_t, short_n = _box([(-148.0, 213.0), (-196.0, 53.0), (-38.0, 6.0), (12.0, 164.0)], -63.0, Y0)

    inward = ROOM_CENTRE
    return [
        _model(light_t, _flat_normals(light_t, inward), light, "cb_light"),
        _model(main_t, _flat_normals(main_t, inward), gray, "cb_main"),
        _model(right_t, _flat_normals(right_t, inward), red, "cb_right"),
        _model(left_t, _flat_normals(left_t, inward), green, "cb_left"),
        _model(tall_t, tall_n, tall_material or gray, "cb_box_tall"),
        _model(short_t, short_n, short_material or gray, "cb_box_short"),
    ]


def reference_camera(aspect_ratio):
    """src/main.rs:122-127: look_from (0,50,1000), look_at (0,50,0), fov 60 degrees."""
    return Camera.new((0.0, 50.0, 1000.0), (0.0, 50.0, 0.0), 60.0, aspect_ratio)


def cornell_box(width=256, height=256) -> SceneDesc:
    """BASELINE.json configs[0]/[1]: 6 quads + 2 diffuse boxes + area light = 36 triangles."""
    return SceneDesc.new(cornell_models(), reference_camera(width / height), "cornell")


def cornell_mixed(width=256, height=256) -> SceneDesc:
    """Material-variety Cornell: tall box = GGX metal (main.rs:86), short box = smooth dielectric (main.rs:89)."""
    models = cornell_models(GGX.new_metal((0.1, 0.1, 0.45), 0.4), Dielectric.new((0.95, 0.95, 0.95), 1.5, None))
    return SceneDesc.new(models, reference_camera(width / height), "cornell_mixed")


# ---------------------------------------------------------------------------------------------------------------------
# Procedural meshes for BASELINE.json configs[2..4] (no Stanford / Sponza assets exist offline).  float64 numpy only uses
# + - * / sqrt (correctly rounded everywhere) and integer hashing, so the float32 arrays are identical on every host.

def _icosphere(level):
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.sqrt((v * v).sum(1, keepdims=True))
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * (len(v) + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid = v[a] + v[b]
        mid /= np.sqrt((mid * mid).sum(1, keepdims=True))
        m = len(v) + inv.reshape(3, -1).T  # midpoint ids of edges (01, 12, 20) per face
        v = np.concatenate([v, mid])
        f = np.concatenate([np.stack([f[:, 0], m[:, 0], m[:, 2]], 1), np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
                            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1), np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)])
    return v, f


def _hash01(i):
    x = (np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(29)
    x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(32)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def sphere_mesh(level, centre, radius, displacement=0.0, smooth=True):
    """Icosphere (20 * 4^level triangles), optionally displaced radially by a per-vertex hash; returns (positions, normals)."""
    v, f = _icosphere(level)
    if displacement:
        v = v * (1.0 + displacement * (_hash01(np.arange(len(v))) - 0.5))[:, None]
    p = v * radius + np.asarray(centre, dtype=np.float64)
    tris = p[f]
    if smooth:
        fn = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
        vn = np.zeros_like(p)
        for k in range(3):
            np.add.at(vn, f[:, k], fn)
        vn /= np.sqrt((vn * vn).sum(1, keepdims=True))
        normals = vn[f]
    else:
        normals = _flat_normals(tris)
    return tris, normals + 0.0


def cornell_mesh(width=256, height=256, level=6, material=None) -> SceneDesc:
    """configs[2] class: Cornell room + light + one displaced icosphere (level 6 = 81 920 triangles) — deep BLAS, BVH in HBM/L2."""
    room = cornell_models()[:4]
    t, n = sphere_mesh(level, (0.0, -60.0, -20.0), 160.0, displacement=0.18)
    mesh = _model(t, n, material or Lambertian.new((0.73, 0.73, 0.73)), f"icosphere_l{level}")
    return SceneDesc.new(room + [mesh], reference_camera(width / height), f"cornell_mesh_l{level}")


def cornell_spheres(width=256, height=256, level=4) -> SceneDesc:
    """configs[4] class: diffuse + dielectric + GGX-metal spheres (materials of main.rs:82-91) in the Cornell room, with an
    instanced mirror sphere turned by the reference's own `Quat::from_rotation_y(PI)` (main.rs:97: a GENERAL matrix in binary32, its
    off-diagonal terms are +-8.742278e-8 because cos(PI_f32 / 2) is not 0) to exercise non-identity instance transforms."""
    room = cornell_models()[:4]
    s0 = sphere_mesh(level, (-150.0, -128.0, 60.0), 100.0)
    s1 = sphere_mesh(level, (40.0, -108.0, 130.0), 120.0)
    s2 = sphere_mesh(level, (140.0, -98.0, -120.0), 130.0)
    s3 = sphere_mesh(max(level - 1, 0), (-120.0, 180.0, -150.0), 60.0)
    rot_y_pi = affine_from_rotation_translation(quat_from_rotation_y_pi(), (0.0, 0.0, 0.0))[None]   # main.rs:97,112 (translation 0 here)
    models = room + [
        _model(*s0, Lambertian.new((0.05, 0.05, 0.25)), "sphere_diffuse"),                       # main.rs:85
        _model(*s1, Dielectric.new((0.95, 0.95, 0.95), 1.5, None), "sphere_glass"),              # main.rs:89
        _model(*s2, GGX.new_metal((0.1, 0.1, 0.45), 0.4), "sphere_ggx"),                         # main.rs:86
        Model.new(s3[0].astype(np.float32), s3[1].astype(np.float32), Specular.new((1.0, 1.0, 1.0)),   # main.rs:90
                  np.concatenate([np.eye(3, 4, dtype=np.float32)[None], rot_y_pi]), "sphere_mirror_x2"),
    ]
    return SceneDesc.new(models, reference_camera(width / height), "cornell_spheres")


def rigid_from_quat(a, b, c, d, translation=(0.0, 0.0, 0.0)) -> np.ndarray:
    """from_rotation_translation(unit quaternion (a, b, c, d) / |..|, translation); the caller picks quaternions that pass model.rs:40-44"""
    m = affine_from_rotation_translation(quat_unit(a, b, c, d), translation)
    assert is_rigid(m), (a, b, c, d)
    return m


def cornell_instanced(width=256, height=256, level=2) -> SceneDesc:
    """Instancing as the reference's own scene does it (main.rs:97-113: one model, `vec![IDENTITY, from_rotation_translation(
    from_rotation_y(PI), (0, 200, 0))]`) and beyond: every non-room model is placed by GENERAL rigid matrices built glam's way
    (unit quaternion -> Mat3A::from_quat in binary32, kept only if Model::new's scale == ONE assert passes), so every product in
    Ray::transform (ray.rs:22-28), Affine3A::inverse (tlas_bvh.rs:99), the normal transform (tlas.rs:105) and the corner-only
    AABB::transform (boundingbox.rs:51-57: TLAS leaf boxes that do NOT bound their rotated geometry) is inexact."""
    room = cornell_models()[:4]
    box_t, box_n = _box([(-40.0, 40.0), (-40.0, -40.0), (40.0, -40.0), (40.0, 40.0)], 60.0, -60.0)     # centred on the origin
    ball = sphere_mesh(level, (0.0, 0.0, 0.0), 70.0)
    slab_t, slab_n = _box([(-90.0, 25.0), (-90.0, -25.0), (90.0, -25.0), (90.0, 25.0)], 8.0, -8.0)
    models = room + [
        # the reference's own pair of instances, translation and all (main.rs:98,112), + two rotations about general axes
        Model.new(box_t.astype(np.float32), box_n.astype(np.float32), Lambertian.new((0.73, 0.73, 0.73)),
                  np.stack([np.array([[1, 0, 0, 150.0], [0, 1, 0, -168.0], [0, 0, 1, 40.0]], np.float32), reference_turn((-130.0, 32.0, -60.0)),
                            rigid_from_quat(1, -7, -3, 2, (20.0, -140.0, 150.0)), rigid_from_quat(3, -1, 2, 4, (-150.0, -150.0, 120.0))]), "box_x4"),
        # no identity instance at all: the BLAS is only ever seen through general matrices
        Model.new(ball[0].astype(np.float32), ball[1].astype(np.float32), Dielectric.new((0.95, 0.95, 0.95), 1.5, None),
                  np.stack([rigid_from_quat(1, -6, -5, 4, (-40.0, -150.0, -40.0)), rigid_from_quat(2, 3, -5, 7, (130.0, 60.0, -150.0))]), "glass_x2"),
        Model.new(slab_t.astype(np.float32), slab_n.astype(np.float32), GGX.new_metal((0.9, 0.6, 0.2), 0.3),
                  np.stack([rigid_from_quat(1, -7, 2, 3, (0.0, 120.0, -120.0)), reference_turn((0.0, 200.0, 0.0))]), "metal_slab_x2"),
        Model.new(slab_t.astype(np.float32), slab_n.astype(np.float32), Specular.new((0.9, 0.95, 1.0)),
                  rigid_from_quat(5, 1, 1, 7, (-160.0, 40.0, -200.0))[None], "mirror_slab"),
    ]
    return SceneDesc.new(models, reference_camera(width / height), "cornell_instanced")


def cornell_media(width=256, height=256, level=3) -> SceneDesc:
    """Participating media (material/volume.rs): the reference's own volume (main.rs:80) inside a rough-glass sphere
    (GGX::new_dielectric, main.rs:87), a smooth glass sphere with absorption only, and a purely scattering one nested in it."""
    room = cornell_models()[:4]
    vol_ref = Volume.new((0.4, 0.62, 0.7), 0.1, 1.0 / 200.0, 0.6)            # main.rs:80
    vol_abs = Volume.new((0.9, 0.2, 0.1), 0.02, 0.0, 0.0)                    # absorption only
    vol_sca = Volume.new((0.0, 0.0, 0.0), 0.0, 1.0 / 60.0, 0.0)              # isotropic scattering only
    s0 = sphere_mesh(level, (-120.0, -100.0, 40.0), 120.0)
    s1 = sphere_mesh(level, (120.0, -90.0, -40.0), 135.0)
    s2 = sphere_mesh(max(level - 1, 0), (120.0, -90.0, -40.0), 70.0)         # nested inside s1
    models = room + [
        _model(*s0, GGX.new_dielectric((0.95, 0.95, 0.95), 0.2, 1.5, vol_ref), "sphere_ggx_media"),
        _model(*s1, Dielectric.new((0.95, 0.95, 0.95), 1.5, vol_abs), "sphere_glass_absorbing"),
        _model(*s2, Dielectric.new((1.0, 1.0, 1.0), 1.3, vol_sca), "sphere_scattering_core"),
    ]
    return SceneDesc.new(models, reference_camera(width / height), "cornell_media")


def general_turn(rng, spread=60) -> np.ndarray:
    """A rigid instance matrix as the reference's host would build it — unit quaternion -> Mat3A::from_quat in binary32 (scene_desc) — that
    passes Model::new's own `scale == ONE` assert (model.rs:40-44; about a third of random quaternions do: the three column lengths must
    round to exactly 1.0f).  Every product of M3 * v is inexact: operation order in Ray::transform (ray.rs:22-28), Affine3A::inverse
    (tlas_bvh.rs:99), the normal transform (tlas.rs:105) and the corner-only AABB::transform (boundingbox.rs:51-57) all show."""
    while True:
        q = rng.integers(-50, 51, 4)
        if not q.any():
            continue
        m = affine_from_rotation_translation(quat_unit(*q), rng.integers(-spread, spread + 1, 3).astype(np.float32))
        if is_rigid(m):
            return m


def reference_turn(translation=(0.0, 200.0, 0.0)) -> np.ndarray:
    """The second dragon instance of the reference's own scene: from_rotation_translation(from_rotation_y(PI), (0, 200, 0))  main.rs:97-113."""
    return affine_from_rotation_translation(quat_from_rotation_y_pi(), translation)


def random_scene(seed, width=48, height=32, with_media=True, general=True) -> SceneDesc:
    """Seeded stress scene for parity fuzzing: several emissive models (multi-entry light CDF), triangle soups and spheres with
    every material kind, instanced rigid transforms — quarter turns + integer translations (exact) and, with `general`, glam-built
    rotations about arbitrary axes (general_turn; at least one instance of every scene) and the reference's own from_rotation_y(PI) —
    optional participating media.  Only numpy's IEEE + - * / sqrt and integer RNG output are used.  (general=False reproduces the
    scenes of the round 1-3 fuzz campaigns seed for seed.)"""
    rng = np.random.default_rng(seed)

    def soup(n, centre, spread):
        c = np.asarray(centre, np.float64) + rng.integers(-spread, spread + 1, (n, 1, 3)).astype(np.float64)
        return c + rng.integers(-40, 41, (n, 3, 3)).astype(np.float64) * 0.5

    def quarter_turn():
        axes = rng.permutation(3)
        signs = rng.choice([-1.0, 1.0], 3)
        m = np.zeros((3, 4), np.float32)
        for r in range(3):
            m[r, axes[r]] = signs[r]
        if np.linalg.det(m[:, :3].astype(np.float64)) < 0:
            m[0, :3] *= -1.0
        m[:, 3] = rng.integers(-60, 61, 3).astype(np.float32)
        return m

    vol = [Volume.new((0.4, 0.62, 0.7), 0.1, 1.0 / 200.0, 0.6), Volume.new((0.8, 0.3, 0.2), 0.03, 0.0, 0.0), Volume.new((0, 0, 0), 0.0, 1.0 / 90.0, -0.3)]
    palette = [
        Lambertian.new((0.73, 0.73, 0.73)), Lambertian.new((0.65, 0.05, 0.05)), Specular.new((0.9, 0.95, 1.0)),
        GGX.new_metal((0.9, 0.6, 0.2), float(rng.choice([0.05, 0.3, 0.8]))), GGX.new_dielectric((0.95, 0.95, 0.95), 0.25, 1.5, vol[0] if with_media else None),
        Dielectric.new((0.95, 0.95, 0.95), 1.5, vol[1] if with_media else None), Dielectric.new((1.0, 1.0, 1.0), 1.33, vol[2] if with_media else None),
        GGX.new_dielectric((1.0, 1.0, 1.0), 0.0, 1.4, None),
    ]
    models = cornell_models()[1:4]
    # two or three lights of different size / power
    for k in range(int(rng.integers(2, 4))):
        x, z, s = rng.integers(-180, 181), rng.integers(-180, 181), int(rng.integers(20, 70))
        y = 327.0 - k
        quad = _quad((x - s, y, z - s), (x + s, y, z - s), (x + s, y, z + s), (x - s, y, z + s))
        e = float(rng.integers(4, 30))
        models.append(_model(quad, _flat_normals(quad, ROOM_CENTRE), Emissive.new((e, e * 0.9, e * 0.7)), f"light{k}"))
    n_obj = int(rng.integers(3, 6))
    for k in range(n_obj):
        mat = palette[int(rng.integers(0, len(palette)))]
        centre = (float(rng.integers(-160, 161)), float(rng.integers(-150, 100)), float(rng.integers(-160, 161)))
        if rng.random() < 0.5:
            t, n = sphere_mesh(int(rng.integers(1, 4)), centre, float(rng.integers(40, 110)))
        else:
            t = soup(int(rng.integers(1, 40)), centre, 60)
            n = _flat_normals(t)
        mats = None
        if rng.random() < 0.5:
            mats = np.stack([np.eye(3, 4, dtype=np.float32)] + [quarter_turn() for _ in range(int(rng.integers(1, 3)))])
        if general and (k == 0 or rng.random() < 0.5):
            # general rotations: object 0 always has one (sometimes WITHOUT an identity instance beside it), the others half the time
            extra = [general_turn(rng) for _ in range(int(rng.integers(1, 3)))]
            if rng.random() < 0.25:
                extra.append(reference_turn(rng.integers(-60, 61, 3).astype(np.float32)))
            base = [] if (mats is None and rng.random() < 0.5) else [np.eye(3, 4, dtype=np.float32)[None] if mats is None else mats]
            mats = np.concatenate(base + [np.stack(extra)])
        models.append(Model.new(t.astype(np.float32), n.astype(np.float32), mat, mats, f"obj{k}"))
    return SceneDesc.new(models, reference_camera(width / height), f"random_{seed}" + ("" if general else "_exact"))


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] as SURVEY.md 8(d) fixes it: a procedurally generated ~250 k-triangle ATRIUM (columns / arches grid) —
# what makes a Sponza-class scene different from one big mesh is many objects, heavy instancing, occlusion and a deep TLAS.

def _circle(k):
    """4 * 2^k points on the unit circle, counter-clockwise from (1, 0), by repeated normalised midpoints (sqrt only: host-independent)."""
    p = np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0]])
    for _ in range(k):
        mid = p + np.roll(p, -1, axis=0)
        mid /= np.sqrt((mid * mid).sum(1, keepdims=True))
        q = np.empty((2 * len(p), 2))
        q[0::2], q[1::2] = p, mid
        p = q
    return p


def _grid_mesh(P):
    """triangles of a [rows, cols, 3] vertex grid that is closed around its columns (cols wraps), smooth vertex normals"""
    rows, cols, _ = P.shape
    idx = np.arange(rows * cols).reshape(rows, cols)
    a, b = idx[:-1], idx[1:]
    an, bn = np.roll(a, -1, axis=1), np.roll(b, -1, axis=1)
    f = np.concatenate([np.stack([a, an, bn], -1).reshape(-1, 3), np.stack([a, bn, b], -1).reshape(-1, 3)])
    v = P.reshape(-1, 3)
    tris = v[f]
    fn = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    vn /= np.sqrt((vn * vn).sum(1, keepdims=True))
    return tris, vn[f] + 0.0


def _column(variant, height=520.0, k=2, rings=48):
    """a fluted column standing on y = 0, axis = the y axis: (4 * 2^k) segments x `rings` rings, capped; four profile variants"""
    c = _circle(k)
    seg = len(c)
    # radius profile along the height (fractions of the height -> radius), piecewise linear: plinth, shaft with entasis, capital
    prof = [(0.0, 46.0), (0.04, 46.0), (0.05, 36.0), (0.5, 33.0 + 2.0 * variant), (0.9, 28.0), (0.93, 40.0), (1.0, 44.0 + 3.0 * (variant % 2))]
    t = np.arange(rings + 1) / float(rings)
    ys, rs = np.array([p[0] for p in prof]), np.array([p[1] for p in prof])
    j = np.clip(np.searchsorted(ys, t, side="right") - 1, 0, len(ys) - 2)
    r = rs[j] + (rs[j + 1] - rs[j]) * ((t - ys[j]) / (ys[j + 1] - ys[j]))
    flute = 1.0 - (0.05 + 0.01 * variant) * (np.arange(seg) % 2)              # every other meridian lies deeper: flutes
    shaft = ((t > 0.05) & (t < 0.9)).astype(np.float64)
    R = r[:, None] * (1.0 - (1.0 - flute[None, :]) * shaft[:, None])
    P = np.stack([R * c[None, :, 0], np.broadcast_to((t * height)[:, None], R.shape), R * c[None, :, 1]], -1)
    tris, nrm = _grid_mesh(P)
    top = np.array([0.0, height, 0.0])
    cap = np.stack([np.broadcast_to(top, (seg, 3)), P[-1], np.roll(P[-1], -1, axis=0)], 1)
    cap_n = np.broadcast_to(np.array([0.0, 1.0, 0.0]), cap.shape)
    return np.concatenate([tris, cap]), np.concatenate([nrm, cap_n]) + 0.0


def _arch(span=300.0, k=3, depth=50.0, thick=34.0):
    """a semicircular arch in the x-y plane, springing from (-span/2, 0) and (span/2, 0), extruded along z by `depth`: intrados,
    extrados and the two faces; flat normals"""
    c = _circle(k)
    half = c[: len(c) // 2 + 1]                                                    # angle 0 .. pi
    ri, ro = span / 2.0 - 30.0, span / 2.0 - 30.0 + thick
    z0, z1 = -depth / 2.0, depth / 2.0
    quads = []
    for (x0, y0), (x1, y1) in zip(half[:-1], half[1:]):
        ia, ib = (ri * x0, ri * y0), (ri * x1, ri * y1)
        oa, ob = (ro * x0, ro * y0), (ro * x1, ro * y1)
        quads.append(_quad((*ia, z1), (*ib, z1), (*ib, z0), (*ia, z0)))                # intrados (faces the centre)
        quads.append(_quad((*oa, z0), (*ob, z0), (*ob, z1), (*oa, z1)))                # extrados
        quads.append(_quad((*ia, z1), (*oa, z1), (*ob, z1), (*ib, z1)))                # front face
        quads.append(_quad((*ib, z0), (*ob, z0), (*oa, z0), (*ia, z0)))                # back face
    tris = np.concatenate(quads)
    return tris, _flat_normals(tris)


def _y_turns():
    """rotations about the y axis by general angles, built glam's way, that pass Model::new's assert (five inexact matrix entries)"""
    out = []
    for b in range(1, 40):
        for d in range(1, 40):
            if np.gcd(b, d) != 1 or b == d:
                continue
            m = affine_from_rotation_translation(quat_unit(0, b, 0, d), (0.0, 0.0, 0.0))
            if is_rigid(m):
                out.append(m)
    return out


def atrium(width=1920, height=1080, statue_level=2) -> SceneDesc:
    """configs[3] (SURVEY 8d "atrium (columns/arches grid)"): a hall of 6 x 12 fluted columns (4 BLAS variants, turned about their axes by
    quarter turns and by general glam-built rotations; three lie toppled under full 3-D rotations), two storeys of arches between them
    (2 BLASes, ~230 instances, half of them quarter-turned), and 280 statues that are each their own BLAS — 293 models in all, so the
    first-hit id `blas index as u8` (integrator.rs:184, main.rs:206) WRAPS for the statues nearest the camera — under skylight panels.
    ~250 k instanced triangles over ~96 k unique ones, a world TLAS of ~590 leaves (tlas_bvh.rs:85-138: agglomerative, quadratic)."""
    stone = [Lambertian.new((0.62, 0.58, 0.5)), Lambertian.new((0.55, 0.5, 0.45)), Lambertian.new((0.7, 0.66, 0.6)), Lambertian.new((0.5, 0.48, 0.46))]
    X0, X1, Z0, Z1, Y0, Y1 = -1050.0, 1050.0, -3450.0, 450.0, 0.0, 1240.0
    inward = (0.0, 500.0, -1500.0)
    floor = _quad((X0, Y0, Z1), (X1, Y0, Z1), (X1, Y0, Z0), (X0, Y0, Z0))
    ceil = _quad((X0, Y1, Z0), (X1, Y1, Z0), (X1, Y1, Z1), (X0, Y1, Z1))
    back = _quad((X0, Y0, Z0), (X1, Y0, Z0), (X1, Y1, Z0), (X0, Y1, Z0))
    right = _quad((X1, Y0, Z0), (X1, Y0, Z1), (X1, Y1, Z1), (X1, Y1, Z0))
    left = _quad((X0, Y0, Z1), (X0, Y0, Z0), (X0, Y1, Z0), (X0, Y1, Z1))
    models = []
    # skylight panels: one emissive model each (a lights TLAS with several leaves and a light CDF with several entries)
    for i, zc in enumerate((-300.0, -1200.0, -2100.0, -3000.0)):
        q = _quad((-160.0, Y1 - 1.0, zc - 220.0), (160.0, Y1 - 1.0, zc - 220.0), (160.0, Y1 - 1.0, zc + 220.0), (-160.0, Y1 - 1.0, zc + 220.0))
        e = 14.0 + 2.0 * i
        models.append(_model(q, _flat_normals(q, inward), Emissive.new((e, e * 0.95, e * 0.85)), f"skylight{i}"))
    shell = np.concatenate([floor, ceil, back])
    models.append(_model(shell, _flat_normals(shell, inward), Lambertian.new((0.6, 0.6, 0.6)), "hall_shell"))
    models.append(_model(right, _flat_normals(right, inward), Lambertian.new((0.6, 0.25, 0.2)), "hall_right"))
    models.append(_model(left, _flat_normals(left, inward), Lambertian.new((0.25, 0.35, 0.55)), "hall_left"))
    # columns: rows at x = -900, -600, -300 | aisle | 300, 600, 900; twelve bays along z
    xs = [-900.0, -600.0, -300.0, 300.0, 600.0, 900.0]
    zs = [300.0 - 300.0 * j for j in range(12)]
    turns = _y_turns()
    quarter = [np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0]], np.float32), np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0]], np.float32),
               np.array([[0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0]], np.float32)]
    col_mats = [[] for _ in range(4)]
    n = 0
    for i, x in enumerate(xs):
        for j, z in enumerate(zs):
            v = (i + 2 * j) % 4
            if n % 3 == 0:
                m = IDENTITY_3x4.copy()
            elif n % 3 == 1:
                m = quarter[(n // 3) % 3].copy()
            else:
                m = turns[(7 * n) % len(turns)].copy()
            m[:, 3] = (x, 0.0, z)
            col_mats[v].append(m)
            n += 1
    # three toppled columns in the aisle: rotations about general axes (nine inexact entries), lying near the floor
    col_mats[0].append(rigid_from_quat(2, -1, -2, 3, (-140.0, 60.0, -700.0)))
    col_mats[1].append(rigid_from_quat(3, 2, -1, 3, (120.0, 70.0, -1650.0)))
    col_mats[2].append(rigid_from_quat(4, -2, 2, 6, (-60.0, 80.0, -2500.0)))
    for v in range(4):
        t, nn = _column(v)
        models.append(Model.new(t.astype(np.float32), nn.astype(np.float32), stone[v], np.stack(col_mats[v]), f"column{v}"))
    # arches: along z inside every row, across x between neighbouring rows (not over the aisle), on two storeys
    arch_mats = [[], []]
    turn_y = np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0]], np.float32)      # the arch's span direction x -> z
    for storey, y in enumerate((520.0, 860.0)):
        for x in xs:
            for j in range(len(zs) - 1):
                m = turn_y.copy(); m[:, 3] = (x, y, (zs[j] + zs[j + 1]) / 2.0)
                arch_mats[storey].append(m)
        for i in (0, 1, 3, 4):
            for z in zs:
                m = IDENTITY_3x4.copy(); m[:, 3] = ((xs[i] + xs[i + 1]) / 2.0, y, z)
                arch_mats[storey].append(m)
    for storey in range(2):
        t, nn = _arch(k=4 - storey, thick=34.0 + 10.0 * storey)
        models.append(Model.new(t.astype(np.float32), nn.astype(np.float32), stone[2 + storey], np.stack(arch_mats[storey]), f"arch{storey}"))
    # statues: each its own BLAS (a displaced icosphere with its own seed), the LAST ones placed nearest the camera along the aisle
    palette = [Lambertian.new((0.75, 0.72, 0.68)), Lambertian.new((0.3, 0.5, 0.3)), Lambertian.new((0.55, 0.3, 0.25)), GGX.new_metal((0.9, 0.6, 0.2), 0.3),
               Lambertian.new((0.25, 0.3, 0.55)), Specular.new((0.9, 0.9, 0.9)), Lambertian.new((0.7, 0.65, 0.3)), Dielectric.new((0.95, 0.95, 0.95), 1.5, None)]
    spots = [(x, 564.0 + 44.0, z) for x in xs for z in zs]                                    # on the capitals: 72
    spots += [(sx * 1000.0, 70.0, 375.0 - 150.0 * j) for j in range(26) for sx in (-1.0, 1.0)]  # along the side walls: 52
    spots += [(sx * (150.0 + 75.0 * (j % 2)), 30.0, -3300.0 + 23.0 * j) for j in range(156) for sx in ((-1.0,) if j % 2 else (1.0,))]  # aisle, far -> near: 156
    v0, f0 = _icosphere(statue_level)
    for s, (x, y, z) in enumerate(spots):
        disp = 1.0 + 0.35 * (_hash01(np.arange(len(v0)) + 7919 * (s + 1)) - 0.5)
        radius = (42.0 + (s % 5) * 4.0) if s < 124 else (20.0 + (s % 5) * 2.0)             # the aisle's statues are small and stand clear of each other
        p = v0 * disp[:, None] * radius + np.array([x, y, z])
        tris = p[f0]
        models.append(_model(tris, _flat_normals(tris), palette[s % len(palette)], f"statue{s}"))
    cam = Camera.new((0.0, 180.0, 380.0), (0.0, 260.0, -1500.0), 60.0, width / height)
    return SceneDesc.new(models, cam, "atrium")
