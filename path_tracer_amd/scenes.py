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

from .scene_desc import Camera, Dielectric, Emissive, GGX, Lambertian, Model, SceneDesc, Specular


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
