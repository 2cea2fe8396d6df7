"""Plain-data scene description mirroring the reference's construction API.

Names follow the reference (all citations relative to /root/reference):
  Lambertian::new(albedo)                       src/tlas/tlas_bvh/blas/primitive/material.rs:99
  Emissive::new(emitted)                        material.rs:126
  Specular::new(colour)                         material.rs:146
  GGX::new_metal(colour, roughness)             material.rs:290
  GGX::new_dielectric(colour, roughness, ior, Option<Volume>)   material.rs:305
  Dielectric::new(colour, ior, Option<Volume>)  material.rs:475
  Volume::new(absorption, k, c, g)              material/volume.rs:136
  Model::new(path, material, matrices)          primitive/model.rs:36   (geometry arrays instead of an OBJ path)
  Camera::new(origin, target, fov, aspect, _, _)  src/camera.rs:17
  Scene::new(models)                            src/scene.rs:21

This module holds data only (numpy arrays); it computes nothing on the hot path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# material kinds: values are the C-ABI's pt_material_kind (include/pt_api.h)
LAMBERTIAN, EMISSIVE, SPECULAR, GGX_METAL, GGX_DIELECTRIC, DIELECTRIC = range(6)


@dataclass(frozen=True)
class Volume:
    absorption: tuple
    k: float
    c: float
    g: float

    @staticmethod
    def new(absorption, k, c, g) -> "Volume":
        return Volume(tuple(float(x) for x in absorption), float(k), float(c), float(g))


@dataclass(frozen=True)
class Material:
    kind: int
    colour: tuple
    roughness: float = 0.0
    ior: float = 1.0
    volume: Optional[Volume] = None


def _c3(v):
    v = tuple(float(x) for x in v)
    assert len(v) == 3
    return v


class Lambertian:
    @staticmethod
    def new(albedo) -> Material:
        return Material(LAMBERTIAN, _c3(albedo))


class Emissive:
    @staticmethod
    def new(emitted) -> Material:
        return Material(EMISSIVE, _c3(emitted))


class Specular:
    @staticmethod
    def new(colour) -> Material:
        return Material(SPECULAR, _c3(colour))


class GGX:
    @staticmethod
    def new_metal(colour, roughness) -> Material:
        return Material(GGX_METAL, _c3(colour), float(roughness))

    @staticmethod
    def new_dielectric(colour, roughness, ior, volume: Optional[Volume] = None) -> Material:
        return Material(GGX_DIELECTRIC, _c3(colour), float(roughness), float(ior), volume)


class Dielectric:
    @staticmethod
    def new(colour, ior, volume: Optional[Volume] = None) -> Material:
        return Material(DIELECTRIC, _c3(colour), 0.0, float(ior), volume)


IDENTITY_3x4 = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float32)


# --------------------------------------------------------------------------------------------------------------------------------
# Instance matrices the way the reference's host code builds them (src/main.rs:97-113): glam 0.23 quaternion -> Mat3A, all in binary32.
# numpy float32 scalars round every + - * once (IEEE), so these are the bits glam's scalar formulas produce (recalled).

# (sin, cos)(PI_f32 / 2) as binary32: what `Quat::from_rotation_y(std::f32::consts::PI)` holds in (y, w).  cos(1.57079637...) is not 0 in
# binary32: the reference's own second dragon instance is therefore a GENERAL matrix with +-8.742278e-8 off the diagonal (main.rs:97).
SIN_HALF_PI_F32 = np.float32(1.0)
COS_HALF_PI_F32 = np.array([0xB33BBD2E], np.uint32).view(np.float32)[0]   # -4.37113883e-8


def quat_from_rotation_y_pi() -> np.ndarray:
    """glam::Quat::from_rotation_y(PI) = (0, sin(PI/2), 0, cos(PI/2)) in binary32 (main.rs:97)."""
    return np.array([0.0, SIN_HALF_PI_F32, 0.0, COS_HALF_PI_F32], np.float32)


def quat_unit(a, b, c, d) -> np.ndarray:
    """The unit quaternion (a, b, c, d) / |(a, b, c, d)|, divided in binary64 (sqrt and / are correctly rounded on every host) and
    rounded once to binary32 — test input data standing in for what `Quat::from_axis_angle(..)` / `.normalize()` would hand the
    reference."""
    q = np.array([a, b, c, d], np.float64)
    return (q / np.sqrt((q * q).sum())).astype(np.float32)


def mat3_from_quat(q) -> np.ndarray:
    """glam Mat3A::from_quat in binary32; returns the 3x3 matrix (row-major array, columns are glam's x/y/z_axis)."""
    f = np.float32
    x, y, z, w = (f(v) for v in q)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    one = f(1.0)
    cols = [[one - (yy + zz), xy + wz, xz - wy], [xy - wz, one - (xx + zz), yz + wx], [xz + wy, yz - wx, one - (xx + yy)]]
    return np.array(cols, np.float32).T.copy()


def affine_from_rotation_translation(q, translation) -> np.ndarray:
    """glam::Affine3A::from_rotation_translation(rotation, translation) as the C-ABI's row-major 3x4 (main.rs:112)."""
    m = np.zeros((3, 4), np.float32)
    m[:, :3] = mat3_from_quat(q)
    m[:, 3] = np.asarray(translation, np.float32)
    return m


def is_rigid(m34) -> bool:
    """Model::new's assert (model.rs:40-44): `to_scale_rotation_translation().0 == Vec3::ONE`, i.e. the three column lengths
    sqrt((x*x + y*y) + z*z) are exactly 1.0f and the determinant is positive — in binary32, in glam's operation order."""
    f = np.float32
    m = np.asarray(m34, np.float32)
    c = [m[:, k] for k in range(3)]

    def dot(a, b):
        return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]

    def cross(a, b):
        return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], np.float32)

    det = dot(c[2], cross(c[0], c[1]))
    return bool(det > 0 and all(np.sqrt(dot(v, v)) == f(1.0) for v in c))


@dataclass
class Model:
    """One BLAS: triangle soup + ONE material + rigid instance transforms (model.rs:26-52)."""

    positions: np.ndarray  # [n_tris, 3, 3] float32
    normals: np.ndarray    # [n_tris, 3, 3] float32 (vertex normals; the OBJ loader's face-normal fallback is the caller's job)
    material: Material
    matrices: np.ndarray = field(default_factory=lambda: IDENTITY_3x4[None].copy())  # [n_inst, 3, 4] row-major
    name: str = ""
    obj_path: Optional[str] = None  # Model::new(path, ...): geometry read by the library's load_obj (blas.rs:44-131)

    @staticmethod
    def new(positions, normals, material: Material, matrices: Optional[Sequence] = None, name: str = "") -> "Model":
        p = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3, 3)
        n = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3, 3)
        assert p.shape == n.shape and p.shape[0] > 0
        m = IDENTITY_3x4[None].copy() if matrices is None else np.ascontiguousarray(matrices, dtype=np.float32).reshape(-1, 3, 4)
        return Model(p, n, material, m, name)

    @staticmethod
    def from_obj(path: str, material: Material, matrices: Optional[Sequence] = None, name: str = "") -> "Model":
        """Model::new(file_path, material, matrices) src/.../model.rs:36: the OBJ file is parsed by whoever consumes the scene."""
        m = IDENTITY_3x4[None].copy() if matrices is None else np.ascontiguousarray(matrices, dtype=np.float32).reshape(-1, 3, 4)
        z = np.zeros((0, 3, 3), np.float32)
        return Model(z, z, material, m, name or path, obj_path=path)


@dataclass(frozen=True)
class CameraDesc:
    origin: tuple
    target: tuple
    fov: float          # vertical, degrees (camera.rs:20 `fov.to_radians()`)
    aspect_ratio: float


class Camera:
    @staticmethod
    def new(origin, target, fov, aspect_ratio, _aperture=0.0, _focus=0.0) -> CameraDesc:
        return CameraDesc(_c3(origin), _c3(target), float(fov), float(aspect_ratio))


@dataclass
class SceneDesc:
    models: List[Model]
    camera: Optional[CameraDesc] = None
    name: str = ""

    @staticmethod
    def new(models: Sequence[Model], camera: Optional[CameraDesc] = None, name: str = "") -> "SceneDesc":
        return SceneDesc(list(models), camera, name)

    def materials(self) -> List[Material]:
        """Distinct materials in first-use order (index = material id handed to the C-ABI)."""
        out: List[Material] = []
        for m in self.models:
            if m.material not in out:
                out.append(m.material)
        return out

    def n_triangles(self) -> int:
        return int(sum(m.positions.shape[0] * m.matrices.shape[0] for m in self.models))
