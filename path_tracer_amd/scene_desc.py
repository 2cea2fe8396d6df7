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
