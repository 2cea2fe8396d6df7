"""Host-side mirror of the reference's Camera / Scene / integrate surface over the libptmi C-ABI (include/pt_api.h).

    Scene::new(models)                 src/scene.rs:21        -> Renderer(scene_desc, ...)
    Camera::new(...)                   src/camera.rs:17       -> Renderer.set_camera(CameraDesc)
    per-frame pixel loop + accumulate  src/main.rs:181-207    -> Renderer.render(first_sample, n_samples)
    TLAS::intersect / any_intersect    src/tlas.rs:66,111     -> Renderer.trace_closest / trace_any

All compute happens in libptmi.so (HIP, gfx950).  If the library is missing or no GPU is present the calls raise:
there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build
from .scene_desc import CameraDesc, SceneDesc

PT_OK = 0
FLAG_TIMING = 1
FLAG_NO_LDS_SCENE = 2
FLAG_TIMING_ALL = 4
FLAG_NO_PRIMARY_CULL = 8
DEFAULT_SEED = 0x5EED5EED

# every symbol include/pt_api.h declares
EXPORTS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_set_config", "pt_add_material", "pt_add_model", "pt_add_model_obj", "pt_model_vertices", "pt_build", "pt_set_camera",
    "pt_camera_matrices", "pt_set_environment", "pt_create_ray", "pt_render", "pt_render_device", "pt_reset_accumulation", "pt_accum_device_ptr",
    "pt_read_accumulation", "pt_read_frame", "pt_write_accumulation", "pt_render_samples", "pt_active_pixels", "pt_local_rows", "pt_set_stream", "pt_synchronize", "pt_camera_input", "pt_camera_angles", "pt_frame", "pt_inv_projection", "pt_present", "pt_post_velocity", "pt_post_reproject", "pt_post_tonemap", "pt_post_rgb8", "pt_present_rgb8", "pt_write_image", "pt_trace_closest", "pt_trace_any",
    "pt_ss_sobol", "pt_math_batch", "pt_material_eval", "pt_blas_count", "pt_blas_dump", "pt_tlas_dump", "pt_tlas_instances", "pt_light_cdf",
    "pt_triangle_dump", "pt_get_stats", "pt_reset_stats", "pt_last_batch_counters", "pt_last_batch_shade_pids", "pt_last_batch_step_stats",
    "pt_multi_create", "pt_multi_destroy", "pt_multi_last_error", "pt_multi_ctx", "pt_multi_render", "pt_multi_framebuffer_device_ptr",
    "pt_multi_reset_accumulation", "pt_multi_get_stats", "pt_multi_used_rccl", "pt_multi_write_image",
]


EV_MOUSE_MOTION, EV_KEY_W, EV_KEY_S, EV_KEY_A, EV_KEY_D = range(5)


class MaterialDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("colour", C.c_float * 3), ("roughness", C.c_float), ("ior", C.c_float), ("has_volume", C.c_int32),
                ("vol_absorption", C.c_float * 3), ("vol_k", C.c_float), ("vol_c", C.c_float), ("vol_g", C.c_float)]


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_bounces", C.c_uint32), ("n_sobol", C.c_uint32),
                ("enable_nee", C.c_uint32), ("seed", C.c_uint64), ("rank", C.c_uint32), ("world_size", C.c_uint32),
                ("strip_rows", C.c_uint32), ("batch_spp", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("stack_lds_levels", C.c_uint32), ("queue_slack", C.c_uint32), ("pipelines", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("rays_light_closest", C.c_uint64), ("paths", C.c_uint64),
                ("launches_trace_closest", C.c_uint64), ("ms_trace_closest", C.c_double), ("ms_trace_any", C.c_double),
                ("ms_trace_light", C.c_double), ("ms_shade", C.c_double), ("ms_generate", C.c_double), ("ms_accumulate", C.c_double),
                ("ms_total", C.c_double), ("scene_bytes", C.c_uint64), ("lds_scene", C.c_uint32), ("stack_entries", C.c_uint32),
                ("state_bytes", C.c_uint64), ("rays_light_closest_traced", C.c_uint64), ("rays_primary_culled", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    @property
    def rays(self):
        return self.rays_closest + self.rays_any + self.rays_light_closest


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libptmi error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libptmi.so (building it in-tree if the sources are newer).  Raises if that is impossible."""
    global _lib
    if _lib is None:
        path = os.environ.get("PTMI_LIB")  # tuning builds only; the default is the in-tree library
        if not path:
            path = _build.LIB_PATH
            if _build.is_stale():
                path = _build.build()
        # One HIP runtime per process: torch ships its own libamdhip64 / libhsa-runtime64 / librccl, and a process that has already
        # opened the GPU through /opt/rocm's copies cannot open it again through torch's ("No HIP GPUs are available").  This module
        # hands device pointers to torch (framebuffer_tensor, dist.gather_framebuffer), so torch's copies go in first and libptmi binds
        # to them by soname.  A C/C++/Rust host that never loads torch uses /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        vp, u32, f32p = C.c_void_p, C.c_uint32, C.c_void_p
        L.pt_create.restype = vp
        L.pt_create.argtypes = [C.POINTER(Config)]
        L.pt_destroy.argtypes = [vp]
        L.pt_last_error.restype = C.c_char_p
        L.pt_last_error.argtypes = [vp]
        L.pt_set_config.argtypes = [vp, C.POINTER(Config)]
        L.pt_add_material.argtypes = [vp, C.POINTER(MaterialDesc)]
        L.pt_add_model.argtypes = [vp, f32p, f32p, u32, C.c_int, f32p, u32]
        L.pt_add_model_obj.argtypes = [vp, C.c_char_p, C.c_int, f32p, u32]
        L.pt_model_vertices.argtypes = [vp, C.c_int, f32p, f32p, u32, C.POINTER(u32)]
        L.pt_build.argtypes = [vp]
        L.pt_set_camera.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float]
        L.pt_camera_matrices.argtypes = [vp, vp, vp]
        L.pt_set_environment.argtypes = [vp, u32, u32, vp]
        L.pt_create_ray.argtypes = [vp, C.c_float, C.c_float, vp, vp]
        L.pt_render.argtypes = [vp, u32, u32, vp, vp, vp]
        L.pt_render_device.argtypes = [vp, u32, u32]
        L.pt_active_pixels.argtypes = [vp, vp, vp]
        L.pt_reset_accumulation.argtypes = [vp]
        L.pt_accum_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.pt_read_accumulation.argtypes = [vp, vp]
        L.pt_write_accumulation.argtypes = [vp, vp, vp, vp]
        L.pt_read_frame.argtypes = [vp, vp, vp, vp]
        L.pt_render_samples.argtypes = [vp, u32, u32, vp]
        L.pt_local_rows.argtypes = [vp, C.POINTER(u32), vp, u32]
        L.pt_set_stream.argtypes = [vp, vp]
        L.pt_synchronize.argtypes = [vp]
        L.pt_frame.argtypes = [vp, u32, vp, vp, vp, vp]
        L.pt_camera_input.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float]
        L.pt_camera_angles.argtypes = [vp, vp]
        L.pt_inv_projection.argtypes = [vp, vp]
        L.pt_present.argtypes = [vp, vp]
        L.pt_multi_create.restype = vp
        L.pt_multi_create.argtypes = [C.POINTER(Config), vp, u32]
        L.pt_multi_destroy.argtypes = [vp]
        L.pt_multi_last_error.restype = C.c_char_p
        L.pt_multi_last_error.argtypes = [vp]
        L.pt_multi_ctx.restype = vp
        L.pt_multi_ctx.argtypes = [vp, u32]
        L.pt_multi_render.argtypes = [vp, u32, u32, vp]
        L.pt_multi_framebuffer_device_ptr.argtypes = [vp, C.POINTER(vp)]
        L.pt_multi_reset_accumulation.argtypes = [vp]
        L.pt_multi_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.pt_multi_used_rccl.argtypes = [vp]
        L.pt_multi_write_image.argtypes = [vp, C.c_char_p]
        L.pt_post_velocity.argtypes = [vp, u32, u32, vp, vp, vp]
        L.pt_post_reproject.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp]
        L.pt_post_tonemap.argtypes = [vp, u32, u32, vp, vp]
        L.pt_post_rgb8.argtypes = [vp, u32, u32, vp, vp]
        L.pt_present_rgb8.argtypes = [vp, vp]
        L.pt_write_image.argtypes = [vp, C.c_char_p]
        L.pt_trace_closest.argtypes = [vp, C.c_int, u32] + [vp] * 8
        L.pt_trace_any.argtypes = [vp, C.c_int, u32] + [vp] * 4
        L.pt_ss_sobol.argtypes = [vp, u32, u32, vp, vp, vp]
        L.pt_math_batch.argtypes = [vp, C.c_int, u32, vp, vp, vp, vp]
        L.pt_material_eval.argtypes = [vp, C.c_int, u32, vp, vp, vp, vp, vp, u32, vp]
        L.pt_blas_count.argtypes = [vp]
        L.pt_blas_dump.argtypes = [vp, C.c_int] + [vp] * 8 + [u32, u32]
        L.pt_tlas_dump.argtypes = [vp, C.c_int] + [vp] * 6 + [u32]
        L.pt_tlas_instances.argtypes = [vp, C.c_int, vp, vp, vp, u32]
        L.pt_light_cdf.argtypes = [vp] + [vp] * 6 + [u32]
        L.pt_triangle_dump.argtypes = [vp, C.c_int, u32, vp]
        L.pt_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.pt_reset_stats.argtypes = [vp]
        L.pt_last_batch_counters.argtypes = [vp, vp, u32, C.POINTER(u32)]
        L.pt_last_batch_shade_pids.argtypes = [vp, u32, u32, u32, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Renderer:
    """Scene + Camera + integrator behind one pt_ctx."""

    def __init__(self, scene: SceneDesc, width: int, height: int, max_bounces: int = 8, n_sobol: int = 512, enable_nee: bool = True,
                 seed: int = DEFAULT_SEED, rank: int = 0, world_size: int = 1, strip_rows: int = 4, batch_spp: int = 0, device: int = -1,
                 flags: int = 0, stack_lds_levels: int = 0, queue_slack: int = 0, pipelines: int = 0):
        self.L = lib()
        self.cfg = Config(width, height, max_bounces, n_sobol, int(enable_nee), seed, rank, world_size, strip_rows, batch_spp, device, flags,
                          stack_lds_levels, queue_slack, pipelines, 0)
        self.ctx = C.c_void_p(self.L.pt_create(C.byref(self.cfg)))
        if not self.ctx:
            raise PtError(-1, "pt_create failed (bad configuration)")
        self.desc = scene
        self._materials = []
        for mod in scene.models:
            self.add_model(mod)
        self.rebuild()
        if scene.camera is not None:
            self.set_camera(scene.camera)

    def _material_index(self, m) -> int:
        """pt_add_material once per distinct material, in first-use order (SceneDesc.materials() gives the same indices)"""
        if m not in self._materials:
            d = MaterialDesc()
            d.kind = m.kind
            d.colour[:] = m.colour
            d.roughness, d.ior = m.roughness, m.ior
            if m.volume is not None:
                d.has_volume = 1
                d.vol_absorption[:] = m.volume.absorption
                d.vol_k, d.vol_c, d.vol_g = m.volume.k, m.volume.c, m.volume.g
            self._chk(self.L.pt_add_material(self.ctx, C.byref(d)), allow_positive=True)
            self._materials.append(m)
        return self._materials.index(m)

    def add_model(self, mod) -> int:
        """Model::new + push onto the scene's model list; call rebuild() (Scene::new) before the next render"""
        mi = self._material_index(mod.material)
        if getattr(mod, "obj_path", None):
            return self._chk(self.L.pt_add_model_obj(self.ctx, mod.obj_path.encode(), mi, _p(mod.matrices), mod.matrices.shape[0]), allow_positive=True)
        return self._chk(self.L.pt_add_model(self.ctx, _p(mod.positions), _p(mod.normals), mod.positions.shape[0], mi, _p(mod.matrices),
                                             mod.matrices.shape[0]), allow_positive=True)

    def rebuild(self):
        self._chk(self.L.pt_build(self.ctx))

    # ---- plumbing
    def _chk(self, r, allow_positive=False):
        if r < 0 or (r != 0 and not allow_positive):
            raise PtError(r, self.L.pt_last_error(self.ctx).decode())
        return r

    @classmethod
    def attach(cls, ctx, cfg):
        """Wrap a context somebody else owns (a member of a MultiRenderer): same calls, close() leaves it alone."""
        self = cls.__new__(cls)
        self.L = lib()
        self.cfg = cfg
        self.ctx = C.c_void_p(ctx)
        self.desc = None
        self._materials = []
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "ctx", None):
            if not getattr(self, "_borrowed", False):
                self.L.pt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_config(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        self._chk(self.L.pt_set_config(self.ctx, C.byref(self.cfg)))

    def set_camera(self, cam: CameraDesc):
        self._chk(self.L.pt_set_camera(self.ctx, _f3(cam.origin), _f3(cam.target), cam.fov, cam.aspect_ratio))

    def camera_input(self, event, a=0.0, b=0.0, dt=0.0) -> bool:
        """Camera::input (camera.rs:56-92): event = EV_MOUSE_MOTION (a, b = delta) or EV_KEY_W/S/A/D; True if consumed"""
        return bool(self._chk(self.L.pt_camera_input(self.ctx, int(event), a, b, dt), allow_positive=True))

    def camera_angles(self):
        out = np.zeros(2, np.float32)
        self._chk(self.L.pt_camera_angles(self.ctx, _p(out)))
        return out

    def set_environment(self, rgb):
        """rgb: (h, w, 3) linear float32 equirect image, or None for the constant-ambient branch."""
        if rgb is None:
            self._chk(self.L.pt_set_environment(self.ctx, 0, 0, None))
        else:
            rgb = np.ascontiguousarray(rgb, np.float32)
            self._chk(self.L.pt_set_environment(self.ctx, rgb.shape[1], rgb.shape[0], _p(rgb)))

    def set_stream(self, hip_stream: Optional[int]):
        self._chk(self.L.pt_set_stream(self.ctx, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._chk(self.L.pt_synchronize(self.ctx))

    # ---- geometry of the local framebuffer
    def local_rows(self) -> np.ndarray:
        n = C.c_uint32()
        self._chk(self.L.pt_local_rows(self.ctx, C.byref(n), None, 0))
        rows = np.zeros(n.value, np.uint32)
        self._chk(self.L.pt_local_rows(self.ctx, C.byref(n), _p(rows), n.value))
        return rows

    @property
    def width(self):
        return self.cfg.width

    # ---- Camera
    def camera_matrices(self):
        m = np.zeros(12, np.float32)
        ip = np.zeros(16, np.float32)
        self._chk(self.L.pt_camera_matrices(self.ctx, _p(m), _p(ip)))
        return m.reshape(3, 4), ip.reshape(4, 4).T.copy()

    def create_ray(self, s, t):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        self._chk(self.L.pt_create_ray(self.ctx, s, t, _p(o), _p(d)))
        return o, d

    # ---- integrate over the frame
    def render(self, first_sample: int, n_samples: int, ident: Optional[np.ndarray] = None, want_position=True):
        rows = len(self.local_rows())
        acc = np.zeros((rows, self.cfg.width, 4), np.float32)
        pos = np.zeros((rows, self.cfg.width, 4), np.float32) if want_position else None
        idb = np.zeros((rows, self.cfg.width), np.uint32) if ident is None else ident
        self._chk(self.L.pt_render(self.ctx, first_sample, n_samples, _p(acc), _p(pos), _p(idb)))
        return acc, pos, idb

    def active_pixels(self):
        """(x0, width, local row0, rows) of the rectangle camera rays are generated for, and the world root box (min xyz, max xyz)"""
        rect = np.zeros(4, np.uint32); box = np.zeros(6, np.float32)
        self._chk(self.L.pt_active_pixels(self.ctx, _p(rect), _p(box)))
        return tuple(int(v) for v in rect), box

    def render_device(self, first_sample: int, n_samples: int):
        self._chk(self.L.pt_render_device(self.ctx, first_sample, n_samples))

    def render_samples(self, first_sample: int, n_samples: int):
        rows = len(self.local_rows())
        out = np.zeros((n_samples, rows, self.cfg.width, 4), np.float32)
        self._chk(self.L.pt_render_samples(self.ctx, first_sample, n_samples, _p(out)))
        return out

    def reset_accumulation(self):
        self._chk(self.L.pt_reset_accumulation(self.ctx))

    def read_accumulation(self):
        rows = len(self.local_rows())
        acc = np.zeros((rows, self.cfg.width, 4), np.float32)
        self._chk(self.L.pt_read_accumulation(self.ctx, _p(acc)))
        return acc

    def read_frame(self):
        """(accumulation, position, id history) as they lie on the device, e.g. after render_device()"""
        rows = len(self.local_rows())
        acc = np.zeros((rows, self.cfg.width, 4), np.float32); pos = np.zeros((rows, self.cfg.width, 4), np.float32); idb = np.zeros((rows, self.cfg.width), np.uint32)
        self._chk(self.L.pt_read_frame(self.ctx, _p(acc), _p(pos), _p(idb)))
        return acc, pos, idb

    def write_accumulation(self, data, position=None, ident=None):
        """Restore a frame's state (what render() returned) into this context: checkpoint / resume across contexts."""
        data = np.ascontiguousarray(data, np.float32)
        pos = None if position is None else np.ascontiguousarray(position, np.float32)
        idb = None if ident is None else np.ascontiguousarray(ident, np.uint32)
        px = self.cfg.width * self.cfg.height      # the library copies width * height texels from each pointer
        if data.size != px * 4 or (pos is not None and pos.size != px * 4) or (idb is not None and idb.size != px):
            raise PtError(-1, f"write_accumulation: arrays are not those of a {self.cfg.width}x{self.cfg.height} frame")
        self._chk(self.L.pt_write_accumulation(self.ctx, _p(data), None if pos is None else _p(pos), None if idb is None else _p(idb)))

    def accum_device_ptr(self):
        p = C.c_void_p()
        n = C.c_uint64()
        self._chk(self.L.pt_accum_device_ptr(self.ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- after the path: State::update / State::render
    def inv_projection(self):
        m = np.zeros(16, np.float32)
        self._chk(self.L.pt_inv_projection(self.ctx, _p(m)))
        return m

    def frame(self, frame_index, last_inv_projection=None, ident=None, download=True):
        """one event-loop iteration (main.rs:179-218): 1 spp pixel loop + State::update; returns data, position, id
        (download=False keeps everything on the device: the id history then lives in the library's own texture)"""
        m = None if last_inv_projection is None else np.ascontiguousarray(last_inv_projection, np.float32)
        if not download:
            self._chk(self.L.pt_frame(self.ctx, frame_index, _p(m), None, None, None))
            return None
        h, w = self.cfg.height, self.cfg.width
        data = np.zeros((h, w, 4), np.float32); pos = np.zeros((h, w, 4), np.float32)
        idb = np.zeros((h, w), np.uint32) if ident is None else ident
        self._chk(self.L.pt_frame(self.ctx, frame_index, _p(m), _p(data), _p(pos), _p(idb)))
        return data, pos, idb

    def present(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), np.float32)
        self._chk(self.L.pt_present(self.ctx, _p(out)))
        return out

    def present_rgb8(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 3), np.uint8)
        self._chk(self.L.pt_present_rgb8(self.ctx, _p(out)))
        return out

    def write_image(self, path):
        self._chk(self.L.pt_write_image(self.ctx, str(path).encode()))

    def post_rgb8(self, accum):
        accum = np.ascontiguousarray(accum, np.float32)
        h, w = accum.shape[:2]
        out = np.zeros((h, w, 3), np.uint8)
        self._chk(self.L.pt_post_rgb8(self.ctx, w, h, _p(accum), _p(out)))
        return out

    def post_velocity(self, position, last_inv_projection):
        position = np.ascontiguousarray(position, np.float32)
        h, w = position.shape[:2]
        v = np.zeros((h, w, 2), np.float32)
        self._chk(self.L.pt_post_velocity(self.ctx, w, h, _p(position), _p(np.ascontiguousarray(last_inv_projection, np.float32)), _p(v)))
        return v

    def post_reproject(self, inp, accum, velocity, ident):
        inp = np.ascontiguousarray(inp, np.float32); accum = np.ascontiguousarray(accum, np.float32)
        velocity = np.ascontiguousarray(velocity, np.float32); ident = np.ascontiguousarray(ident, np.uint32)
        h, w = inp.shape[:2]
        out = np.zeros((h, w, 4), np.float32)
        self._chk(self.L.pt_post_reproject(self.ctx, w, h, _p(inp), _p(accum), _p(velocity), _p(ident), _p(out)))
        return out

    def post_tonemap(self, accum):
        accum = np.ascontiguousarray(accum, np.float32)
        h, w = accum.shape[:2]
        out = np.zeros((h, w, 4), np.float32)
        self._chk(self.L.pt_post_tonemap(self.ctx, w, h, _p(accum), _p(out)))
        return out

    # ---- unit hooks
    def trace_closest(self, o, d, tmax=None, which=0):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        inst = np.zeros(n, np.uint32); prim = np.zeros(n, np.uint32)
        self._chk(self.L.pt_trace_closest(self.ctx, which, n, _p(o), _p(d), _p(tm), _p(t), _p(u), _p(v), _p(inst), _p(prim)))
        return dict(t=t, u=u, v=v, inst=inst, prim=prim)

    def trace_any(self, o, d, tmax, which=0):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        tm = np.ascontiguousarray(tmax, np.float32)
        hit = np.zeros(o.shape[0], np.uint8)
        self._chk(self.L.pt_trace_any(self.ctx, which, o.shape[0], _p(o), _p(d), _p(tm), _p(hit)))
        return hit

    def ss_sobol(self, n_points, index, seed):
        index = np.ascontiguousarray(index, np.uint32)
        seed = np.ascontiguousarray(seed, np.uint32)
        out = np.zeros((index.size, 2), np.float32)
        self._chk(self.L.pt_ss_sobol(self.ctx, n_points, index.size, _p(index), _p(seed), _p(out)))
        return out

    def math_batch(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        o0 = np.zeros_like(a)
        o1 = np.zeros_like(a)
        self._chk(self.L.pt_math_batch(self.ctx, fn, a.size, _p(a), _p(b), _p(o0), _p(o1)))
        return o0, o1

    def material_eval(self, material, incoming, normal, front, pixel, sample, draws_consumed=0):
        i = np.ascontiguousarray(incoming, np.float32); n = np.ascontiguousarray(normal, np.float32)
        f = np.ascontiguousarray(front, np.uint8); px = np.ascontiguousarray(pixel, np.uint32); sm = np.ascontiguousarray(sample, np.uint32)
        out = np.zeros((i.shape[0], 9), np.float32)
        self._chk(self.L.pt_material_eval(self.ctx, material, i.shape[0], _p(i), _p(n), _p(f), _p(px), _p(sm), draws_consumed, _p(out)))
        return out

    def model_vertices(self, model):
        n = C.c_uint32()
        self._chk(self.L.pt_model_vertices(self.ctx, model, None, None, 0, C.byref(n)))
        p = np.zeros((n.value, 3, 3), np.float32)
        nr = np.zeros((n.value, 3, 3), np.float32)
        self._chk(self.L.pt_model_vertices(self.ctx, model, _p(p), _p(nr), n.value, C.byref(n)))
        return p, nr

    # ---- host-builder introspection (CPU)
    def blas_count(self):
        return self._chk(self.L.pt_blas_count(self.ctx), allow_positive=True)

    def blas_dump(self, blas, cap=1 << 20):
        nn = C.c_uint32(); root = C.c_uint32(); nids = C.c_uint32()
        boxes = np.zeros((cap, 6), np.float32); kind = np.zeros(cap, np.uint32); a = np.zeros(cap, np.uint32); b = np.zeros(cap, np.uint32)
        ids = np.zeros(cap, np.uint32)
        self._chk(self.L.pt_blas_dump(self.ctx, blas, C.byref(nn), C.byref(root), _p(boxes), _p(kind), _p(a), _p(b), C.byref(nids), _p(ids), cap, cap))
        n = nn.value
        return dict(root=root.value, boxes=boxes[:n].copy(), kind=kind[:n].copy(), a=a[:n].copy(), b=b[:n].copy(), prim_ids=ids[:nids.value].copy())

    def tlas_dump(self, which=0, cap=1 << 16):
        nn = C.c_uint32(); root = C.c_uint32()
        boxes = np.zeros((cap, 6), np.float32); kind = np.zeros(cap, np.uint32); a = np.zeros(cap, np.uint32); b = np.zeros(cap, np.uint32)
        self._chk(self.L.pt_tlas_dump(self.ctx, which, C.byref(nn), C.byref(root), _p(boxes), _p(kind), _p(a), _p(b), cap))
        n = nn.value
        return dict(root=root.value, boxes=boxes[:n].copy(), kind=kind[:n].copy(), a=a[:n].copy(), b=b[:n].copy())

    def tlas_instances(self, which=0, cap=1 << 16):
        """matrix / inv_matrix of every TLAS leaf (leaf allocation order), [n, 3, 4] each"""
        n = C.c_uint32()
        m = np.zeros((cap, 3, 4), np.float32); inv = np.zeros((cap, 3, 4), np.float32)
        self._chk(self.L.pt_tlas_instances(self.ctx, which, C.byref(n), _p(m), _p(inv), cap))
        return dict(matrix=m[:n.value].copy(), inv_matrix=inv[:n.value].copy())

    def light_cdf(self, cap=1 << 20):
        n = C.c_uint32(); mx = C.c_float()
        pdf = np.zeros(cap, np.float32); cdf = np.zeros(cap, np.float32); bl = np.zeros(cap, np.uint32); pr = np.zeros(cap, np.uint32)
        self._chk(self.L.pt_light_cdf(self.ctx, C.byref(n), _p(pdf), _p(cdf), _p(bl), _p(pr), C.byref(mx), cap))
        k = n.value
        return dict(pdf=pdf[:k].copy(), cdf=cdf[:k].copy(), blas=bl[:k].copy(), prim=pr[:k].copy(), max=mx.value)

    def triangle(self, blas, prim):
        out = np.zeros(36, np.float32)
        self._chk(self.L.pt_triangle_dump(self.ctx, blas, prim, _p(out)))
        return out

    # ---- measurement
    def stats(self) -> Stats:
        s = Stats()
        self._chk(self.L.pt_get_stats(self.ctx, C.byref(s)))
        return s

    def last_batch_counters(self):
        rows = np.zeros((self.cfg.max_bounces + 2, 16), np.uint32)
        n = C.c_uint32()
        self._chk(self.L.pt_last_batch_counters(self.ctx, _p(rows), rows.shape[0], C.byref(n)))
        return rows[: n.value]

    def last_batch_shade_pids(self, qclass, first, count):
        out = np.zeros(count, np.uint32)
        self._chk(self.L.pt_last_batch_shade_pids(self.ctx, qclass, first, count, _p(out)))
        return out

    def last_batch_step_stats(self):
        rows = np.zeros((self.cfg.max_bounces + 2, 8), np.uint32)
        n = C.c_uint32()
        self._chk(self.L.pt_last_batch_step_stats(self.ctx, _p(rows), rows.shape[0], C.byref(n)))
        return rows[: n.value]

    def reset_stats(self):
        self._chk(self.L.pt_reset_stats(self.ctx))


class MultiRenderer:
    """One process, several devices (pt_multi): rows dealt to the devices in strips, one host thread per device while rendering, one
    RCCL gather of the strip framebuffers to devices[0].  `devices` may list a device more than once (contexts then share it and the
    gather is device-to-device copies): that is how the assembly is tested on a one-GPU box."""

    def __init__(self, scene: SceneDesc, width: int, height: int, devices, **kw):
        self.L = lib()
        devices = [int(d) for d in devices]
        kw.setdefault("strip_rows", 4)
        cfg = Config(width, height, kw.get("max_bounces", 8), kw.get("n_sobol", 512), int(kw.get("enable_nee", True)), kw.get("seed", DEFAULT_SEED), 0, 1,
                     kw["strip_rows"], kw.get("batch_spp", 0), -1, kw.get("flags", 0), 0, 0, kw.get("pipelines", 0), 0)
        arr = (C.c_int32 * len(devices))(*devices)
        self.m = C.c_void_p(self.L.pt_multi_create(C.byref(cfg), arr, len(devices)))
        if not self.m:
            raise PtError(-1, "pt_multi_create failed")
        self.width, self.height, self.n = width, height, len(devices)
        self.rank0 = Renderer.attach(self.L.pt_multi_ctx(self.m, 0), cfg)
        for mod in scene.models:
            self.rank0.add_model(mod)
        self.rank0.rebuild()
        if scene.camera is not None:
            self.rank0.set_camera(scene.camera)

    def _chk(self, r):
        if r != 0:
            raise PtError(r, self.L.pt_multi_last_error(self.m).decode())

    def render(self, first_sample: int, n_samples: int, download=True):
        out = np.zeros((self.height, self.width, 4), np.float32) if download else None
        self._chk(self.L.pt_multi_render(self.m, first_sample, n_samples, _p(out)))
        return out

    def reset_accumulation(self):
        self._chk(self.L.pt_multi_reset_accumulation(self.m))

    def write_image(self, path):
        self._chk(self.L.pt_multi_write_image(self.m, str(path).encode()))

    def used_rccl(self) -> bool:
        return bool(self.L.pt_multi_used_rccl(self.m))

    def stats(self) -> Stats:
        s = Stats()
        self._chk(self.L.pt_multi_get_stats(self.m, C.byref(s)))
        return s

    def close(self):
        if getattr(self, "m", None):
            self.rank0.close()
            self.L.pt_multi_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
