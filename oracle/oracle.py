"""ctypes wrapper of the CPU oracle (oracle/libpt_oracle.so).

ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/pto_math.h).
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")
N_COUNTERS = 8


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("pt_oracle.cpp", "pt_oracle.h", "pto_math.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libpt_oracle.so"], check=True)
    return _LIB_PATH


class VolumeDesc(C.Structure):
    _fields_ = [("absorption", C.c_float * 3), ("k", C.c_float), ("c", C.c_float), ("g", C.c_float), ("present", C.c_int)]


class RenderCfg(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("first_sample", C.c_uint32), ("n_samples", C.c_uint32),
        ("max_bounces", C.c_uint32), ("n_sobol", C.c_uint32), ("seed", C.c_uint64), ("enable_nee", C.c_uint32),
        ("threads", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.pto_create.restype = C.c_void_p
        L.pto_destroy.argtypes = [C.c_void_p]
        L.pto_add_material.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(VolumeDesc)]
        L.pto_add_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32]
        L.pto_add_model_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_uint32]
        L.pto_model_vertices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.pto_build.argtypes = [C.c_void_p]
        L.pto_set_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float]
        L.pto_set_environment.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pto_camera_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_inv_projection.argtypes = [C.c_void_p, C.c_void_p]
        L.pto_camera_move.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.pto_camera_rotate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.pto_camera_angles.argtypes = [C.c_void_p, C.c_void_p]
        L.pto_post_accumulate.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pto_post_velocity.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_post_reproject.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 5
        L.pto_post_tonemap.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pto_post_rgb8.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pto_create_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.pto_primary_ray.argtypes = [C.c_void_p, C.POINTER(RenderCfg), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pto_render.argtypes = [C.c_void_p, C.POINTER(RenderCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_render_samples.argtypes = [C.c_void_p, C.POINTER(RenderCfg), C.c_void_p]
        L.pto_integrate.argtypes = [C.c_void_p, C.POINTER(RenderCfg), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_trace_closest.argtypes = [C.c_void_p, C.c_int, C.c_uint32] + [C.c_void_p] * 10
        L.pto_trace_any.argtypes = [C.c_void_p, C.c_int, C.c_uint32] + [C.c_void_p] * 4
        L.pto_blas_count.argtypes = [C.c_void_p]
        L.pto_blas_dump.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_uint32, C.c_uint32]
        L.pto_tlas_dump.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_uint32]
        L.pto_tlas_instances.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.pto_light_cdf.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_uint32]
        L.pto_triangle_dump.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
        L.pto_ss_sobol_raw.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pto_ss_sobol.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pto_sobol_dim1.argtypes = [C.c_uint32]
        L.pto_sobol_dim1.restype = C.c_uint32
        L.pto_low_bias_hash.argtypes = [C.c_uint32]
        L.pto_low_bias_hash.restype = C.c_uint32
        L.pto_lk_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.pto_lk_hash.restype = C.c_uint32
        L.pto_wyrand.argtypes = [C.c_uint64, C.c_uint32]
        L.pto_wyrand.restype = C.c_uint64
        L.pto_stream_state0.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.pto_stream_state0.restype = C.c_uint64
        L.pto_math_batch.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_material_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


DEFAULT_SEED = 0x5EED5EED


class Oracle:
    """CPU restatement of Scene + Camera + integrate for one SceneDesc."""

    def __init__(self, scene_desc):
        L = lib()
        self.L = L
        self.ctx = C.c_void_p(L.pto_create())
        self.desc = scene_desc
        mats = scene_desc.materials()
        for m in mats:
            vol = VolumeDesc()
            if m.volume is not None:
                vol.absorption[:] = m.volume.absorption
                vol.k, vol.c, vol.g, vol.present = m.volume.k, m.volume.c, m.volume.g, 1
            r = L.pto_add_material(self.ctx, m.kind, _f3(m.colour), m.roughness, m.ior, C.byref(vol))
            assert r >= 0
        for mod in scene_desc.models:
            if getattr(mod, "obj_path", None):
                r = L.pto_add_model_obj(self.ctx, mod.obj_path.encode(), mats.index(mod.material), _p(mod.matrices), mod.matrices.shape[0])
            else:
                r = L.pto_add_model(self.ctx, _p(mod.positions), _p(mod.normals), mod.positions.shape[0], mats.index(mod.material),
                                    _p(mod.matrices), mod.matrices.shape[0])
            if r < 0:
                raise ValueError(f"pto_add_model failed: {r}")
        r = L.pto_build(self.ctx)
        assert r == 0, r
        if scene_desc.camera is not None:
            self.set_camera(scene_desc.camera)

    def __del__(self):
        try:
            self.L.pto_destroy(self.ctx)
        except Exception:
            pass

    def model_vertices(self, model):
        n = C.c_uint32()
        self.L.pto_model_vertices(self.ctx, model, None, None, 0, C.byref(n))
        p = np.zeros((n.value, 3, 3), np.float32); nr = np.zeros((n.value, 3, 3), np.float32)
        r = self.L.pto_model_vertices(self.ctx, model, _p(p), _p(nr), n.value, C.byref(n))
        assert r == 0
        return p, nr

    def set_camera(self, cam):
        self.L.pto_set_camera(self.ctx, _f3(cam.origin), _f3(cam.target), cam.fov, cam.aspect_ratio)

    def set_environment(self, rgb):
        if rgb is None:
            self.L.pto_set_environment(self.ctx, 0, 0, None)
        else:
            rgb = np.ascontiguousarray(rgb, np.float32)
            self.L.pto_set_environment(self.ctx, rgb.shape[1], rgb.shape[0], _p(rgb))

    def cfg(self, width, height, first_sample=0, n_samples=1, max_bounces=8, n_sobol=512, seed=DEFAULT_SEED, enable_nee=1,
            threads=0, rows=(0, 0)):
        return RenderCfg(width, height, first_sample, n_samples, max_bounces, n_sobol, seed, enable_nee, threads, rows[0], rows[1])

    def render(self, width, height, n_samples, accum=None, ident=None, **kw):
        cfg = self.cfg(width, height, n_samples=n_samples, **kw)
        acc = np.zeros((height, width, 4), np.float32) if accum is None else accum
        pos = np.zeros((height, width, 4), np.float32)
        idb = np.zeros((height, width), np.uint32) if ident is None else ident
        ctr = np.zeros(N_COUNTERS, np.uint64)
        r = self.L.pto_render(self.ctx, C.byref(cfg), _p(acc), _p(pos), _p(idb), _p(ctr))
        if r != 0:
            raise RuntimeError(f"pto_render: {r}")
        return acc, pos, idb, ctr

    def render_samples(self, width, height, n_samples, **kw):
        cfg = self.cfg(width, height, n_samples=n_samples, **kw)
        out = np.zeros((n_samples, height, width, 4), np.float32)
        r = self.L.pto_render_samples(self.ctx, C.byref(cfg), _p(out))
        assert r == 0, r
        return out

    def primary_ray(self, width, height, pixel, sample, **kw):
        cfg = self.cfg(width, height, **kw)
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        self.L.pto_primary_ray(self.ctx, C.byref(cfg), pixel, sample, _p(o), _p(d))
        return o, d

    def camera_input(self, event, a=0.0, b=0.0, dt=0.0):
        """Camera::input's dispatch (camera.rs:56-92); event codes as include/pt_api.h's pt_event"""
        if event == 0:
            self.L.pto_camera_rotate(self.ctx, a, b, dt); return True
        moves = {1: (0.0, 1.0), 2: (0.0, -1.0), 3: (-1.0, 0.0), 4: (1.0, 0.0)}
        if event in moves:
            self.L.pto_camera_move(self.ctx, moves[event][0], moves[event][1], dt); return True
        return False

    def camera_angles(self):
        out = np.zeros(2, np.float32)
        self.L.pto_camera_angles(self.ctx, _p(out))
        return out

    def inv_projection(self):
        m = np.zeros(16, np.float32)
        self.L.pto_inv_projection(self.ctx, _p(m))
        return m

    def create_ray(self, s, t):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        self.L.pto_create_ray(self.ctx, s, t, _p(o), _p(d))
        return o, d

    def camera_matrices(self):
        m = np.zeros(12, np.float32)
        ip = np.zeros(16, np.float32)
        rm = np.zeros(16, np.float32)
        self.L.pto_camera_matrices(self.ctx, _p(m), _p(ip), _p(rm))
        return m.reshape(3, 4), ip.reshape(4, 4).T.copy(), rm.reshape(4, 4).T.copy()

    def trace_closest(self, o, d, tmax=None, which=0):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = np.full(n, np.inf, np.float32) if tmax is None else np.ascontiguousarray(tmax, np.float32)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        inst = np.zeros(n, np.uint32); prim = np.zeros(n, np.uint32)
        nrm = np.zeros((n, 3), np.float32); front = np.zeros(n, np.uint8)
        r = self.L.pto_trace_closest(self.ctx, which, n, _p(o), _p(d), _p(tm), _p(t), _p(u), _p(v), _p(inst), _p(prim), _p(nrm), _p(front))
        assert r == 0
        return dict(t=t, u=u, v=v, inst=inst, prim=prim, normal=nrm, front=front)

    def trace_any(self, o, d, tmax, which=0):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        tm = np.ascontiguousarray(tmax, np.float32)
        hit = np.zeros(o.shape[0], np.uint8)
        r = self.L.pto_trace_any(self.ctx, which, o.shape[0], _p(o), _p(d), _p(tm), _p(hit))
        assert r == 0
        return hit

    def integrate(self, o, d, pixel, sample, draws_consumed=1, **kw):
        cfg = self.cfg(1, 1, **kw)
        col = np.zeros(4, np.float32); pos = np.zeros(4, np.float32); idv = np.zeros(1, np.uint8)
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        self.L.pto_integrate(self.ctx, C.byref(cfg), _p(o), _p(d), pixel, sample, draws_consumed, _p(col), _p(pos), _p(idv))
        return col, pos, int(idv[0])

    def blas_count(self):
        return self.L.pto_blas_count(self.ctx)

    def blas_dump(self, blas, which=0, cap=1 << 20):
        nn = C.c_uint32(); root = C.c_uint32(); nids = C.c_uint32()
        boxes = np.zeros((cap, 6), np.float32); kind = np.zeros(cap, np.uint32); a = np.zeros(cap, np.uint32); b = np.zeros(cap, np.uint32)
        ids = np.zeros(cap, np.uint32)
        r = self.L.pto_blas_dump(self.ctx, which, blas, C.byref(nn), C.byref(root), _p(boxes), _p(kind), _p(a), _p(b), C.byref(nids),
                                 _p(ids), cap, cap)
        assert r == 0, r
        n = nn.value
        return dict(root=root.value, boxes=boxes[:n].copy(), kind=kind[:n].copy(), a=a[:n].copy(), b=b[:n].copy(),
                    prim_ids=ids[:nids.value].copy())

    def tlas_dump(self, which=0, cap=1 << 16):
        nn = C.c_uint32(); root = C.c_uint32()
        boxes = np.zeros((cap, 6), np.float32); kind = np.zeros(cap, np.uint32); a = np.zeros(cap, np.uint32); b = np.zeros(cap, np.uint32)
        r = self.L.pto_tlas_dump(self.ctx, which, C.byref(nn), C.byref(root), _p(boxes), _p(kind), _p(a), _p(b), cap)
        assert r == 0, r
        n = nn.value
        return dict(root=root.value, boxes=boxes[:n].copy(), kind=kind[:n].copy(), a=a[:n].copy(), b=b[:n].copy())

    def tlas_instances(self, which=0, cap=1 << 16):
        n = C.c_uint32()
        m = np.zeros((cap, 3, 4), np.float32); inv = np.zeros((cap, 3, 4), np.float32)
        r = self.L.pto_tlas_instances(self.ctx, which, C.byref(n), _p(m), _p(inv), cap)
        assert r == 0, r
        return dict(matrix=m[:n.value].copy(), inv_matrix=inv[:n.value].copy())

    def light_cdf(self, cap=1 << 20):
        n = C.c_uint32(); mx = C.c_float()
        pdf = np.zeros(cap, np.float32); cdf = np.zeros(cap, np.float32); bl = np.zeros(cap, np.uint32); pr = np.zeros(cap, np.uint32)
        r = self.L.pto_light_cdf(self.ctx, C.byref(n), _p(pdf), _p(cdf), _p(bl), _p(pr), C.byref(mx), cap)
        assert r == 0
        k = n.value
        return dict(pdf=pdf[:k].copy(), cdf=cdf[:k].copy(), blas=bl[:k].copy(), prim=pr[:k].copy(), max=mx.value)

    def triangle(self, blas, prim, which=0):
        out = np.zeros(36, np.float32)
        r = self.L.pto_triangle_dump(self.ctx, which, blas, prim, _p(out))
        assert r == 0
        return out

    def material_eval(self, material, incoming, normal, front, pixel, sample, draws_consumed=0, seed=DEFAULT_SEED):
        out = np.zeros(9, np.float32)
        i = np.ascontiguousarray(incoming, np.float32); n = np.ascontiguousarray(normal, np.float32)
        r = self.L.pto_material_eval(self.ctx, material, _p(i), _p(n), int(front), seed, pixel, sample, draws_consumed, _p(out))
        assert r == 0
        return out


# ---- free functions (samplers / math hooks)
def ss_sobol_raw(n_points, index, seed):
    out = np.zeros(3, np.uint32)
    lib().pto_ss_sobol_raw(n_points, index, seed, _p(out))
    return tuple(int(x) for x in out)


def ss_sobol(n_points, index, seed):
    out = np.zeros(2, np.float32)
    lib().pto_ss_sobol(n_points, index, seed, _p(out))
    return out


def math_batch(fn, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float32)
    o0 = np.zeros_like(a); o1 = np.zeros_like(a)
    lib().pto_math_batch(fn, a.size, _p(a), _p(b), _p(o0), _p(o1))
    return o0, o1


# ---- after the path (State::update / State::render)
def post_accumulate(inp, accum):
    inp = np.ascontiguousarray(inp, np.float32)
    accum = np.ascontiguousarray(accum, np.float32).copy()
    h, w = inp.shape[:2]
    lib().pto_post_accumulate(w, h, _p(inp), _p(accum))
    return accum


def post_velocity(position, last_inv_projection):
    position = np.ascontiguousarray(position, np.float32)
    h, w = position.shape[:2]
    v = np.zeros((h, w, 2), np.float32)
    lib().pto_post_velocity(w, h, _p(position), _p(np.ascontiguousarray(last_inv_projection, np.float32)), _p(v))
    return v


def post_reproject(inp, accum, velocity, ident):
    inp = np.ascontiguousarray(inp, np.float32); accum = np.ascontiguousarray(accum, np.float32)
    velocity = np.ascontiguousarray(velocity, np.float32); ident = np.ascontiguousarray(ident, np.uint32)
    h, w = inp.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    lib().pto_post_reproject(w, h, _p(inp), _p(accum), _p(velocity), _p(ident), _p(out))
    return out


def post_tonemap(accum):
    accum = np.ascontiguousarray(accum, np.float32)
    h, w = accum.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    lib().pto_post_tonemap(w, h, _p(accum), _p(out))
    return out


def post_rgb8(accum):
    accum = np.ascontiguousarray(accum, np.float32)
    h, w = accum.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    lib().pto_post_rgb8(w, h, _p(accum), _p(out))
    return out
