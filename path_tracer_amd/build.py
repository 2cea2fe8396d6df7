"""In-tree build of libptmi.so (hipcc, gfx950).  The library is the product: there is no Python or CPU fallback."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libptmi.so")
_SOURCES = ["pt_api.cpp", "pt_scene.cpp", "pt_kernels.hip", "pt_post.hip", "pt_png.cpp", "pt_png.h", "pt_math.h", "pt_types.h", "pt_scene.h", "pt_kernels.h", "pt_materials.h", "Makefile"]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, s) for s in _SOURCES] + [os.path.join(_HERE, "..", "include", "pt_api.h")]
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def build(force: bool = False) -> str:
    if force or is_stale():
        hipcc = "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            raise RuntimeError("hipcc not found: libptmi.so cannot be built (and there is no fallback path)")
        subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), check=True)
    return LIB_PATH
