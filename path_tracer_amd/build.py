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


ROOT = os.path.dirname(_HERE)
HEADLESS = os.path.join(ROOT, "examples", "headless")


def build_host_driver(force: bool = False) -> str:
    """examples/headless: the C++ host side (include/ptmi.hpp) driving the C-ABI the way src/main.rs drives its integrator."""
    src = os.path.join(ROOT, "examples", "headless.cpp")
    deps = [src, os.path.join(ROOT, "include", "ptmi.hpp"), os.path.join(ROOT, "include", "pt_api.h"), LIB_PATH]
    if force or not os.path.exists(HEADLESS) or any(os.path.getmtime(d) > os.path.getmtime(HEADLESS) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-o", HEADLESS, src,
                        "-L" + _HERE, "-lptmi", "-Wl,-rpath,$ORIGIN/../path_tracer_amd"], check=True)
    return HEADLESS
