"""CPU: libptmi's HOST code (OBJ reader, BVH builders + flattening, camera, PNG encoder) under AddressSanitizer + UBSan
(`make -C path_tracer_amd/csrc host-asan`; GPU ASan does not exist on the pool, and these files are where user-supplied bytes enter
the library).  Reference for what is being parsed: load_obj, src/tlas/tlas_bvh/blas.rs:44-131."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_host import OBJ_TEXT

CSRC = os.path.join(ROOT, "path_tracer_amd", "csrc")
EXE = os.path.join(ROOT, "path_tracer_amd", "host_sanitize")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module")
def exe():
    r = subprocess.run(["make", "-C", CSRC, "host-asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EXE


def _run(args, cwd=None):
    r = subprocess.run(args, capture_output=True, text=True, env=ENV, cwd=cwd, timeout=600)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stderr[-3000:])
    return r.stdout


def test_obj_reader_and_builders_on_valid_malformed_and_mutated_files(exe, tmp_path):
    rng = np.random.default_rng(5)
    files = []

    def put(name, data):
        p = tmp_path / name
        p.write_bytes(data if isinstance(data, bytes) else data.encode())
        files.append(str(p))

    put("ok.obj", OBJ_TEXT)
    # the error cases of tests/test_host.py (where the reference panics) and a few more shapes of nonsense
    for i, text in enumerate(["v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", "v 0 0 zero\n", "v 0 0 0\nf 1//1 2//1 3//1\n", "# nothing\n", "", "f\n", "v\nvn\nf 1//1\n",
                              "v 1 2\nv 1 2 3 4 5\n", "vn 0 0 0\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//1 3//1\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 99999999999//1\n",
                              "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf -9//1 2//1 3//1\n", "v 1e9999 0 0\nv nan inf -inf\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n",
                              "f 1/2/3/4/5 6/7 8\n", "v 0 0 0\n" * 3 + "vn 0 0 1\n" + "f " + " ".join("1//1" for _ in range(5000)) + "\n", "\x00\xff\xfe binary\n"]):
        put(f"bad{i}.obj", text)
    base = OBJ_TEXT.encode()
    for k in range(300):                                    # byte-mutated copies: flips, deletions, duplications, truncations
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 6))):
            op, pos = int(rng.integers(0, 4)), int(rng.integers(0, len(b)))
            if op == 0: b[pos] = int(rng.integers(0, 256))
            elif op == 1: del b[pos:pos + int(rng.integers(1, 8))]
            elif op == 2: b[pos:pos] = b[pos:pos + int(rng.integers(1, 16))]
            else: b = b[:max(pos, 1)]
            if not b: b = bytearray(b"v")
        put(f"mut{k}.obj", bytes(b))
    out = _run([exe, "obj"] + files + [str(tmp_path / "does_not_exist.obj")])
    import json
    res = json.loads(out.strip().split("\n")[-1])
    assert res["parsed"] >= 1 and res["rejected"] >= 15 and res["parsed"] + res["rejected"] == len(files) + 1


def test_camera_walk_and_png_encoder(exe, tmp_path):
    assert "checksum" in _run([exe, "walk", "200", "7"])
    assert "checksum" in _run([exe, "walk", "2000", "11"])
    assert '"bytes": 172800' in _run([exe, "png", "320", "180", str(tmp_path / "a.png")])
    assert "bytes" in _run([exe, "png", "1", "1", str(tmp_path / "b.png")])
    assert "error" in _run([exe, "png", "8", "8", str(tmp_path / "no_dir" / "c.png")])


def test_forked_sweep_builder_under_asan_and_tsan(exe):
    """The SAH sweep forks its subtrees onto threads (PTMI_BUILD_THREADS): no race, no bad access, and the arena does not depend on
    the thread count."""
    import json
    r = subprocess.run(["make", "-C", CSRC, "host-tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    arenas = set()
    for binary in (exe, EXE + "_tsan"):
        for threads in ("1", "2", "16"):
            env = dict(ENV, PTMI_BUILD_THREADS=threads, TSAN_OPTIONS="halt_on_error=1:exitcode=97")
            r = subprocess.run([binary, "soup", "30000", "5"], capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (binary, threads, r.returncode, r.stderr[-3000:])
            arenas.add(json.loads(r.stdout.strip().split("\n")[-1])["arena"])
    assert len(arenas) == 1, arenas
