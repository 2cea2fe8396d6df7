import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    """u32 bit patterns of an f32 array.  A NaN is a NaN: x86 creates 0xFFC00000 ("real indefinite") where gfx950 creates
    0x7FC00000, and neither the reference nor this library ever looks at a NaN's sign or payload."""
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        return a
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ba, bb = bits(a), bits(b)
    if not np.array_equal(ba, bb):
        bad = np.argwhere(ba != bb)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {ba.size} words differ; first at {i}: {a[i]!r} vs {b[i]!r}")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def cornell64():
    from path_tracer_amd import scenes
    return scenes.cornell_box(64, 64)


@pytest.fixture(scope="session")
def cornell256():
    from path_tracer_amd import scenes
    return scenes.cornell_box(256, 256)
