import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def has_gpu() -> bool:
    return os.path.exists("/dev/kfd") and os.path.exists("/dev/dri")


if has_gpu():
    # Tests that hand torch tensors to the library (the multi-GPU exchange) need torch's bundled HIP runtime to be the
    # one in the process: import it before anything loads libnrtgpu.so (as bench.py does), whatever order the test
    # files run in.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o


@pytest.fixture
def dev_lib():
    """The test talks to the DEVELOPMENT library (include/nrtgpu_dev.h: the product sources + test hooks) instead of the product
    library: for the few GPU tests that park callers behind nrtgpu_debug_hold_coalescers or count live handles.  Everything the
    test creates (contexts, segments) must be created and closed inside it -- handles of one library mean nothing to the other."""
    from nrtsearch_amd import _lib, build

    build.build_dev()
    prev = _lib._lib
    _lib._lib = _lib.load_dev()
    try:
        yield _lib._lib
    finally:
        _lib._lib = prev
