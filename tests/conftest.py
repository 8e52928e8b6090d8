import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the drop-in's seams in front of small pieces of work do not wait for a device that is still being opened
# (integration/c4gpu_shim.c, shim_ctx_nowait); the tests count what the device served, so here every seam waits
os.environ.setdefault("C4GPU_WAIT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI shared library (host-side entry points work without a GPU)."""
    from exonerate_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _abi.load()


@pytest.fixture(scope="session")
def params(lib):
    from exonerate_amd import _abi
    p = _abi.Params()
    lib.c4gpu_params_default(p)
    return p
