import importlib
import os
import sys

import pytest

# every device output the Python wrapper allocates starts as 0x7B bytes (capi.POISON_OUTPUTS, read at import): a pixel no kernel
# writes cannot pass for a correct one because the caching allocator handed out the block of an earlier, correct run
os.environ.setdefault("SLR_POISON_OUTPUTS", "1")
# ... and the library's own scratch buffers (phases, codes, buckets, staging) before every call that gets them (capi.POISON_SCRATCH)
os.environ.setdefault("SLR_POISON_SCRATCH", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def slr():
    """the product package (directory name contains '-', hence importlib)"""
    return importlib.import_module("structure-light-reconstructor_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("structure-light-reconstructor_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle -- test infrastructure only"""
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def ctx(slr):
    """one slr_ctx on cuda:0; loudly fails (never skips) when the HIP library or the GPU is missing"""
    c = slr.Context(0)
    yield c
    c.close()
