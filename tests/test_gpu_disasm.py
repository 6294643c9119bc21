"""The two static checks of the fused decodes' device code (tests/test_capi_symbols.py: the ticket SGPR that nothing else may name,
the LDS read destinations that nothing may touch before their counted wait) -- here under -m gpu and on the library file that THIS
process has actually mapped on the GPU box, so that the run which precedes a bench covers them."""
import os

import pytest

from test_capi_symbols import check_lds_reads_in_flight, check_ticket_register

pytestmark = pytest.mark.gpu


def _mapped_library(ctx):
    """path of the libslr_hip.so mapped into this process (the context exists, so it is loaded)"""
    assert ctx is not None
    paths = {line.split()[-1] for line in open("/proc/self/maps") if line.rstrip().endswith("libslr_hip.so")}
    assert len(paths) == 1, paths
    return paths.pop()


def test_loaded_library_ticket_register(ctx, tmp_path):
    path = _mapped_library(ctx)
    assert os.path.samefile(path, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                               "structure-light-reconstructor_amd", "libslr_hip.so")), path   # the in-tree build
    assert check_ticket_register(path, tmp_path) >= 6


def test_loaded_library_lds_reads_in_flight(ctx, tmp_path):
    assert check_lds_reads_in_flight(_mapped_library(ctx), tmp_path) > 5000
