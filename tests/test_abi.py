"""CPU: the C-ABI library builds for gfx950, loads through ctypes without a GPU and exports exactly the entry
points include/dtk.h declares; the Python binding lists a signature for each; argument validation paths that do not
touch the device behave (error code + message)."""
import ctypes

import pytest

import __graft_entry__ as entry
from dino_tracker_amd import _lib


@pytest.fixture(scope="module")
def handle():
    entry.build()
    return _lib.lib()


def test_exports_match_header(handle):
    declared = entry.declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(handle, name), name
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_error_paths(handle):
    assert handle.dtk_version() == 1
    g = _lib.make_geom(4, 30, 140, 210)  # C=30 is not a multiple of 4
    rc = handle.dtk_sample_points(g, 1, 1, 1, None, 1, 1, None)
    assert rc == -1 and b"multiple of 4" in handle.dtk_last_error()
    g = _lib.make_geom(4, 32, 140, 210)
    g.ph += 1
    rc = handle.dtk_track(g, 1, 1, None, 1, 1, None, 1, None, 1, 1, None, 0, 0, 1, 1024, None)
    assert rc == -1 and b"inconsistent" in handle.dtk_last_error()
    assert handle.dtk_track_workspace_bytes(_lib.make_geom(4, 32, 140, 210), 100, 0) == 100 * (576 + 1) * 4


def test_hot_path_refuses_cpu_tensors():
    import torch
    from dino_tracker_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pack_features(torch.zeros(1, 4, 2, 2))
