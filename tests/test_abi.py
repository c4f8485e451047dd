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
    exact = _lib.TrackOpts(0, 0, 0, 0)
    rc = handle.dtk_track(g, 1, 1, None, 1, 1, None, 1, None, 1, 1, None, exact, None, 1, 1024, None)
    assert rc == -1 and b"inconsistent" in handle.dtk_last_error()
    g = _lib.make_geom(4, 32, 140, 210)
    rc = handle.dtk_track(g, 1, 1, None, 1, 1, None, 1, None, 1, 1, None, _lib.TrackOpts(0, 0, 0, 7), None, 1, 1024, None)
    assert rc == -1 and b"bad options" in handle.dtk_last_error()
    assert handle.dtk_track_workspace_bytes(g, 100, exact) == 100 * (576 + 1) * 4
    # the MFMA workspace follows the round size (smaller rounds -> smaller staging)
    big = handle.dtk_track_workspace_bytes(g, 100000, _lib.TrackOpts(1, 0, 0, 0))
    small = handle.dtk_track_workspace_bytes(g, 100000, _lib.TrackOpts(1, 0, 1024, 0))
    assert 0 < small < big


def test_production_library_has_no_development_switches(handle):
    """The shipped libdtk.so is built without -DDTK_DEV: no getenv("DTK_DEBUG"), no debug exports."""
    import subprocess
    for name in ("dtk_debug_counters", "dtk_debug_track_counts"):
        assert not hasattr(handle, name), name
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_hot_path_refuses_cpu_tensors():
    import torch
    from dino_tracker_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pack_features(torch.zeros(1, 4, 2, 2))
