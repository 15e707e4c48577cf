"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls: no GPU here)."""
import ctypes
import glob
import os
import re

from conftest import ROOT


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        names |= set(re.findall(r"\b(epn_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    from epn_pointcloud_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 14
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.EXPORTS)


def test_binding_loads_and_reports_version():
    from epn_pointcloud_amd import _lib
    lib = _lib.get_lib()
    assert b"gfx950" in lib.epn_version()
    assert lib.epn_strerror(0) == b"success"
    assert b"workspace" in lib.epn_strerror(-2)


def test_workspace_query_is_host_only():
    from epn_pointcloud_amd import _lib
    lib = _lib.get_lib()
    d = _lib.InterDesc()
    d.b, d.p1, d.p2, d.nn, d.na, d.ks, d.cin, d.cout = 2, 256, 128, 16, 60, 24, 1, 8
    d.sigma = 0.08
    n = lib.epn_inter_workspace_bytes(ctypes.byref(d))
    assert n >= 2 * 128 * 60 * 1 * 24 * 4           # generic path: grouped features are materialised


def test_packed_position_table_is_the_documented_permutation():
    """epn_inter_packed_position (host-only) against the formula in include/epn_so3conv.h, for both channel-group widths
    and a ragged second kernel-point tile; refusals for widths the packed grouping does not serve."""
    import numpy as np
    from epn_pointcloud_amd import _lib
    lib = _lib.get_lib()
    for cin, ks in ((32, 24), (64, 24), (96, 24), (256, 24), (64, 16), (64, 12), (128, 20)):
        pos = np.full(cin * ks, -1, dtype=np.int32)
        assert lib.epn_inter_packed_position(cin, ks, pos.ctypes.data) == 0
        cg = 4 if cin % 64 == 0 else 2
        w0 = min(ks, 16)
        for c in range(cin):
            cl = c % (16 * cg)
            slot = c - cl + 16 * (cl % cg) + cl // cg
            for k in range(ks):
                want = slot * w0 + k if k < 16 else w0 * cin + slot * (ks - 16) + (k - 16)
                assert pos[c * ks + k] == want
        assert sorted(pos.tolist()) == list(range(cin * ks))
    buf = np.zeros(16 * 24, dtype=np.int32)
    assert lib.epn_inter_packed_position(16, 24, buf.ctypes.data) != 0      # cin % 32
    assert lib.epn_inter_packed_position(64, 22, buf.ctypes.data) != 0      # ks % 4
    assert lib.epn_inter_packed_position(64, 24, None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    from epn_pointcloud_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libepn_so3conv.so")
    try:
        _lib.get_lib()
    except RuntimeError as e:
        assert "no CPU/eager fallback" in str(e)
    else:
        raise AssertionError("get_lib() must raise when the HIP library is missing")
