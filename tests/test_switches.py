"""The package's run-time switches (epn_pointcloud_amd/_ab.py): A/B variables move a process onto the other form of a settled
kernel pair ONLY in A/B mode (EPN_AB=1: tests, tools); a default process ignores them.  And the documented list is the real list:
every EPN_* variable the package reads is either a supported switch or an A/B switch."""
import os
import re

from conftest import ROOT

SUPPORTED = {"EPN_LIB", "EPN_INTER_MODE", "EPN_INTER_BWD_DATA", "EPN_INTRA_MODE", "EPN_GEMM_FP32", "EPN_DETERMINISTIC",
             "EPN_SKIP_STREAM", "EPN_DP_BACKEND", "EPN_DP_SHARE_GPU", "EPN_DP_TIMEOUT", "EPN_BENCH_DETAIL", "EPN_AB"}
INTERNAL = {"EPN_DP_CHILD",                                     # set by dp.launch for its children
            "EPN_TUNING", "EPN_BUILD_TAG", "EPN_EXTRA_FLAGS",  # build.py (tools/ A/B builds)
            "EPN_ABI_VERSION"}                                  # a C macro named in a message, not a variable


def test_ab_switches_are_inert_outside_ab_mode(monkeypatch):
    from epn_pointcloud_amd import _ab
    monkeypatch.setenv("EPN_SHARE_INPUT_GRAD", "0")
    monkeypatch.setenv("EPN_AB", "0")
    assert _ab.ab("EPN_SHARE_INPUT_GRAD") == "1"
    monkeypatch.setenv("EPN_AB", "1")
    assert _ab.ab("EPN_SHARE_INPUT_GRAD") == "0"
    monkeypatch.delenv("EPN_SHARE_INPUT_GRAD")
    assert _ab.ab("EPN_SHARE_INPUT_GRAD") == "1"


def test_a_stray_ab_variable_is_reported_once(monkeypatch):
    """Outside A/B mode an A/B variable changes nothing -- and says so, once per process (advisor finding, round 5)."""
    import warnings
    from epn_pointcloud_amd import _ab
    monkeypatch.setenv("EPN_AB", "0")
    monkeypatch.setenv("EPN_NORM_PAIR", "0")
    monkeypatch.setattr(_ab, "_WARNED", False)
    monkeypatch.setattr(_ab, "_RESOLVED", {})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _ab.ab("EPN_NORM_PAIR") == "1"
        assert _ab.ab("EPN_GROUP_PACKED") == "1"
        assert _ab.ab("EPN_NORM_PAIR") == "1"
    msgs = [str(x.message) for x in w if "EPN_NORM_PAIR=0 ignored" in str(x.message)]
    assert len(msgs) == 1, [str(x.message) for x in w]


def test_every_variable_the_package_reads_is_listed():
    from epn_pointcloud_amd import _ab
    seen = set()
    for d, _, files in os.walk(os.path.join(ROOT, "epn_pointcloud_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                seen |= set(re.findall(r"\bEPN_[A-Z][A-Z0-9_]*\b", open(os.path.join(d, f)).read())) if f.endswith(".py") else set()
    seen |= set(re.findall(r"\bEPN_[A-Z][A-Z0-9_]*\b", open(os.path.join(ROOT, "bench.py")).read()))
    unknown = seen - SUPPORTED - INTERNAL - set(_ab.AB_DEFAULTS)
    assert not unknown, unknown
    assert len(SUPPORTED) <= 12
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in SUPPORTED | set(_ab.AB_DEFAULTS):
        assert name in text, f"{name} is not documented in INTEGRATION.md"


def test_the_library_reads_no_environment():
    for f in os.listdir(os.path.join(ROOT, "epn_pointcloud_amd", "csrc")):
        assert "getenv" not in open(os.path.join(ROOT, "epn_pointcloud_amd", "csrc", f)).read(), f
