"""Run-time switches of the package, in ONE place.

SUPPORTED switches (read in every build; INTEGRATION.md "Run-time switches" documents them):
    EPN_LIB, EPN_INTER_MODE, EPN_INTER_BWD_DATA, EPN_INTRA_MODE, EPN_GEMM_FP32, EPN_DETERMINISTIC, EPN_SKIP_STREAM,
    EPN_DP_BACKEND, EPN_DP_SHARE_GPU, EPN_DP_TIMEOUT (multi-rank rehearsal of tests / bench), EPN_BENCH_DETAIL (bench.py)

A/B switches -- forms that were measured against each other and settled (DESIGN.md 5.1 lists what each one compared and by how
much) -- are read ONLY when EPN_AB=1 is set (the test suite and tools/ set it to compare the two forms of a kernel pair); in a
default process `ab()` returns the settled default without looking at the environment, so a stray variable cannot move a
production run onto a slower or less-tested form."""
import os

AB_DEFAULTS = {
    "EPN_SHARE_INPUT_GRAD": "1",     # skip branch's input gradient accumulated by the data-gradient scatter
    "EPN_GROUP_PACKED": "1",         # grouped features in the grouping kernel's own column order
    "EPN_EPILOGUE_STATS": "1",       # norm statistics from the producers' epilogues
    "EPN_NORM_PAIR": "1",            # block tail (final norm + residual + skip norm) in one pass per direction
    "EPN_NORM_ON_LOAD": "1",         # first norm + leaky_relu applied as the basis change loads its rows
    "EPN_NORM_BWD_EPILOGUE": "1",    # first norm's backward sums from the inverse basis change's epilogue
    "EPN_SPECTRAL_WEIGHTS": "fused", # spectral weights by one kernel | "torch"
    "EPN_C1_SAVE": "1",              # first layer keeps its grouped values for the weight gradient
    "EPN_C1_DW": "gemm",             # ... and computes it as a library TN GEMM | "kernel"
    "EPN_C1_MFMA": "1",              # first layer's forward on the matrix pipe
    "EPN_ATTN_POOL": "1",            # 3DMatch head's softmax-over-anchors pooling as one kernel per direction
    "EPN_POINTNET": "auto",          # PointnetSO3Conv: GEMM-composed from 16 k rows | "gemm" | "fused"
    "EPN_HEAD_BF16": "1",            # bf16 networks: rotation head reads bf16 features itself
    "EPN_REG_MLP_BF16": "1",         # ... and its anchor-pair MLP keeps bf16 values
    "EPN_CHECK_AMAX": "0",           # debug: re-derive every maximum a two-piece fp16 GEMM consumes and assert (synchronises; eager only)
}


_RESOLVED = {}        # name -> value, resolved once per process (the switches are read on hot forward paths)
_WARNED = False


def _warn_ignored():
    """One warning per process when A/B variables are set but EPN_AB=1 is not: earlier rounds honoured them unconditionally, and a
    tool or a debugging workaround that still sets one would otherwise silently measure the default form (advisor, round 5)."""
    global _WARNED
    if _WARNED:
        return
    _WARNED = True
    stray = sorted(k for k in AB_DEFAULTS if k in os.environ and os.environ[k] != AB_DEFAULTS[k])
    if stray:
        import warnings
        warnings.warn("epn_pointcloud_amd: " + ", ".join(f"{k}={os.environ[k]}" for k in stray) + " ignored -- A/B switches are "
                      "read only when EPN_AB=1 is set (INTEGRATION.md 'Run-time switches'); the settled defaults are in effect",
                      RuntimeWarning, stacklevel=3)


def ab(name):
    """Value of an A/B switch: the settled default unless the process opted into A/B mode with EPN_AB=1.  In A/B mode the
    environment is looked at on every call (tests flip switches with monkeypatch between calls); outside it the answer is the
    cached default."""
    if os.environ.get("EPN_AB", "0") == "1":
        return os.environ.get(name, AB_DEFAULTS[name])
    v = _RESOLVED.get(name)
    if v is None:
        _warn_ignored()
        v = _RESOLVED[name] = AB_DEFAULTS[name]
    return v
