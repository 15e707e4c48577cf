"""bf16 feature path (BASELINE configs 3-4: bf16 features, fp32 accumulation), through the C ABI.

Tolerance.  A bf16 value carries 8 significant bits: rounding once is a relative error of at most 2^-9 = 1.95e-3.  The
HIP path rounds (i) the grouped features G, (ii) the weights W, (iii) the layer output; the oracle below is the fp32
restatement fed the SAME bf16-rounded inputs (features and weights), so what is compared is (i) + (iii) plus the
re-association inside the GEMMs: an output element is a sum of K = cin*ks products each carrying an independent 2^-9
relative error, i.e. ~ 2^-9 * |term| * sqrt(K) in absolute terms, bounded here by BF16_TOL = 2e-2 (a few 2^-9 units,
with the random-walk factor folded in) relative to the largest reference magnitude of the tensor.  Gradients see one
more rounding (dOut, dG) and use the same bound on the relative L2 error.
"""
import numpy as np
import pytest
import torch

from conftest import unit_ball_cloud
from oracle import so3conv_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy
BF16_TOL = 2e-2


def _mods(vgtk_alias):
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    return sptk, zptk


def r16(t):
    """Round to bf16 and back (what the HIP path stores)."""
    return t.to(torch.bfloat16).float()


def rel_max(got, want):
    return (got.float().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


def rel_l2(got, want):
    return ((got.float().cpu() - want).norm() / want.norm().clamp_min(1e-12)).item()


# ------------------------------------------------------------------------------------------------ GEMM kernels
@pytest.fixture(params=["f16x2", "split", "native"])
def fp32_mode(request):
    """fp32 contractions: two fp16 pieces x three matrix products / lossless 3 x bf16 split / v_mfma_f32_32x32x2_f32."""
    from epn_pointcloud_amd import gemm
    old = gemm.FP32_MODE
    gemm.set_fp32_mode(request.param)
    yield request.param
    gemm.set_fp32_mode(old)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1000, 64, 256), (513, 192, 320), (256, 32, 64), (777, 320, 128), (100, 24, 40),
                                   (4096, 128, 1536), (1, 64, 64), (3000, 768, 32), (2000, 32, 32), (900, 64, 96),
                                   (1500, 200, 16)])
def test_gemm_nt_vs_fp64(gpu, fp32_mode, dt, M, N, K):
    from epn_pointcloud_amd import gemm
    if dt != torch.float32 and fp32_mode != "split":
        pytest.skip("mode only concerns fp32 operands")
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=gpu).to(dt)
    B = torch.randn(N, K, device=gpu).to(dt)
    C = gemm.gemm_nt(A, B)
    ref = A.double() @ B.double().t()
    tol = 1e-5 if dt == torch.float32 else 1e-2         # fp32: exact-f32 MFMA; bf16: one output rounding (2^-9)
    assert (C.double() - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1024, 64, 256), (992, 192, 320), (4096, 256, 1536), (1600, 200, 64), (96, 24, 40),
                                   (3840, 128, 128), (7680, 32, 64), (98304, 64, 64)])
def test_gemm_nt_epilogue_column_statistics(gpu, fp32_mode, dt, M, N, K):
    """col_stats of the NT kernels (per-column sum / sum of squares of every 32-row block of C, taken from the accumulators
    in the epilogue) against the same sums of the tensor the kernel stored -- all tile shapes incl. a ragged last row tile
    (M = 992), ragged column tiles (N = 200, 24), the generic kernel (K = 40), bf16 outputs (sums of the ROUNDED values) --
    and epn_stats_finish's group sums (BatchNorm: 1 group, InstanceNorm: one per cloud; M = 98304: the two-level reduction
    of long partial lists)."""
    from epn_pointcloud_amd import gemm, ops
    if dt != torch.float32 and fp32_mode != "split":
        pytest.skip("mode only concerns fp32 operands")
    torch.manual_seed(M + N + K)
    A = (torch.randn(M, K, device=gpu) + 0.3).to(dt)
    B = torch.randn(N, K, device=gpu).to(dt)
    C, part = gemm.gemm_nt(A, B, col_stats=True)
    assert torch.equal(C, gemm.gemm_nt(A, B))                       # the output itself is untouched
    assert tuple(part.shape) == (M // 32, N, 2)
    blocks = C.double().reshape(M // 32, 32, N)
    want = torch.stack((blocks.sum(1), (blocks * blocks).sum(1)), -1)
    scale = want.abs().amax(dim=(0, 1))
    assert ((part.double() - want).abs().amax(dim=(0, 1)) <= 1e-5 * scale).all()
    for groups in (1, 2, 4):
        if (M // 32) % groups:
            continue
        sums = ops.sums_from_partials(part, groups, M // groups, N)
        ref = want.reshape(groups, M // 32 // groups, N, 2).sum(1)
        assert ((sums.double() - ref).abs() <= 2e-5 * ref.abs().amax(dim=(0, 1))).all()
    assert ops.sums_from_partials(part, 1, M + 32, N) is None        # partials of another shape are refused


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R_,N1,N2", [(4096, 64, 512), (2048, 128, 192), (960, 32, 768), (1024, 256, 256), (100, 20, 36),
                                      (61440, 64, 1536), (32, 8, 8),
                                      # narrow outputs (1x1-convolution weight gradients; bf16: every tile of the ring-of-
                                      # stages kernel, row counts that end inside a stage, ragged N1 / N2, and enough
                                      # rows for 16+ partial slabs = the shared-quad reduction)
                                      (1184, 32, 32), (7200, 64, 32), (4128, 32, 64), (9696, 64, 64), (3232, 128, 64),
                                      (2080, 256, 64), (2080, 24, 128), (4160, 56, 120), (8352, 128, 128),
                                      (4128, 256, 128), (99968, 40, 24), (245760, 32, 32), (122880, 64, 64),
                                      (61440, 256, 128)])
def test_gemm_tn_vs_fp64(gpu, fp32_mode, dt, R_, N1, N2):
    from epn_pointcloud_amd import gemm
    if dt != torch.float32 and fp32_mode != "split":
        pytest.skip("mode only concerns fp32 operands")
    torch.manual_seed(R_ + N1 + N2)
    X = torch.randn(R_, N1, device=gpu).to(dt)
    Y = torch.randn(R_, N2, device=gpu).to(dt)
    C = gemm.gemm_tn(X, Y)
    ref = X.double().t() @ Y.double()
    assert (C.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() * max(1.0, (R_ / 4096) ** 0.5)
    # the split over R is summed in a fixed order: bitwise repeatable
    assert torch.equal(C, gemm.gemm_tn(X, Y))


def test_gemm_grouped_and_transpose(gpu, fp32_mode):
    from epn_pointcloud_amd import gemm
    torch.manual_seed(3)
    probs = []
    for d in (1, 3, 3, 4, 5):
        probs.append((torch.randn(640 * d, 64 * d, device=gpu), torch.randn(64 * d, 64 * d, device=gpu), None))
    outs = gemm.gemm_nt_grouped(probs)
    for (A, B, _), C in zip(probs, outs):
        ref = A.double() @ B.double().t()
        assert (C.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    W = torch.randn(96, 200, device=gpu)
    assert torch.equal(gemm.transpose_cast(W, torch.float32), W.t().contiguous())
    assert torch.equal(gemm.transpose_cast(W, torch.bfloat16), W.t().contiguous().bfloat16())
    assert torch.equal(gemm.cast(W, torch.bfloat16), W.bfloat16())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_tn_grouped(gpu, fp32_mode, dt):
    """One launch for the five weight-gradient GEMMs of a spectral IntraSO3Conv layer (R = pts*d rows, d*c x d*c outputs)."""
    from epn_pointcloud_amd import gemm
    if dt != torch.float32 and fp32_mode != "split":
        pytest.skip("mode only concerns fp32 operands")
    torch.manual_seed(8)
    for w1, w2 in ((64, 32), (256, 256), (512, 512)):   # (512, 512): every output >= 512 wide -- the split form pre-splits X
        probs = [(torch.randn(2048 * d, w1 * d, device=gpu).to(dt), torch.randn(2048 * d, w2 * d, device=gpu).to(dt))
                 for d in (1, 3, 3, 4, 5)]
        outs = gemm.gemm_tn_grouped(probs)
        for (X, Y), C in zip(probs, outs):
            ref = X.double().t() @ Y.double()
            assert (C.double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
        again = gemm.gemm_tn_grouped(probs)
        assert all(torch.equal(a, b) for a, b in zip(outs, again))     # fixed-order reduction: bitwise repeatable


@pytest.mark.parametrize("M,N,K", [(8192, 128, 1536), (4096, 256, 6144), (2048, 64, 32768), (4096, 768, 256)])
def test_split_gemm_has_fp32_accuracy(gpu, M, N, K):
    """The split form is an fp32 GEMM, not a bf16 one: against fp64 its rms error is no worse than that of the native
    fp32 MFMA kernel (measured 0.85-0.95x), on operands spanning twelve decades (row scales 1e-6 .. 1e6), and the
    bf16-rounded product -- what a bf16 GEMM would return -- is 1000x further away."""
    from epn_pointcloud_amd import gemm
    torch.manual_seed(K)
    A = torch.randn(M, K, device=gpu) * (10.0 ** torch.linspace(-6, 6, M, device=gpu))[:, None]
    B = torch.randn(N, K, device=gpu)
    ref = A.double() @ B.double().t()
    scale = ref.pow(2).mean(1, keepdim=True).sqrt()                    # per-row rms: rows differ by 1e12
    err = {}
    old = gemm.FP32_MODE
    try:
        for mode in ("native", "split"):
            gemm.set_fp32_mode(mode)
            C = gemm.gemm_nt(A, B)
            err[mode] = ((C.double() - ref) / scale).pow(2).mean().sqrt().item()
    finally:
        gemm.set_fp32_mode(old)
    bf = ((A.bfloat16().double() @ B.bfloat16().double().t() - ref) / scale).pow(2).mean().sqrt().item()
    assert err["split"] <= 1.1 * err["native"], err
    assert err["split"] < 2e-6 * max(1.0, (K / 4096) ** 0.5)
    assert bf > 300 * err["split"]
    # weight-gradient form (both operands split in registers), contraction over 123k rows: 1e-5 of the output scale
    X = torch.randn(122880, 64, device=gpu)
    Y = torch.randn(122880, 512, device=gpu)
    ref = X.double().t() @ Y.double()
    for mode in ("native", "split"):
        gemm.set_fp32_mode(mode)
        try:
            C = gemm.gemm_tn(X, Y)
        finally:
            gemm.set_fp32_mode(old)
        assert ((C.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item() < 1e-5


@pytest.mark.parametrize("net", ["cls", "reg"])
def test_producer_side_maxima_are_the_maxima(gpu, monkeypatch, net):
    """Every max|x| a producing kernel leaves behind for the two-piece fp16 GEMMs (ops._tag_amax: basis change, block tail,
    norm backward) IS the maximum of the tensor it tags -- over a whole classification step (forward + backward, 4 clouds).
    An under-reported maximum is silent until an operand exceeds twice it and overflows fp16: round 5 shipped one for a day
    (a vector-subscript bit cast that compiled to element 0: every fourth value only; up to 1.7x low on gradients, one
    non-finite training step in ten)."""
    from epn_pointcloud_amd import models as M, ops, schedule as S
    seen = []
    real = ops._tag_amax

    def recording(t, amax):
        seen.append((t, amax))
        real(t, amax)

    monkeypatch.setattr(ops, "_tag_amax", recording)
    torch.manual_seed(3)
    pts = S.synthetic_clouds(4, 1024, gpu, seed=5)
    if net == "cls":
        layers = S.cls_so3net_schedule(1024)
        m = S.set_feature_dtype(M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention").to(gpu).train(), torch.float32)
        loss = torch.nn.functional.cross_entropy(m(pts)[0], torch.tensor([1, 2, 3, 4], device=gpu))
    else:                                   # the rotation network in fp32: 32-channel layers (two points per basis-change task), K = 64
        m = S.set_feature_dtype(M.RegSO3ConvModel(S.reg_so3net_schedule(1024)).to(gpu).train(), torch.float32)
        out = m(pts.view(2, 2, 1024, 3))
        loss = out[0].square().mean() + out[1].square().mean()
    loss.backward()
    assert len(seen) >= 30, len(seen)
    for t, amax in seen:
        true = t.float().abs().max().item()
        assert abs(amax.item() - true) <= 1e-6 * max(true, 1e-30), (tuple(t.shape), amax.item(), true)


def test_absmax_pass_is_exact(gpu):
    """epn_absmax_f32 (the pass an entry point makes when nobody supplies a maximum): contiguous, strided rows, a tail that is
    not a multiple of four, non-finite elements left out, an all-zero tensor."""
    from epn_pointcloud_amd import gemm
    torch.manual_seed(9)
    for shape, sl in (((245760, 256), None), ((4099, 67), None), ((8192, 160), slice(0, 128)), ((3, 5), None)):
        t = torch.randn(*shape, device=gpu) * 3.0
        t[torch.randint(0, shape[0], (1,)), torch.randint(0, shape[1] if sl is None else 128, (1,))] = -77.5
        v = t if sl is None else t[:, sl]
        assert gemm.absmax(v).item() == v.abs().max().item(), shape
    t = torch.randn(1024, 64, device=gpu)
    t[5, 7], t[9, 1] = float("inf"), float("nan")
    fin = torch.where(torch.isfinite(t), t, torch.zeros_like(t))
    assert gemm.absmax(t).item() == fin.abs().max().item()
    assert gemm.absmax(torch.zeros(256, 32, device=gpu)).item() == 0.0


@pytest.mark.parametrize("M,N,K", [(8192, 128, 1536), (4096, 256, 6144), (4096, 768, 256)])
def test_f16x2_gemm_has_fp32_accuracy(gpu, M, N, K):
    """The two-piece fp16 form (x 2^s = h + l, products hh + hl + lh, fp32 accumulate) against fp64, beside the fp32 matrix
    instruction on the same operands.  What is asserted is what DESIGN.md 3.2c claims:
      * rms error <= 1.1 x the native fp32 MFMA kernel's on N(0,1) operands, on non-negative (post-activation) operands and
        on gradient-sized ones (x 1e-7: the power-of-two scale from max|x| keeps them out of fp16's subnormal range);
      * every ROW keeps that accuracy as long as its magnitude is within ~2^-11 of the tensor's largest -- rows spanning four
        decades: per-row-normalised error <= 1.5 x native;
      * an over-estimated maximum (the K x max|F| bound the grouped features use: up to 64 x) costs nothing measurable;
      * a row 1e-8 of the maximum is NOT kept to fp32 relative accuracy (absolute error <= max|x| 2^-39 instead): stated, and
        pinned here so that the documentation cannot drift from the kernel."""
    from epn_pointcloud_amd import gemm
    torch.manual_seed(K)
    B = torch.randn(N, K, device=gpu)
    old = gemm.FP32_MODE

    def errs(A, a_amax=None, per_row=False):
        ref = A.double() @ B.double().t()
        scale = ref.pow(2).mean(1, keepdim=True).sqrt() if per_row else ref.pow(2).mean().sqrt()
        out = {}
        try:
            for mode in ("native", "f16x2"):
                gemm.set_fp32_mode(mode)
                C = gemm.gemm_nt(A, B, a_amax=a_amax if mode == "f16x2" else None)
                out[mode] = ((C.double() - ref) / scale).pow(2).mean().sqrt().item()
        finally:
            gemm.set_fp32_mode(old)
        return out

    A = torch.randn(M, K, device=gpu)
    for name, op in (("randn", A), ("post-activation", A.abs()), ("gradient-sized", A * 1e-7)):
        e = errs(op)
        assert e["f16x2"] <= 1.1 * e["native"], (name, e)
    e = errs(A * (10.0 ** torch.linspace(-2, 2, M, device=gpu))[:, None], per_row=True)
    assert e["f16x2"] <= 1.5 * e["native"], ("four decades", e)
    e = errs(A, a_amax=gemm.absmax(A) * 64.0)
    assert e["f16x2"] <= 1.1 * e["native"], ("64 x over-estimated maximum", e)
    tiny = A.clone()
    tiny[0] *= 1e-8
    ref0 = tiny[:1].double() @ B.double().t()
    try:
        gemm.set_fp32_mode("f16x2")
        C0 = gemm.gemm_nt(tiny, B)[:1]
    finally:
        gemm.set_fp32_mode(old)
    rel0 = ((C0.double() - ref0).pow(2).mean().sqrt() / ref0.pow(2).mean().sqrt()).item()
    assert 1e-6 < rel0 < 1e-2, rel0            # a row 1e-8 of the maximum: ~1e-4 relative, far below the tensor's scale in absolute terms
    # the WEIGHT operand is scaled per row (round 5: the wave that splits a weight row takes its maximum): output columns whose
    # weight rows span sixteen decades -- down to 1e-8 of the largest, what a tensor-wide scale loses (above) -- keep the native
    # kernel's per-column relative accuracy, and an all-zero weight row gives exact zeros
    Bw = B * (10.0 ** torch.linspace(-8, 8, N, device=gpu))[:, None]
    Bw[N // 2] = 0.0
    refw = A.double() @ Bw.double().t()
    colscale = refw.pow(2).mean(0, keepdim=True).sqrt().clamp_min(1e-300)
    rw = {}
    try:
        for mode in ("native", "f16x2"):
            gemm.set_fp32_mode(mode)
            Cw = gemm.gemm_nt(A, Bw)
            assert torch.all(Cw[:, N // 2] == 0), mode
            rw[mode] = ((Cw.double() - refw) / colscale).pow(2).mean().sqrt().item()
    finally:
        gemm.set_fp32_mode(old)
    assert rw["f16x2"] <= 1.1 * rw["native"], ("weight rows over sixteen decades", rw)
    # weight-gradient forms: narrow output (both operands split in registers) and wide output (X pre-split into octet planes)
    X = torch.randn(61440, 64, device=gpu)
    for n2 in (128, 1536):
        Y = torch.randn(61440, n2, device=gpu).abs()
        ref = X.double().t() @ Y.double()
        r = {}
        try:
            for mode in ("native", "f16x2"):
                gemm.set_fp32_mode(mode)
                r[mode] = ((gemm.gemm_tn(X, Y).double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        finally:
            gemm.set_fp32_mode(old)
        assert r["f16x2"] <= 1.2 * r["native"] + 1e-7, (n2, r)


def test_split_gemm_ragged_and_fallback(gpu):
    """Shapes the split kernels do not take (K not a multiple of 32, unaligned rows) run on the native fp32 kernels;
    ragged M / N tiles and a strided A are handled by the split kernel itself."""
    from epn_pointcloud_amd import gemm
    torch.manual_seed(5)
    old = gemm.FP32_MODE
    gemm.set_fp32_mode("split")
    try:
        for (M, N, K) in [(1000, 72, 40), (333, 100, 96), (1, 1, 32), (257, 513, 64), (300, 40, 24)]:
            A = torch.randn(M, K + 32, device=gpu)[:, :K] if K % 32 == 0 else torch.randn(M, K, device=gpu)
            B = torch.randn(N, K, device=gpu)
            C = gemm.gemm_nt(A, B)
            ref = A.double() @ B.double().t()
            assert (C.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), (M, N, K)
    finally:
        gemm.set_fp32_mode(old)


def test_split_gemm_c_abi_workspace_contract(gpu):
    """epn_gemm_nt_split_f32 without a workspace (or with one that is too small) runs the native fp32 kernels -- same
    result to rounding; epn_gemm_tn_split_f32 needs its workspace (partial slabs + the planes of X) and says so."""
    import ctypes
    from epn_pointcloud_amd import _lib, gemm
    lib = _lib.get_lib()
    torch.manual_seed(9)
    A, B = torch.randn(700, 96, device=gpu), torch.randn(72, 96, device=gpu)
    ref = A.double() @ B.double().t()
    arr = (_lib.GemmNtProblem * 1)()
    for ws_bytes in (None, 16, "exact"):
        C = torch.zeros(700, 72, device=gpu)
        arr[0] = gemm._problem(A, B, C)
        need = int(lib.epn_gemm_nt_split_workspace_bytes(1, arr))
        assert need >= 6 * 72 * 96
        n = need if ws_bytes == "exact" else (ws_bytes or 0)
        ws = torch.empty(max(n, 1), dtype=torch.uint8, device=gpu)
        rc = lib.epn_gemm_nt_split_f32(1, arr, ws.data_ptr() if ws_bytes else None, n, _lib.stream_of(A))
        assert rc == 0
        assert (C.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    X, Y = torch.randn(4096, 64, device=gpu), torch.randn(4096, 768, device=gpu)
    Cw = torch.empty(64, 768, device=gpu)
    need = int(lib.epn_gemm_tn_workspace_bytes(2, 4096, 64, 768))
    assert need >= 6 * 4096 * 64                                     # at least the bf16 planes of X
    small = torch.empty(256, dtype=torch.uint8, device=gpu)
    rc = lib.epn_gemm_tn_split_f32(X.data_ptr(), 64, Y.data_ptr(), 768, Cw.data_ptr(), 768, 4096, 64, 768, small.data_ptr(),
                                   small.numel(), _lib.stream_of(X))
    assert rc != 0 and b"workspace" in lib.epn_strerror(rc).lower()
    ws = torch.empty(need, dtype=torch.uint8, device=gpu)
    _lib.check(lib.epn_gemm_tn_split_f32(X.data_ptr(), 64, Y.data_ptr(), 768, Cw.data_ptr(), 768, 4096, 64, 768, ws.data_ptr(),
                                         need, _lib.stream_of(X)), "gemm_tn_split")
    ref = X.double().t() @ Y.double()
    assert (Cw.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


# ------------------------------------------------------------------------------------------------ convolutions
@pytest.mark.parametrize("cin,cout,stride,K", [(32, 32, 1, 32), (32, 64, 2, 64), (64, 64, 1, 16), (16, 48, 2, 20)])
def test_inter_bf16_vs_oracle(gpu, vgtk_alias, cin, cout, stride, K):
    sptk, zptk = _mods(vgtk_alias)
    rng = np.random.default_rng(cin + K)
    torch.manual_seed(cin + K)
    xyz = T(unit_ball_cloud(rng, 2, 128))
    conv = sptk.InterSO3Conv(cin, cout, 1, stride, 0.4, 0.08, K, lazy_sample=True)
    conv.basic_conv.W.data = r16(conv.basic_conv.W.data)
    feats = r16(torch.randn(2, cin, 128, 60))
    fo = feats.clone().requires_grad_(True)
    Wo = conv.basic_conv.W.detach().clone().requires_grad_(True)
    o_idx, _, o_sidx, _, oy = R.inter_so3conv(xyz, fo, Wo, conv.anchors, conv.kernels, stride, 0.4, 0.08, K, True)
    gy = r16(torch.randn_like(oy))
    odW, odF = torch.autograd.grad(oy, [Wo, fo], gy)
    conv = conv.to(gpu)
    fg = feats.to(gpu).bfloat16().requires_grad_(True)
    iidx, _, sidx, y = conv(zptk.SphericalPointCloud(xyz.to(gpu), fg, None))
    assert y.feats.dtype == torch.bfloat16
    dW, dF = torch.autograd.grad(y.feats, [conv.basic_conv.W, fg], gy.to(gpu).bfloat16())
    assert torch.equal(iidx.cpu(), o_idx) and torch.equal(sidx.cpu(), o_sidx)      # index work is fp32: bit-exact
    assert dW.dtype == torch.float32 and dF.dtype == torch.bfloat16
    assert rel_max(y.feats.detach(), oy.detach()) < BF16_TOL
    assert rel_l2(dW, odW) < BF16_TOL
    assert rel_l2(dF, odF) < BF16_TOL


@pytest.mark.parametrize("cin,cout,p", [(64, 64, 48), (128, 64, 16), (32, 32, 40), (32, 64, 24)])
def test_intra_bf16_vs_oracle(gpu, vgtk_alias, cin, cout, p):
    """32-multiples take the spectral form (bf16 basis change + bf16 block GEMMs; round 3: also the 32-channel layers of the
    rotation / 3DMatch schedules, whose half-empty 64-channel block is masked in the basis-change kernels)."""
    sptk, zptk = _mods(vgtk_alias)
    torch.manual_seed(cin + p)
    conv = sptk.IntraSO3Conv(cin, cout)
    conv.basic_conv.W.data = r16(conv.basic_conv.W.data)
    feats = r16(torch.randn(2, cin, p, 60))
    fo = feats.clone().requires_grad_(True)
    Wo = conv.basic_conv.W.detach().clone().requires_grad_(True)
    oy = R.intra_so3conv(fo, Wo, conv.intra_idx)
    gy = r16(torch.randn_like(oy))
    odW, odF = torch.autograd.grad(oy, [Wo, fo], gy)
    conv = conv.to(gpu)
    fg = feats.to(gpu).bfloat16().requires_grad_(True)
    y = conv(zptk.SphericalPointCloud(torch.zeros(2, 3, p, device=gpu), fg, None))
    assert y.feats.dtype == torch.bfloat16
    dW, dF = torch.autograd.grad(y.feats, [conv.basic_conv.W, fg], gy.to(gpu).bfloat16())
    assert rel_max(y.feats.detach(), oy.detach()) < BF16_TOL
    assert rel_l2(dW, odW) < BF16_TOL
    assert rel_l2(dF, odF) < BF16_TOL


@pytest.mark.parametrize("instance", [False, True])
def test_norm_act_bf16(gpu, instance):
    from epn_pointcloud_amd import ops
    torch.manual_seed(5)
    x = r16(torch.randn(3, 32, 50, 60) * 2 + 0.5)
    res = r16(torch.randn(3, 32, 50, 60))
    norm = (torch.nn.InstanceNorm2d(32, affine=False) if instance else torch.nn.BatchNorm2d(32)).train()
    xo = x.clone().requires_grad_(True)
    ro = res.clone().requires_grad_(True)
    yo = torch.nn.functional.leaky_relu(norm(xo)) + ro
    gy = r16(torch.randn_like(yo))
    odx, odr = torch.autograd.grad(yo, [xo, ro], gy)
    norm2 = (torch.nn.InstanceNorm2d(32, affine=False) if instance else torch.nn.BatchNorm2d(32)).to(gpu).train()
    xg = x.to(gpu).bfloat16().requires_grad_(True)
    rg = res.to(gpu).bfloat16().requires_grad_(True)
    y = ops.norm_act(xg, norm2, residual=rg)
    assert y.dtype == torch.bfloat16
    dx, dr = torch.autograd.grad(y, [xg, rg], gy.to(gpu).bfloat16())
    assert rel_max(y.detach(), yo.detach()) < 1e-2          # statistics and arithmetic are fp32: one output rounding
    assert rel_l2(dx, odx) < 1e-2 and rel_l2(dr, odr) < 1e-2


def test_separable_block_bf16_tracks_fp32(gpu, vgtk_alias):
    """One FusedSeparableBlock (inter -> norm -> intra -> norm + skip) in bf16 against the same block in fp32."""
    from epn_pointcloud_amd import schedule as S
    sptk, zptk = _mods(vgtk_alias)
    rng = np.random.default_rng(9)
    torch.manual_seed(9)
    layer = S.Layer(64, 64, 1, 0.4, 0.08, 16, True, 1)
    blk = S.FusedSeparableBlock(layer, 60, None).to(gpu).train()
    xyz = T(unit_ball_cloud(rng, 2, 128)).to(gpu)
    feats = torch.randn(2, 64, 128, 60, device=gpu)
    _, _, _, y32 = blk(zptk.SphericalPointCloud(xyz, feats, None))
    S.set_feature_dtype(blk, torch.bfloat16)
    _, _, _, y16 = blk(zptk.SphericalPointCloud(xyz, feats.bfloat16(), None))
    assert y16.feats.dtype == torch.bfloat16
    assert rel_l2(y16.feats, y32.feats.float().cpu()) < 3e-2


# ------------------------------------------------------------------------------------------------ configs 3 and 4
@pytest.mark.parametrize("model,points,batch", [("reg", 1024, 64), ("inv", 2048, 64)])
def test_config_full_size_bf16(gpu, model, points, batch):
    """BASELINE configs 3 (ModelNet40 rotation estimation: 32 pairs = 64 clouds, N=1024) and 4 (3DMatch descriptor:
    64 patches, N=2048) at their stated size in bf16: forward + backward run, everything finite, and the network output
    tracks the fp32 network (same weights, same clouds; whose kernels are oracle-checked in test_gpu_conv.py) -- the
    size-independent property available for a float pipeline.  The index work (FPS, ball query) is fp32 in both."""
    from epn_pointcloud_amd import models as M, schedule as S
    torch.manual_seed(11)
    build = M.build_reg if model == "reg" else M.build_inv
    net = build(points).to(gpu).train()
    scale = 0.4 if model == "inv" else 1.0
    pts = S.synthetic_clouds(batch, points, gpu, seed=77, scale=scale)
    inp = pts.view(batch // 2, 2, points, 3) if model == "reg" else pts

    def run():
        for p in net.parameters():
            p.grad = None
        out = net(inp)
        if model == "reg":
            loss = out[0].float().square().mean() + out[1].float().square().mean()
        else:                                   # descriptors are unit vectors (their squared mean is a constant, and the pairwise
            # products bench.py pushes apart are small differences: a 0.4 % output error reads as 19 % of that loss's gradient):
            # a fixed random linear functional of the descriptors is the well-conditioned probe of the backward pass
            gen = torch.Generator(device="cpu").manual_seed(5)
            probe = torch.randn(out[0].shape, generator=gen).to(out[0].device)
            loss = (out[0].float() * probe).sum()
        loss.backward()
        feats = out[0].detach().float().cpu()
        g = torch.cat([p.grad.flatten().float().cpu() for p in net.parameters() if p.grad is not None])
        return loss.item(), feats, g

    def run_backbone():
        """The backbone's gradients probed BEFORE the head (review item 5, round 5): a fixed random linear functional of the
        feature tensor the head would receive -- the 3DMatch head (softmax over raw attention logits -> max over 64 points -> L2
        normalisation) amplifies a 0.4 % input difference into 13-27 % of gradient, which forced a 40 % bound on the
        through-the-head comparison below; in front of it the backward kernels of all eight blocks are seen at their own
        accuracy.  Returns {parameter name: gradient}."""
        for p in net.parameters():
            p.grad = None
        x = net.features(inp if model != "reg" else torch.cat((inp[:, 0], inp[:, 1]), dim=0))
        gen = torch.Generator(device="cpu").manual_seed(9)
        probe = torch.randn(x.feats.shape, generator=gen).to(x.feats.device)
        (x.feats.float() * probe).sum().backward()
        return {n: p.grad.detach().float().cpu() for n, p in net.backbone.named_parameters() if p.grad is not None}

    l32, f32, g32 = run()
    gb32 = run_backbone()
    S.set_feature_dtype(net, torch.bfloat16)
    l16, f16, g16 = run()
    gb16 = run_backbone()
    # The first block's skip convolution sees the constant occupancy feature: its output is constant per channel, the
    # InstanceNorm behind it subtracts that constant and divides the ROUNDING NOISE that is left by sqrt(eps) -- the weight's
    # exact gradient is zero and what either network holds there is O(1) noise (same exclusion as
    # test_captured_step_reproduces_the_eager_gradients).  Everything else is compared, parameter by parameter.
    noise = {"0.blocks.0.skip_conv.weight", "0.blocks.0.skip_conv.bias"}
    gmax = max(v.abs().max().item() for n, v in gb32.items() if n not in noise)
    rows = []
    for n, v in gb32.items():
        if n in noise or v.abs().max().item() < 1e-3 * gmax:
            continue
        rows.append((rel_l2(gb16[n], v), n, v.abs().max().item()))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print(f"{model}: backbone gradient bf16 vs fp32 (probe in front of the head) {r[1]}: rel-L2 {r[0]:.4f}, scale {r[2]:.3g}")
    keep = [n for _, n, _ in rows]
    dgb = rel_l2(torch.cat([gb16[n].flatten() for n in keep]), torch.cat([gb32[n].flatten() for n in keep]))
    print(f"{model}: backbone gradient bf16 vs fp32, {len(keep)} parameters together: rel-L2 {dgb:.4f}")
    assert all(torch.isfinite(v).all() for v in gb16.values()) and len(keep) >= 20
    # What this measured on its first run (round 6): 0.23 (rotation) / 0.26 (3DMatch) together, 0.26-0.33 for EVERY parameter of
    # the first blocks -- so the 40 % of the through-the-head comparison is not the head's doing.  It is leaky_relu: a
    # pre-activation within one bf16 rounding of zero (a fraction ~ 2^-9 x pdf(0) ~ 1.5e-3 of the elements per layer) takes the
    # other slope in the other network and changes that element's gradient by 99 %; in relative L2 that is sqrt(fraction) ~ 4 % per
    # nonlinearity, ~ 15-30 % after the 14-16 of a backbone.  A network-level bf16-vs-fp32 gradient can therefore only catch a
    # BROKEN layer (100 % and more); the backward kernels themselves are pinned per layer against the oracle on identical
    # inputs (test_gpu_fullsize.py: 8e-3 asserted, 3-4e-3 measured).
    assert dgb < 0.45, dgb
    assert rows[0][0] < 0.6, rows[0]


    assert np.isfinite(l16) and torch.isfinite(f16).all() and torch.isfinite(g16).all()
    dl, df, dg = abs(l16 - l32) / abs(l32), rel_l2(f16, f32), rel_l2(g16, g32)
    print(f"{model}: bf16 vs fp32 network: loss {dl:.4f}, output rel-L2 {df:.4f}, gradient rel-L2 {dg:.4f}")
    # 7-8 blocks deep, each rounding its activations to 8 bits.  Measured (round 4): rotation network loss 1.9e-3, output
    # rel-L2 1.2e-3, all parameter gradients together rel-L2 1.1e-2 (with the head's anchor-pair MLP in fp32,
    # EPN_REG_MLP_BF16=0 EPN_HEAD_BF16=0: 3e-5 / 1.1e-3 / 8.4e-3); 3DMatch output 3.9e-3.  Round 3 asserted 10 % / 15 % --
    # loose enough to hide a wrong layer (review); the bounds below are ~5x what is measured, and the GRADIENT is checked too.
    assert dl <= 0.01
    assert df < 0.02
    # gradients: the rotation network's are well conditioned (8.4e-3 measured).  The 3DMatch head is not: softmax over raw
    # attention logits -> max over 64 points -> L2 normalisation.  Control (tools/scratch/inv_grad_dbg.py): the FP32 network with
    # only the head's INPUT rounded to bf16 (output moves by 2.7e-4) already moves every backbone gradient by 3 % and the
    # head's by 2-5 %; the bf16 backbone moves the head's input 14x more (3.9e-3) and the gradients by 13-27 %, growing
    # smoothly from the head towards the first block.  The per-layer B = 64 slices against the oracle
    # (test_gpu_fullsize.py) are what pins the backward kernels; here the bound only has to catch a broken layer.
    assert dg < (0.05 if model == "reg" else 0.40)


@pytest.mark.parametrize("model,points,batch", [("reg", 1024, 16), ("inv", 2048, 16)])
def test_bf16_network_gradient_without_the_sign_flips(gpu, monkeypatch, model, points, batch):
    """The well-conditioned network-level check of the bf16 backward path (review item 5, round 5): the same networks with
    leaky_relu's negative slope set to 1 (every activation the identity, so no element can take 'the other slope'): convolutions,
    norms, skip branches, basis changes and their transposes remain, and the bf16 network's backbone gradients must then agree
    with the fp32 network's to a few bf16 roundings -- parameter by parameter."""
    from epn_pointcloud_amd import models as M, ops, schedule as S
    # the slope is a default argument of the three glue entry points the blocks call
    monkeypatch.setattr(ops.norm_act, "__defaults__", (None, 1.0, None))
    d = list(ops.norm_act_pair.__defaults__)
    d[1] = 1.0
    monkeypatch.setattr(ops.norm_act_pair, "__defaults__", tuple(d))
    d = list(ops.intra_so3conv_spectral.__defaults__)
    d[1] = 1.0
    monkeypatch.setattr(ops.intra_so3conv_spectral, "__defaults__", tuple(d))
    import inspect
    assert inspect.signature(ops.norm_act).parameters["slope"].default == 1.0
    assert inspect.signature(ops.norm_act_pair).parameters["slope"].default == 1.0
    assert inspect.signature(ops.intra_so3conv_spectral).parameters["pre_slope"].default == 1.0
    torch.manual_seed(11)
    net = (M.build_reg if model == "reg" else M.build_inv)(points).to(gpu).train()
    pts = S.synthetic_clouds(batch, points, gpu, seed=77, scale=0.4 if model == "inv" else 1.0)

    def run_backbone():
        for p in net.parameters():
            p.grad = None
        x = net.features(pts)
        gen = torch.Generator(device="cpu").manual_seed(9)
        probe = torch.randn(x.feats.shape, generator=gen).to(x.feats.device)
        (x.feats.float() * probe).sum().backward()
        return x.feats.detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in net.backbone.named_parameters()
                                                  if p.grad is not None}

    f32, g32 = run_backbone()
    S.set_feature_dtype(net, torch.bfloat16)
    f16, g16 = run_backbone()
    noise = {"0.blocks.0.skip_conv.weight", "0.blocks.0.skip_conv.bias"}     # constant input: exact zero gradient (see above)
    gmax = max(v.abs().max().item() for n, v in g32.items() if n not in noise)
    rows = sorted(((rel_l2(g16[n], v), n) for n, v in g32.items() if n not in noise and v.abs().max().item() >= 1e-3 * gmax),
                  reverse=True)
    for r in rows[:4]:
        print(f"{model} (slope 1): backbone gradient bf16 vs fp32 {r[1]}: rel-L2 {r[0]:.4f}")
    print(f"{model} (slope 1): output rel-L2 {rel_l2(f16, f32):.4f}")
    assert len(rows) >= 20 and rel_l2(f16, f32) < 0.02
    assert rows[0][0] < 0.08, rows[:3]

@pytest.mark.parametrize("b,p1,p2,nn", [(3, 300, 150, 20), (2, 1024, 1024, 32), (2, 4096, 40, 16), (2, 700, 1100, 32)])
def test_inverse_neighbour_list(gpu, b, p1, p2, nn):
    """epn_inter_inverse_list (C ABI): CSR inverse of an index tensor, entries of a destination in increasing (p, n)
    order, out-of-range (shadow) indices dropped.  Clouds whose p2*nn entries fit the LDS take the counting-sort kernel,
    larger ones (last case: 35200 entries) the scanning kernel; both against a numpy construction."""
    from epn_pointcloud_amd import _lib
    rng = np.random.default_rng(p1 + p2)
    idx = rng.integers(0, p1 + 3, size=(b, p2, nn)).astype(np.int32)          # p1 .. p1+2: shadow indices
    idx[0, 0, :] = 5                                                            # one destination named by a whole row
    t = torch.from_numpy(idx).to(gpu)
    off = torch.empty((b, p1 + 1), dtype=torch.int32, device=gpu)
    ent = torch.full((b, p2 * nn), -1, dtype=torch.int32, device=gpu)
    _lib.check(_lib.get_lib().epn_inter_inverse_list(t.data_ptr(), b, p1, p2, nn, off.data_ptr(), ent.data_ptr(),
                                                      _lib.stream_of(t)), "inverse_list")
    off, ent = off.cpu().numpy(), ent.cpu().numpy()
    for bb in range(b):
        flat = idx[bb].reshape(-1)
        valid = np.nonzero(flat < p1)[0]
        order = valid[np.argsort(flat[valid], kind="stable")]                  # by destination, then by entry index
        counts = np.bincount(flat[valid], minlength=p1)
        assert np.array_equal(off[bb], np.concatenate([[0], np.cumsum(counts)]))
        assert np.array_equal(ent[bb, :len(order)], order)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K,n", [(20, 160), (16, 128), (40, 128), (20, 150)])
def test_deterministic_data_gradient(gpu, vgtk_alias, dt, K, n, monkeypatch):
    """EPN_DETERMINISTIC=1: the InterSO3Conv data gradient without atomics.  Where the output points divide into the
    scatter's workgroups (n = 160 / 128: 16 / 8 / 8 points for K = 20 / 16 / 40) the LDS-pre-reduced scatter STORES one
    row per (workgroup, distinct destination), summed in ascending slot order, and the ordered reduction over the inverse
    neighbour list adds the marked rows; otherwise (n = 150: 75 output points) the per-slot slab.  Bitwise repeatable, and
    equal (to rounding) to the atomic-scatter path and to the oracle."""
    sptk, zptk = _mods(vgtk_alias)
    rng = np.random.default_rng(77 + K + n)
    torch.manual_seed(77)
    xyz = T(unit_ball_cloud(rng, 3, n))
    conv = sptk.InterSO3Conv(32, 48, 1, 2, 0.4, 0.08, K, lazy_sample=False)
    conv.basic_conv.W.data = r16(conv.basic_conv.W.data)
    feats = r16(torch.randn(3, 32, n, 60))
    fo = feats.clone().requires_grad_(True)
    _, _, _, _, oy = R.inter_so3conv(xyz, fo, conv.basic_conv.W.detach().clone(), conv.anchors, conv.kernels, 2, 0.4, 0.08,
                                     K, False)
    gy = r16(torch.randn_like(oy))
    (odF,) = torch.autograd.grad(oy, [fo], gy)
    conv = conv.to(gpu)

    def run(det):
        monkeypatch.setenv("EPN_DETERMINISTIC", det)
        fg = feats.to(gpu).to(dt).requires_grad_(True)
        _, _, _, y = conv(zptk.SphericalPointCloud(xyz.to(gpu), fg, None))
        (dF,) = torch.autograd.grad(y.feats, [fg], gy.to(gpu).to(dt))
        return dF

    d1, d2, d0 = run("1"), run("1"), run("0")
    assert torch.equal(d1, d2)                                   # bitwise repeatable
    tol = 1e-3 if dt == torch.float32 else BF16_TOL * odF.abs().max().item()
    assert (d1.float().cpu() - odF).abs().max().item() < tol     # vs the oracle
    assert (d1.float() - d0.float()).abs().max().item() < tol     # vs the atomic scatter
