"""The scale contract of the two-piece fp16 contractions (csrc/gemm.h, DESIGN.md "fp32 contractions"), made loud.

The default fp32 GEMM form scales every operand by the power of two that puts its REPORTED maximum at 2^14; an element above
2-4 x that maximum overflows fp16 and the product is silently non-finite where the fp32 matmul it replaces
(vgtk/vgtk/so3conv/modules.py:48-55) returns a number.  Round 5 shipped exactly that for a day with every parity test green.
This file pins the three guards round 6 added:
  * the epilogue sentinel (epn_f16x2_overflow_count) sees a violated maximum in every two-piece kernel family;
  * a remembered maximum does not survive a raw-pointer write of its tensor (gemm.mark_written);
  * the debug mode (gemm.CHECK_AMAX) re-derives every consumed maximum and a whole training step passes under it;
and that the form TRAINS: 200 Adam steps of the full-width classification network, same seed, two-piece against the exact-f32
matrix instruction."""
import json
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture
def f16x2():
    from epn_pointcloud_amd import gemm
    old = gemm.FP32_MODE
    gemm.set_fp32_mode("f16x2")
    yield gemm
    gemm.set_fp32_mode(old)


@pytest.mark.nonfinite_inputs      # (the violated runs are non-finite on purpose; the counter is asserted inside)
@pytest.mark.parametrize("form", ["nt_256x256", "nt_narrow", "tn_planes", "tn_registers", "tn_grouped", "nt_grouped"])
def test_overflow_sentinel_sees_an_under_reported_maximum(gpu, f16x2, form):
    """A maximum reported 16 x too small: the kernel's result is non-finite AND the device counter says so; the same call with
    the true maximum (or none: the entry point scans) leaves the counter at zero.  One case per kernel family with a two-piece
    epilogue: gemm_nt_x3_kernel (wide and narrow tiles), gemm_tn_x3_kernel (X as octet planes), gemm_tn_f32_kernel<.., 2>
    (both operands split in registers), and the grouped launches of both."""
    gemm = f16x2
    torch.manual_seed(7)

    def run(lie):
        if form.startswith("nt"):
            M, N, K = (4096, 256, 512) if form != "nt_narrow" else (4096, 64, 256)
            A, B = torch.randn(M, K, device=gpu), torch.randn(N, K, device=gpu)
            am = gemm.absmax(A) * lie if lie else None
            if form == "nt_grouped":
                A2 = torch.randn(1024, K, device=gpu)
                outs = gemm.gemm_nt_grouped([(A, B, None), (A2, B, None)], a_amax=[am, None])
                return outs[0], A.double() @ B.double().t()
            return gemm.gemm_nt(A, B, a_amax=am), A.double() @ B.double().t()
        R, N1, N2 = (8192, 256, 1536) if form == "tn_planes" else (8192, 64, 128)
        X, Y = torch.randn(R, N1, device=gpu), torch.randn(R, N2, device=gpu)
        xa = gemm.absmax(X) * lie if lie else None
        if form == "tn_grouped":
            X2, Y2 = torch.randn(4096, 32, device=gpu), torch.randn(4096, 96, device=gpu)
            outs = gemm.gemm_tn_grouped([(X, Y), (X2, Y2)], x_amax=[xa, None] if lie else None, y_amax=None)
            return outs[0], X.double().t() @ Y.double()
        return gemm.gemm_tn(X, Y, x_amax=xa), X.double().t() @ Y.double()

    gemm.f16x2_overflow_count(reset=True)
    C, ref = run(None)
    assert torch.isfinite(C).all() and ((C.double() - ref).norm() / ref.norm()).item() < 1e-5
    assert gemm.f16x2_overflow_count() == 0
    C, ref = run(1.0)
    assert torch.isfinite(C).all() and gemm.f16x2_overflow_count() == 0
    C, _ = run(1.0 / 16.0)
    assert not torch.isfinite(C).all(), "a 16 x under-reported maximum must overflow the fp16 pieces"
    n = gemm.f16x2_overflow_count()
    assert n > 0, "the epilogue sentinel did not see the non-finite accumulators"
    assert gemm.f16x2_overflow_count(reset=True) == n          # sticky until cleared ...
    assert gemm.f16x2_overflow_count() == 0                    # ... and cleared by the reset


def test_other_fp32_forms_do_not_touch_the_counter(gpu):
    from epn_pointcloud_amd import gemm
    old = gemm.FP32_MODE
    A, B = torch.randn(2048, 256, device=gpu), torch.randn(128, 256, device=gpu)
    A[3, 5] = float("inf")
    try:
        for mode in ("split", "native"):
            gemm.set_fp32_mode(mode)
            gemm.f16x2_overflow_count(reset=True)
            gemm.gemm_nt(A, B)
            gemm.gemm_tn(A, A)
            assert gemm.f16x2_overflow_count() == 0, mode
    finally:
        gemm.set_fp32_mode(old)


def test_a_raw_pointer_write_drops_the_remembered_maximum(gpu, f16x2):
    """Maxima are remembered ON tensors, keyed on torch's version counter; the library writes through data_ptr(), which does not
    move that counter.  Every wrapper that hands an EXISTING tensor to a writing kernel calls gemm.mark_written: the tag of the
    tensor, of its base and of every view / detach() alias dies with the write (advisor finding, round 5)."""
    gemm = f16x2
    from epn_pointcloud_amd import ops
    t = torch.randn(512, 64, device=gpu)
    a = gemm.absmax_cached(t)
    assert gemm.amax_tag(t) is a
    v = t.view(64, 512)                          # a whole-buffer view answers with its base's tag ...
    assert gemm.amax_tag(v) is a
    gemm.mark_written(t.detach())                # ... and a write through ANY alias drops it for all of them
    assert gemm.amax_tag(t) is None and gemm.amax_tag(v) is None
    # a GEMM into a caller's tensor
    out = torch.zeros(512, 32, device=gpu)
    ops._tag_amax(out, gemm.absmax(out))
    assert gemm.amax_tag(out) is not None
    gemm.gemm_nt(t, torch.randn(32, 64, device=gpu), out=out)
    assert gemm.amax_tag(out) is None
    acc = torch.zeros(64, 32, device=gpu)
    ops._tag_amax(acc, gemm.absmax(acc))
    gemm.gemm_tn(t, out, out=acc)
    assert gemm.amax_tag(acc) is None
    # the consequence that matters: the stale zero maximum of `out` would have overflowed the next GEMM that reads it
    gemm.f16x2_overflow_count(reset=True)
    C = gemm.gemm_nt(out, torch.randn(16, 32, device=gpu), a_amax=gemm.absmax_cached(out))
    assert torch.isfinite(C).all() and gemm.f16x2_overflow_count() == 0


def test_the_accumulating_scatter_marks_its_target(gpu, vgtk_alias, f16x2, monkeypatch):
    """ops.InterSO3ConvSplitFn.backward may scatter INTO the incoming gradient of the shared block input
    (epn_inter_ungroup_acc_f32): that tensor's version moves, so a maximum remembered on it -- or on any alias of it -- before
    the scatter is not served afterwards."""
    gemm = f16x2
    import numpy as np
    from conftest import unit_ball_cloud
    from oracle import so3conv_ref as R
    from epn_pointcloud_amd import ops
    from epn_pointcloud_amd.vgtk import pc as pctk
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    from epn_pointcloud_amd.vgtk import functional as fr
    monkeypatch.setenv("EPN_SHARE_INPUT_GRAD", "1")
    monkeypatch.setenv("EPN_INTER_BWD_DATA", "split")       # the atomic scatter: the cloud-resident transpose of "auto" writes a fresh tensor
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    b, n, cin, cout, K, radius, sigma = 2, 96, 32, 48, 16, 0.45, 0.09
    xyz = torch.from_numpy(unit_ball_cloud(rng, b, n)).to(gpu)
    anchors = torch.from_numpy(L.get_anchors(60)).to(gpu)
    kernels = R.scaled_kernel_points(torch.from_numpy(fr.kernel_points_raw(24)), radius).to(gpu)
    _, new_xyz = pctk.furthest_sample(xyz, n // 2, False)
    idx = pctk.ball_query_index(new_xyz, xyz, radius, K)
    geo = ops.InterGeometry(xyz, new_xyz, idx, anchors, kernels, sigma)
    f = torch.randn(b, cin, n, 60, device=gpu, requires_grad=True)
    w = (torch.randn(cout, cin * 24, device=gpu) / (cin * 24) ** 0.5).requires_grad_(True)
    side_w = torch.randn(b, cin, n, 60, device=gpu)
    written = []
    real = gemm.mark_written

    def spy(t):
        # a consumer tags the gradient it hands over (as the block tail's backward does) -- the scatter must drop that tag
        ops._tag_amax(t, gemm.absmax(t.contiguous()))
        assert gemm.amax_tag(t) is not None
        r = real(t)
        written.append(gemm.amax_tag(t))
        return r

    monkeypatch.setattr(gemm, "mark_written", spy)
    out, h2, _ = ops.inter_so3conv(f * 1.0, w, geo, share_input=True)
    ((out ** 2).sum() + (h2 * side_w).sum()).backward()
    assert written, "the accumulating scatter did not mark the tensor it wrote into"
    assert all(t is None for t in written)


@pytest.mark.nonfinite_inputs
def test_debug_mode_rederives_every_consumed_maximum(gpu, f16x2, monkeypatch):
    """gemm.CHECK_AMAX (EPN_AB=1 EPN_CHECK_AMAX=1): an under-reported maximum raises, naming the call; a whole eager training step
    of the full-width classification network (4 clouds: grouping bounds, basis-change and block-tail tags, norm-backward tags,
    scans of untagged gradients) passes -- every maximum the step consumes bounds its operand."""
    gemm = f16x2
    from epn_pointcloud_amd import models as M, schedule as S
    monkeypatch.setattr(gemm, "CHECK_AMAX", True)
    A, B = torch.randn(1024, 128, device=gpu), torch.randn(64, 128, device=gpu)
    with pytest.raises(AssertionError, match="reported max"):
        gemm.gemm_nt(A, B, a_amax=gemm.absmax(A) * 0.5)
    with pytest.raises(AssertionError, match="operand Y"):
        gemm.gemm_tn(A, A, x_amax=gemm.absmax(A), y_amax=gemm.absmax(A) * 0.25)
    gemm.gemm_nt(A, B, a_amax=gemm.absmax(A) * 32.0)            # a bound may over-estimate
    gemm.f16x2_overflow_count(reset=True)
    checked = []
    real = gemm._check_amax
    monkeypatch.setattr(gemm, "_check_amax", lambda t, a, what: (checked.append(what) if a is not None else None, real(t, a, what))[1])
    torch.manual_seed(3)
    m = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(gpu).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    pts = S.synthetic_clouds(4, 1024, gpu, seed=5)
    labels = torch.tensor([1, 2, 3, 4], device=gpu)
    for _ in range(2):                                           # second step: weights moved, BatchNorm statistics moved
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(pts)[0], labels)
        loss.backward()
        opt.step()
    assert torch.isfinite(loss)
    assert len(checked) >= 100, len(checked)                     # ~90 two-piece GEMM operands with a supplied maximum per step
    assert gemm.f16x2_overflow_count() == 0


def _train_trace(gpu, mode, steps, batch):
    from epn_pointcloud_amd import gemm, models as M, schedule as S
    old = gemm.FP32_MODE
    gemm.set_fp32_mode(mode)
    try:
        torch.manual_seed(2913)
        m = M.ClsSO3ConvModel(S.cls_so3net_schedule(1024), out_mlps=(256,), pooling="attention").to(gpu).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        pts = S.synthetic_clouds(batch, 1024, gpu, seed=2913)
        labels = torch.arange(batch, device=gpu) % 40
        losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(m(pts)[0], labels)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        return torch.stack(losses).cpu().tolist()
    finally:
        gemm.set_fp32_mode(old)


def test_two_piece_form_trains_like_the_fp32_matrix_instruction(gpu):
    """200 Adam steps (lr 1e-3) of the full-width classification network (SPConvNets/models/cls_so3net_pn.py:15-40) on 8 fixed
    synthetic clouds, same seed, same data: two-piece fp16 contractions against the exact-f32 MFMA kernels ('native' = plain
    fp32 matmul semantics, vgtk/vgtk/so3conv/modules.py:48-55).  Every loss of both runs is finite, the overflow sentinel stays
    at zero over all 200 steps (weights, activations and gradients move by orders of magnitude while the network memorises its
    batch), and the trajectories agree: within 2 % of the starting loss at step 200 and on average.  (Bit-level agreement is not
    expected: the fp32 atomics of the scatter make even two 'native' runs differ in the last bits, and training amplifies that.)
    The traces are written to gpurun_out/ (committed copies: profiles/r06_loss_trace_*.json)."""
    from epn_pointcloud_amd import gemm
    steps, batch = 200, 8
    gemm.f16x2_overflow_count(reset=True)
    f2 = _train_trace(gpu, "f16x2", steps, batch)
    assert gemm.f16x2_overflow_count() == 0
    nat = _train_trace(gpu, "native", steps, batch)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    for name, tr in (("f16x2", f2), ("native", nat)):
        with open(os.path.join(out, f"loss_trace_cls_b8_{name}.json"), "w") as f:
            json.dump({"model": "cls_so3net_pn full width", "batch": batch, "steps": steps, "optimizer": "Adam lr 1e-3",
                       "fp32_gemm_mode": name, "seed": 2913, "loss": tr}, f)
    assert all(map(lambda v: v == v and abs(v) < 1e30, f2)) and all(map(lambda v: v == v and abs(v) < 1e30, nat))
    l0 = nat[0]
    assert abs(f2[0] - nat[0]) < 1e-4 * l0, (f2[0], nat[0])                       # same network, same data: the first loss agrees
    assert nat[-1] < 0.5 * l0 and f2[-1] < 0.5 * l0, (l0, nat[-1], f2[-1])        # both learn
    assert abs(f2[-1] - nat[-1]) < 0.02 * l0, (f2[-1], nat[-1])                   # within 2 % (of the starting loss) at step 200
    mean_gap = sum(abs(a - b) for a, b in zip(f2, nat)) / steps
    assert mean_gap < 0.02 * l0, mean_gap
    tail = lambda tr: sum(tr[-20:]) / 20.0
    assert abs(tail(f2) - tail(nat)) < 0.02 * l0


@pytest.mark.nonfinite_inputs
def test_basis_change_maximum_leaves_an_infinite_output_out(gpu, f16x2):
    """One inf in the input of the anchor basis change poisons the 60 spectral rows of its (point, channel) -- and nothing else:
    the producer-side maximum of the spectral buffer is the maximum of its FINITE elements, as epn_absmax_f32 and the weight
    split define it.  (Round 5 clamped it to FLT_MAX: scale 2^-113, every finite element of the buffer flushed to zero in the
    consuming GEMMs -- advisor finding.)"""
    gemm = f16x2
    from epn_pointcloud_amd import ops
    from test_models_cpu import tables
    _, _, intra_idx = tables()
    basis = ops.spectral_basis(intra_idx.int().to(gpu))
    torch.manual_seed(4)
    for c in (64, 32):
        f = ops.to_cl(torch.randn(2, c, 40, 60, device=gpu))
        f[1, 5, 7, 11] = float("inf")
        y = ops.ToSpectralFn.apply(f, basis)
        tag = gemm.amax_tag(y)
        assert tag is not None
        fin = torch.where(torch.isfinite(y), y, torch.zeros_like(y))
        bad = (~torch.isfinite(y)).sum().item()
        assert 0 < bad <= 60, bad
        assert tag.item() == fin.abs().max().item(), (c, tag.item(), fin.abs().max().item())
        assert 1.0 < tag.item() < 1e3
