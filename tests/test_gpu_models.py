"""GPU parity of the model tails and of whole networks, through the C ABI:
PointnetSO3Conv (one fused HIP pass) against reference goldens and the oracle on awkward shapes; the three shipped
networks (tiny widths, weights filled from their state_dict keys) against outputs of the unmodified reference builders
(tests/golden/gen_golden_models.py), with the HIP block glue and with stock torch glue."""
import numpy as np
import os

import pytest
import torch

from conftest import golden, unit_ball_cloud
from test_models_cpu import TOL, fill_state_dict, filled_oracle, product_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def close(got, want, tol=TOL):
    """max |got - want| <= 1e-3 ABSOLUTE (north_star: "features within 1e-3 fp32"; logits reach +-6)."""
    want = want if torch.is_tensor(want) else T(want)
    return (got.detach().cpu() - want).abs().max().item() <= tol


def grad_close(got, want, rel=3e-2):
    """Network-level gradients cross discrete routing decisions -- the arg-max point of PointnetSO3Conv, the sign of
    every leaky_relu input -- and a 1e-4 feature difference flips a few near-ties, each moving the gradient by a finite
    amount.  (Per-operator gradients are held to 1e-3 in the operator tests, where both sides see identical inputs.)
    Here: relative L2 error below 3 %."""
    want = want if torch.is_tensor(want) else T(want)
    got = got.detach().cpu().reshape(want.shape)
    return ((got - want).norm() / want.norm().clamp_min(1e-12)).item() < rel


@pytest.mark.parametrize("tag", ["a60", "a1"])
def test_pointnet_vs_reference_golden(gpu, tag):
    from epn_pointcloud_amd.vgtk import so3conv as sptk, spconv as zptk
    g = golden(f"pointnet_{tag}.npz")
    f = T(g["feats"]).to(gpu).requires_grad_(True)
    m = fill_state_dict(sptk.PointnetSO3Conv(f.shape[1], g["out"].shape[1], 60)).to(gpu)
    y = m(zptk.SphericalPointCloud(T(g["xyz"]).to(gpu), f, None))
    assert tuple(y.shape) == g["out"].shape
    assert (y.detach().cpu() - T(g["out"])).abs().max().item() < TOL
    dF, dW, dB = torch.autograd.grad(y, [f, m.embed.weight, m.embed.bias], T(g["gy"]).to(gpu))
    assert (dF.cpu() - T(g["dF"])).abs().max().item() < TOL
    assert (dW.cpu() - T(g["dW"])).abs().max().item() < TOL
    assert (dB.cpu() - T(g["dB"])).abs().max().item() < TOL


@pytest.mark.parametrize("form", ["gemm", "fused"])
@pytest.mark.parametrize("b,c,co,p,na", [(2, 128, 128, 64, 60), (1, 70, 200, 33, 60), (3, 16, 8, 129, 1),
                                         (2, 256, 300, 40, 12), (2, 256, 304, 40, 12), (4, 64, 40, 70, 60)])
def test_pointnet_vs_oracle(gpu, monkeypatch, form, b, c, co, p, na):
    """Tile edges: channels not a multiple of the 64-channel tile, more than 128 outputs, ragged point counts.  Both forms:
    the GEMM-composed one (c % 16 == 0 and co % 8 == 0; the others fall back) and the fused fp32 kernels."""
    from oracle import so3conv_ref as R
    from epn_pointcloud_amd import ops
    monkeypatch.setenv("EPN_POINTNET", form)
    rng = np.random.default_rng(b * 1000 + c)
    torch.manual_seed(c)
    xyz = T(unit_ball_cloud(rng, b, p))
    f = torch.randn(b, c, p, na)
    anchors = torch.linalg.qr(torch.randn(na, 3, 3))[0].contiguous()
    w, bias, gy = torch.randn(co, c + 3, 1, 1) / (c ** 0.5), torch.randn(co), torch.randn(b, co, na)
    fr, wr, br = f.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = R.pointnet_so3conv(xyz, fr, anchors if na > 1 else None, wr, br)
    ref = torch.autograd.grad(yr, [fr, wr, br], gy)
    fg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (f, w, bias))
    y = ops.pointnet_so3conv(fg, xyz.to(gpu), anchors.to(gpu), wg, bg)
    got = torch.autograd.grad(y, [fg, wg, bg], gy.to(gpu))
    assert (y.detach().cpu() - yr.detach()).abs().max().item() < TOL
    for u, v, n in zip(got, ref, ("dF", "dW", "dB")):
        assert (u.cpu() - v).abs().max().item() < TOL * max(1.0, v.abs().max().item()), n


@pytest.mark.nonfinite_inputs          # NaN features go through the two-piece GEMM of the composed form on purpose
@pytest.mark.parametrize("form", ["gemm", "fused"])
def test_pointnet_max_propagates_nan(gpu, monkeypatch, form):
    """The max over points is torch.max's (so3conv/modules.py:230): a NaN activation IS the maximum of its (cloud, anchor,
    channel) column -- a diverged run must show in the head's output, not be filtered out (advisor finding, round 4) -- and the
    other columns are unaffected."""
    from epn_pointcloud_amd import ops
    monkeypatch.setenv("EPN_POINTNET", form)
    b, c, co, p, na = 2, 128, 128, 70, 60
    rng = np.random.default_rng(3)
    torch.manual_seed(3)
    xyz = T(unit_ball_cloud(rng, b, p)).to(gpu)
    f = torch.randn(b, c, p, na, device=gpu)
    anchors = torch.linalg.qr(torch.randn(na, 3, 3))[0].contiguous().to(gpu)
    w, bias = torch.randn(co, c + 3, 1, 1, device=gpu) / c ** 0.5, torch.randn(co, device=gpu)
    clean = ops.pointnet_so3conv(f, xyz, anchors, w, bias)
    f2 = f.clone()
    f2[1, :, 37, 5] = float("nan")                    # every channel of one (cloud, point, anchor) row
    y = ops.pointnet_so3conv(f2, xyz, anchors, w, bias)
    assert torch.isnan(y[1, :, 5]).all()
    mask = torch.ones_like(y, dtype=torch.bool)
    mask[1, :, 5] = False
    assert torch.equal(y[mask], clean[mask])


def test_pointnet_bf16_features(gpu):
    """bf16 features go through the GEMM-composed form in bf16 (bf16 matrix pipe, fp32 accumulate, bf16 dZ / dF): against
    the fp32 oracle on the bf16-rounded inputs, within bf16 rounding of the operands; the arg-max may move between
    near-ties, so the gradients are compared in rel-L2."""
    from oracle import so3conv_ref as R
    from epn_pointcloud_amd import ops
    b, c, co, p, na = 2, 128, 128, 64, 60
    rng = np.random.default_rng(11)
    torch.manual_seed(11)
    xyz = T(unit_ball_cloud(rng, b, p))
    f = torch.randn(b, c, p, na).bfloat16().float()
    anchors = torch.linalg.qr(torch.randn(na, 3, 3))[0].contiguous()
    w = (torch.randn(co, c + 3, 1, 1) / (c ** 0.5))
    w[:, :c] = w[:, :c].bfloat16().float()
    bias, gy = torch.randn(co), torch.randn(b, co, na)
    fr, wr, br = f.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = R.pointnet_so3conv(xyz, fr, anchors, wr, br)
    ref = torch.autograd.grad(yr, [fr, wr, br], gy)
    fg = f.to(gpu).bfloat16().requires_grad_(True)
    wg, bg = (t.to(gpu).requires_grad_(True) for t in (w, bias))
    y = ops.pointnet_so3conv(fg, xyz.to(gpu), anchors.to(gpu), wg, bg)
    got = torch.autograd.grad(y, [fg, wg, bg], gy.to(gpu))
    assert y.dtype == torch.float32 and got[0].dtype == torch.bfloat16 and got[1].dtype == torch.float32
    assert (y.detach().cpu() - yr.detach()).abs().max().item() < 1e-4 * max(1.0, yr.abs().max().item())
    for u, v, n, tol in zip(got, ref, ("dF", "dW", "dB"), (1e-2, 1e-2, 1e-5)):
        assert ((u.float().cpu() - v).norm() / v.norm()).item() < tol, n


def test_pointnet_rejects_bad_arguments(gpu):
    from epn_pointcloud_amd import ops
    f = torch.randn(1, 8, 5, 4, device=gpu)
    with pytest.raises(ValueError):
        ops.pointnet_so3conv(f, torch.randn(1, 3, 6, device=gpu), None, torch.randn(4, 11, 1, 1, device=gpu), None)
    with pytest.raises(ValueError):
        ops.pointnet_so3conv(f, torch.randn(1, 3, 5, device=gpu), None, torch.randn(4, 8, 1, 1, device=gpu), None)
    with pytest.raises(RuntimeError):
        ops.pointnet_so3conv(f.cpu(), torch.randn(1, 3, 5), None, torch.randn(4, 11, 1, 1), None)
    # the same checks in front of the GEMM-composed form (c % 16 == 0, co % 8 == 0), and its C entries' own
    monkey = pytest.MonkeyPatch()
    monkey.setenv("EPN_POINTNET", "gemm")
    try:
        f16 = torch.randn(1, 16, 5, 4, device=gpu)
        with pytest.raises(ValueError):
            ops.pointnet_so3conv(f16, torch.randn(1, 3, 6, device=gpu), None, torch.randn(8, 19, 1, 1, device=gpu), None)
        with pytest.raises(ValueError):
            ops.pointnet_so3conv(f16, torch.randn(1, 3, 5, device=gpu), None, torch.randn(8, 16, 1, 1, device=gpu), None)
    finally:
        monkey.undo()
    import ctypes
    from epn_pointcloud_amd import _lib
    lib = _lib.get_lib()
    g = torch.zeros(1, 4, 12, device=gpu)
    a = torch.zeros(1, 4, 12, dtype=torch.int32, device=gpu)
    dz = torch.zeros(5 * 4 * 12, device=gpu)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.epn_pointnet_dz_f32(vp(g), vp(a), vp(dz), 1, 5, 4, 12, None) == -1     # EPN_EINVAL: co % 8 != 0
    assert lib.epn_pointnet_dz_f32(vp(g), None, vp(dz), 1, 5, 4, 8, None) == -3
    assert lib.epn_pointnet_max_f32(None, vp(g), None, vp(g), None, vp(g), vp(a), vp(g), 1, 5, 4, 16, 8, None) == -3
    assert lib.epn_pointnet_bwd_coord_f32(vp(g), vp(a), vp(g), None, vp(g), None, None, 1, 5, 4, 16, 8, None) == -3


def _set_glue(model, fused):
    from epn_pointcloud_amd import schedule as S
    for m in model.modules():
        if isinstance(m, S.SeparableBlock):
            m.__class__ = S.FusedSeparableBlock if fused else S.SeparableBlock
    return model


@pytest.mark.parametrize("fused", [True, False])
def test_cls_model_vs_reference_golden(gpu, fused):
    g = golden("model_cls_tiny.npz")
    m = _set_glue(fill_state_dict(product_model("cls")), fused).to(gpu).train()
    logits, att = m(T(g["pts"]).to(gpu))
    assert close(logits, g["logits"])
    assert close(att, g["attention"])
    loss = torch.nn.functional.cross_entropy(logits, T(g["labels"]).to(gpu))
    assert abs(loss.item() - float(g["loss"])) < TOL
    pd = dict(m.named_parameters())
    names = g["grad_names"].tolist()
    grads = torch.autograd.grad(loss, [pd[n] for n in names])
    for i, (n, gr) in enumerate(zip(names, grads)):
        assert grad_close(gr, g[f"grad{i}"]), n
    # eval mode (running statistics): stock modules around the HIP convolutions
    m = fill_state_dict(m.cpu()).to(gpu).eval()
    with torch.no_grad():
        logits_eval, _ = m(T(g["pts"]).to(gpu))
    assert close(logits_eval, g["logits_eval"])


@pytest.mark.parametrize("fused", [True, False])
def test_reg_model_vs_reference_golden(gpu, fused):
    g = golden("model_reg_tiny.npz")
    m = _set_glue(fill_state_dict(product_model("reg")), fused).to(gpu).train()
    conf, quats = m(T(g["pairs"]).to(gpu))
    assert close(conf, g["confidence"], 3e-3)      # softmax(3 * attention) of features of scale 10: 3x the feature tol
    assert close(quats, g["quats"])


@pytest.mark.parametrize("fused", [True, False])
def test_inv_model_vs_reference_golden(gpu, fused):
    g = golden("model_inv_tiny.npz")
    m = _set_glue(fill_state_dict(product_model("inv")), fused).to(gpu).train()
    desc, attn = m(T(g["pts"]).to(gpu))
    assert close(desc, g["descriptor"])
    assert close(attn[:, :, ::8], g["attention_sub"])
    (desc @ desc.t()).square().sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


_ORACLE_STEP = {}


@pytest.mark.parametrize("inter_mode", ["auto", "onchip", "auto+bwd_data_onchip", "auto+bwd_data_split"])
def test_full_width_cls_step_matches_oracle_loss(gpu, monkeypatch, inter_mode):
    """Full-width classification network on 2 clouds: logits, loss and EVERY parameter gradient against the CPU oracle
    (SPConvNets/models/cls_so3net_pn.py:15-40 restated by oracle/backbone_ref.py) -- in the default split form and with
    EPN_INTER_MODE=onchip, the COMPLETE path of north_star's fused form: every InterSO3Conv of the step keeps its grouped
    features on chip in both directions (forward csrc/inter_fx.hip, data and weight gradients the fused transposes of
    csrc/inter_mfma.hip; no [cols, cin*ks] tensor is ever allocated -- ops.InterSO3ConvOnChipFn).
    Round 5 compared two of the ~100 gradient tensors the oracle computes (review); now all of them, except those whose exact
    gradient is ZERO (biases a normalisation cancels, the first block's skip convolution over a constant occupancy feature:
    what the oracle holds there is its own rounding noise, < 1e-3 of the largest gradient of the network)."""
    from epn_pointcloud_amd import models as M, schedule as S
    from oracle import backbone_ref as B
    from test_models_cpu import tables
    bwd_onchip = inter_mode == "auto+bwd_data_onchip"       # the split form with dG kept on chip (csrc/inter_bwd_f2.hip, round 6)
    if bwd_onchip:
        inter_mode = "auto"
        monkeypatch.setenv("EPN_INTER_BWD_DATA", "onchip")
    if inter_mode == "auto+bwd_data_split":                 # ... with the LDS-pre-reduced atomic scatter in every layer (auto: the cloud-resident fixed-point transpose)
        inter_mode = "auto"
        monkeypatch.setenv("EPN_INTER_BWD_DATA", "split")
    monkeypatch.setenv("EPN_INTER_MODE", inter_mode)
    layers = S.cls_so3net_schedule(1024)
    torch.manual_seed(5)
    m = M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention").train()
    pts = S.synthetic_clouds(2, 1024, "cpu", seed=11)
    labels = torch.tensor([7, 31])
    if "r" not in _ORACLE_STEP:                 # same seed, same weights for both forms: the oracle runs once
        ref = B.RefClsModel(layers, tables(), out_mlps=(256,), pooling="attention").train()
        ref.load_from_product(m.state_dict())
        lr, _ = ref(pts)
        loss_r = torch.nn.functional.cross_entropy(lr, labels)
        rp = {n: p for n, p in ref.named_parameters() if p.requires_grad}
        gr = torch.autograd.grad(loss_r, list(rp.values()), allow_unused=True)
        _ORACLE_STEP["r"] = (lr.detach(), loss_r.detach(), {n: g.detach() for n, g in zip(rp, gr) if g is not None})
    lr, loss_r, gr = _ORACLE_STEP["r"]
    m = m.to(gpu)
    if inter_mode == "onchip":                  # the form really is taken by the layers it serves (cin >= 16)
        from epn_pointcloud_amd import ops
        taken = []
        real = ops.InterSO3ConvOnChipFn.forward
        monkeypatch.setattr(ops.InterSO3ConvOnChipFn, "forward", staticmethod(lambda ctx, *a: (taken.append(1), real(ctx, *a))[1]))
    if bwd_onchip:
        from epn_pointcloud_amd import ops
        kinds = []
        real_launch = ops._launch
        monkeypatch.setattr(ops, "_launch", lambda kind, *a: (kinds.append(kind), real_launch(kind, *a))[1])
    lg, _ = m(pts.to(gpu))
    loss_g = torch.nn.functional.cross_entropy(lg, labels.to(gpu))
    pd = {n: p for n, p in m.named_parameters() if p.requires_grad}
    assert set(gr) <= set(pd), sorted(set(gr) - set(pd))          # the oracle's module tree is the product's, key for key
    names = sorted(gr)
    gg = dict(zip(names, torch.autograd.grad(loss_g, [pd[n] for n in names], allow_unused=True)))
    assert (lg.detach().cpu() - lr.detach()).abs().max().item() < TOL * max(1.0, lr.abs().max().item())
    assert abs(loss_g.item() - loss_r.item()) < TOL
    gmax = max(g.abs().max().item() for g in gr.values())
    # ONE more family is decided by rounding noise on either side (found by this test's first run, 0.85 relative): the first
    # block's skip branch.  Its 1x1 convolution sees the constant occupancy feature, so its output is one constant per channel;
    # BatchNorm subtracts the batch mean -- the SAME constant up to the rounding of a 983 040-term mean -- and at initialisation
    # beta = 0, so the sign of that rounding residue alone picks leaky_relu's slope (1 or 0.01) for a whole channel, and
    # d loss / d beta_c = slope_c * sum(dy) differs by a factor 100 per channel between any two summation orders (the oracle's
    # included).  gamma's gradient there multiplies the same residue; the convolution's weight is the exact zero already skipped.
    noise_decided = {"backbone.0.blocks.0.norm.bias", "backbone.0.blocks.0.norm.weight"}
    bad, checked = [], 0
    for n in names:
        v = gr[n]
        if n in noise_decided:
            continue
        if v.abs().max().item() < 1e-3 * gmax:                     # exact gradient zero: nothing but rounding on either side
            assert gg[n] is None or gg[n].abs().max().item() < 2e-3 * gmax, n
            continue
        checked += 1
        if gg[n] is None or not grad_close(gg[n], v):
            bad.append((n, None if gg[n] is None else ((gg[n].detach().cpu().reshape(v.shape) - v).norm() / v.norm()).item()))
    assert not bad, bad
    assert checked >= 40, checked                                  # 7 blocks x (2 conv W + skip W + 2 norm affine pairs) + head
    for n in ("outblock.fc2.weight", "backbone.3.blocks.0.inter_conv.conv.basic_conv.W", "backbone.0.blocks.1.intra_conv.conv.basic_conv.W"):
        assert n in gr and gr[n].abs().max().item() >= 1e-3 * gmax, n   # (the named ones of round 5 are among the checked)
    if inter_mode == "onchip":
        assert len(taken) >= 6, taken
    if bwd_onchip:                              # every InterSO3Conv with cin >= 16 took it; no dG GEMM, no separate transpose
        assert kinds.count("inter_bwd_data_f2") >= 6 and "inter_gemm_dg" not in kinds and "inter_ungroup" not in kinds, kinds


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_captured_step_reproduces_the_eager_gradients(gpu, dt):
    """bench.py times a HIP-graph replay of forward + loss + backward.  The replayed graph must compute what the eager step
    computes: loss and every parameter gradient of the full-width classification network (2 clouds), eager vs replay of a
    capture -- with the skip branch on its second stream, the shared-input gradient accumulated in place by the scatter,
    the epilogue statistics and the packed grouping all inside the capture.  Tolerance: the fp32 atomics of the scatter
    land in a different order from run to run (1e-4 of each gradient's scale; bf16 features: 2e-2)."""
    from epn_pointcloud_amd import models as M, schedule as S
    torch.manual_seed(3)
    layers = S.cls_so3net_schedule(1024)
    m = M.ClsSO3ConvModel(layers, out_mlps=(256,), pooling="attention").to(gpu).train()
    if dt == "bf16":
        S.set_feature_dtype(m, torch.bfloat16)
    pts = S.synthetic_clouds(2, 1024, gpu, seed=4)
    labels = torch.tensor([3, 17], device=gpu)
    params = [p for p in m.parameters() if p.requires_grad]

    def compute():
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(m(pts)[0], labels)
        loss.backward()
        return loss

    # BatchNorm running statistics advance on every step: restored so that eager and replay start from the same state
    # (only those: the index tables are cache keys of the library's derived tables -- rewriting them would rebuild the tables
    # on the host in the middle of the capture)
    stats = {k: v for k, v in m.named_buffers() if "running_" in k or "num_batches" in k}
    saved = {k: v.clone() for k, v in stats.items()}

    def restore():
        with torch.no_grad():
            for k, v in stats.items():
                v.copy_(saved[k])
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):                        # eager steps off the default stream, as torch asks before a capture
        compute()
        restore()
        loss_e = compute().detach().clone()
        ge = [p.grad.detach().clone() for p in params]
        restore()
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize(gpu)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        loss_c = compute()
    graph.replay()
    torch.cuda.synchronize(gpu)
    tol = 2e-2 if dt == "bf16" else 1e-4
    assert abs(loss_c.item() - loss_e.item()) <= tol * max(1.0, abs(loss_e.item()))
    # Parameters whose exact gradient is ZERO are left out: the first block's skip convolution (a constant occupancy feature
    # times a weight, then BatchNorm: the output does not depend on the weight) and the biases a normalisation cancels -- what
    # they hold is the atomics' rounding noise (|g| < 1e-3 of the largest gradient in the network), different on every run.
    gmax = max(g.abs().max().item() for g in ge)
    bad, checked = [], 0
    for (n, p), g in zip(((n, p) for n, p in m.named_parameters() if p.requires_grad), ge):
        scale = g.abs().max().item()
        # (bf16 features: the rounded products of that first skip convolution are no longer exactly constant, and the BatchNorm
        # behind it divides their noise by sqrt(eps): an O(1) gradient made of nothing but rounding, left out by name)
        if scale < 1e-3 * gmax or n == "backbone.0.blocks.0.skip_conv.weight":
            continue
        checked += 1
        err = (p.grad - g).abs().max().item()
        if err > tol * scale:
            bad.append((n, err, scale))
    assert not bad, bad
    assert checked > 40


def test_cls_model_kanchor20_vs_reference_golden(gpu):
    """Reduced-anchor configuration: every block is an InterSO3ConvBlock (no intra conv, no skip); outputs and two
    gradients against the reference's own build of that network."""
    from epn_pointcloud_amd import models as M
    from test_models_cpu import tiny_layers
    g = golden("model_cls_k20_tiny.npz")
    m = fill_state_dict(M.ClsSO3ConvModel(tiny_layers("cls"), out_mlps=(32,), pooling="attention", kanchor=20)).to(gpu).train()
    logits, att = m(T(g["pts"]).to(gpu))
    loss = torch.nn.functional.cross_entropy(logits, T(g["labels"]).to(gpu))
    assert close(logits, g["logits"]) and close(att, g["attention"])
    assert abs(loss.item() - float(g["loss"])) < TOL
    pd = dict(m.named_parameters())
    grads = torch.autograd.grad(loss, [pd[n] for n in g["grad_names"].tolist()])
    for i, gr in enumerate(grads):
        assert grad_close(gr, g[f"grad{i}"])


def _latest_kernel_stats():
    import glob
    from conftest import ROOT
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0?_kernel_stats.csv")))
    return found[-1] if found else None


def test_bench_line_kernel_names_are_profiler_names(gpu, tmp_path):
    """bench.py's per-kernel table (the detail file the one-line report names) is keyed by what the LIBRARY reports for each
    call (epn_last_kernel) -- the exact template instances -- so every name must be a kernel name of the committed rocprofv3
    trace of the same command (profiles/r0N_kernel_stats.csv, the latest round), compared after bench.norm_kernel_name (no
    'void ', no '(anonymous namespace)::', no argument list).  The table covers glue / index / cast kernels, not only the
    convolutions.  And the stdout line itself stays under the driver's capture window: < 3000 bytes, one JSON object with
    `roofline` (round 3's 35 KB line could not be parsed)."""
    import csv
    import json
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    stats = _latest_kernel_stats()
    if stats is None:
        pytest.skip("profiles/r0N_kernel_stats.csv not collected yet")
    known = {bench.norm_kernel_name(r["Name"]) for r in csv.DictReader(open(stats))}
    detail_file = str(tmp_path / "bench_detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-native-line", "--no-extra-configs"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, EPN_BENCH_DETAIL=detail_file))
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert len(last) < 3000
    line = json.loads(last)
    assert line["detail"] == detail_file and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert len(json.dumps(line["roofline"])) < 900
    detail = json.load(open(detail_file))["detail"]["headline"]
    assert not any(k.startswith("_Z") for k in detail["per_kernel"]), "per_kernel keys are readable instance names"
    names = {v["kernel_exact"] for v in detail["per_kernel"].values()}
    assert detail["dominant_kernel_exact"] in known
    assert bench.short_kernel(detail["dominant_kernel_exact"]) == line["roofline"]["kernel"]
    missing = sorted(n for n in names if n not in known and not n.startswith("(host)"))
    # kernels added after the committed trace was collected are reported, not failed: the trace is re-collected per round
    frac_known = 1.0 - len(missing) / max(1, len(names))
    assert frac_known > 0.8, missing
    assert any("norm_act" in n for n in names) and any("fps" in n for n in names)      # glue and index kernels are bracketed


def test_default_bench_run_prints_one_small_parsable_line(gpu, tmp_path):
    """The DEFAULT command form the driver runs (`bench.py --steps K --warmup W`, every embedded config, native line, CPU
    baseline on a one-cloud sample to keep the test short): the last stdout line parses, is < 3000 bytes and carries
    `roofline`, `cpu_baseline` and the three embedded configs with their priced dominant kernels."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    detail_file = str(tmp_path / "bench_detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--cpu-clouds", "1",
                          "--cpu-samples", "1"], capture_output=True, text=True, timeout=1500,
                         env=dict(os.environ, EPN_BENCH_DETAIL=detail_file))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 3000, (len(lines), len(lines[-1]))
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "config", "roofline",
              "cpu_baseline", "native_fp32_mfma", "configs"):
        assert k in line, k
    assert set(line["configs"]) == {"cls_fwd", "reg_bf16", "inv_bf16", "cls_dp_rank"}
    for name, c in line["configs"].items():
        assert "error" not in c, (name, c)
        if name != "cls_dp_rank":
            assert c["value"] > 0 and c["bound"] in ("hbm", "mfma") and 0 < c["frac"] < 1.5 and c["kernel"]
    # the program one rank of a multi-GPU job runs (flat gradient buffer + all-reduce on a 1-rank RCCL communicator inside the
    # timed region) costs at most 3 % of the single-GPU step (review item 1; measured 0.x ms)
    dpr = line["configs"]["cls_dp_rank"]
    assert dpr["vs_headline"] >= 0.97 and dpr["overhead_ms"] < 0.03 * line["ms_per_step"], dpr
    assert 0.9 < dpr["predicted_eff_8gpu"] <= 1.02 and "ASSUMED" in dpr["wire"]
    # measured in the rank program's own run: the 1-rank all-reduce + average of the 31 MB buffer, and the step without it
    assert 0 < dpr["allreduce_ms"] < 2.0 and 0.9 * dpr["ms_per_step"] < dpr["no_comm_ms"] <= 1.02 * dpr["ms_per_step"], dpr
    assert line["f16x2_overflow"] == 0                        # the overflow sentinel of the two-piece GEMMs over the timed region
    st = line["roofline"]["step"]
    assert 100 < st["algorithmic_tflops"] < 400 and st["hbm_gb"] > st["algorithmic_gb"]
    assert 0 < st.get("frac_bf16_pipe_x3", st.get("frac_bf16_pipe_x6", 0)) < 1
    assert line["config"]["fp32_gemm"] == "f16x2" and set(line["fp32_modes"]) == {"split"}       # the other forms, same process
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["samples"] == 1
    # bf16 networks: the dominant GEMMs stream the grouped features -> priced against HBM
    assert line["configs"]["reg_bf16"]["bound"] == "hbm"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gather_rows_backward_accumulates_repeated_indices(gpu, dt):
    """ops.gather_rows (strided skip connection): the gradient of a source row that was sampled several times -- FPS repeats
    index 0 for clouds with fewer live points than samples -- is the SUM of its rows' gradients, as torch.gather gives."""
    from epn_pointcloud_amd import ops
    torch.manual_seed(5)
    feats = torch.randn(2, 16, 40, 60, device=gpu).to(dt).requires_grad_(True)
    idx = torch.stack([torch.arange(20), torch.cat([torch.arange(12), torch.zeros(8, dtype=torch.long)])]).int().to(gpu)
    g = torch.randn(2, 16, 20, 60, device=gpu).to(dt)
    (got,) = torch.autograd.grad(ops.gather_rows(feats, idx), [feats], g)
    ref_in = feats.detach().float().requires_grad_(True)
    ii = idx.long().view(2, 1, 20, 1).expand(-1, 16, -1, 60)
    (want,) = torch.autograd.grad(torch.gather(ref_in, 2, ii), [ref_in], g.float())
    tol = 1e-6 if dt == torch.float32 else 4e-2
    assert (got.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
