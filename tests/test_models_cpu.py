"""Network level, no GPU: (1) the oracle's restatement of PointnetSO3Conv and of the three networks against golden
vectors generated from the unmodified reference builders (tests/golden/gen_golden_models.py), (2) the product models'
state_dict layout (keys and shapes) against the reference's, so reference checkpoints load unchanged."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from fill import fill_state_dict  # noqa: E402

T = torch.from_numpy
TOL = 1e-3   # fp32 feature tolerance of BASELINE.json's north_star


def tiny_layers(kind):
    from epn_pointcloud_amd import schedule as S
    if kind == "cls":
        return S.cls_so3net_schedule(256, mlps=((16, 16), (32, 32)), strides=(2, 2))
    if kind == "reg":
        return S.reg_so3net_schedule(256, mlps=((16, 16), (32,)), strides=(2, 2))
    return S.inv_so3net_schedule(1024, 0.4, mlps=((16, 16), (32, 32)), strides=(2, 2))


def tables():
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    from epn_pointcloud_amd.vgtk import functional as fr
    return (T(L.get_anchors(60)), T(fr.kernel_points_raw(24)), T(L.get_intra_idx()).long())


def product_model(kind):
    from epn_pointcloud_amd import models as M
    if kind == "cls":
        return M.ClsSO3ConvModel(tiny_layers("cls"), out_mlps=(32,), pooling="attention")
    if kind == "reg":
        return M.RegSO3ConvModel(tiny_layers("reg"), out_mlps=(32, 16))
    return M.InvSO3ConvModel(tiny_layers("inv"), out_mlps=(32, 16))


def oracle_model(kind):
    from oracle import backbone_ref as B
    if kind == "cls":
        return B.RefClsModel(tiny_layers("cls"), tables(), out_mlps=(32,), pooling="attention")
    if kind == "reg":
        return B.RefRegModel(tiny_layers("reg"), tables(), out_mlps=(32, 16))
    return B.RefInvModel(tiny_layers("inv"), tables(), out_mlps=(32, 16))


def filled_oracle(kind):
    """Oracle network holding the deterministic fill under the REFERENCE's key names (via the product's state_dict)."""
    prod = fill_state_dict(product_model(kind))
    ref = oracle_model(kind)
    ref.load_from_product(prod.state_dict())
    return ref.train()


@pytest.mark.parametrize("kind", ["cls", "reg", "inv"])
def test_state_dict_layout_matches_reference(kind):
    g = golden(f"model_{kind}_tiny.npz")
    m = product_model(kind)
    mine = sorted(f"{k}:{'x'.join(str(d) for d in v.shape)}" for k, v in m.state_dict().items())
    assert mine == sorted(g["layout"].tolist())


@pytest.mark.parametrize("tag", ["a60", "a1"])
def test_oracle_pointnet_vs_reference_golden(tag):
    from oracle import so3conv_ref as R
    from epn_pointcloud_amd.vgtk import so3conv as sptk
    g = golden(f"pointnet_{tag}.npz")
    f = T(g["feats"]).requires_grad_(True)
    m = fill_state_dict(sptk.PointnetSO3Conv(f.shape[1], g["out"].shape[1], 60))
    assert sorted(f"{k}:{'x'.join(str(d) for d in v.shape)}" for k, v in m.state_dict().items()) == sorted(g["layout"].tolist())
    w, b = m.embed.weight.detach().requires_grad_(True), m.embed.bias.detach().requires_grad_(True)
    y = R.pointnet_so3conv(T(g["xyz"]), f, m.anchors, w, b)
    assert (y.detach() - T(g["out"])).abs().max().item() < 1e-5
    dF, dW, dB = torch.autograd.grad(y, [f, w, b], T(g["gy"]))
    assert (dF - T(g["dF"])).abs().max().item() < 1e-5
    assert (dW - T(g["dW"])).abs().max().item() < 1e-4
    assert (dB - T(g["dB"])).abs().max().item() < 1e-4


def test_oracle_cls_model_vs_reference_golden():
    g = golden("model_cls_tiny.npz")
    ref = filled_oracle("cls")
    logits, att = ref(T(g["pts"]))
    assert (logits.detach() - T(g["logits"])).abs().max().item() < TOL
    assert (att.detach() - T(g["attention"])).abs().max().item() < TOL
    loss = torch.nn.functional.cross_entropy(logits, T(g["labels"]))
    assert abs(loss.item() - float(g["loss"])) < TOL
    pd = dict(ref.named_parameters())
    for i, n in enumerate(g["grad_names"].tolist()):
        (gr,) = torch.autograd.grad(loss, pd[n], retain_graph=True)
        want = T(g[f"grad{i}"]).reshape(gr.shape)
        assert (gr - want).abs().max().item() < TOL * max(1.0, want.abs().max().item()), n


def test_oracle_reg_model_vs_reference_golden():
    g = golden("model_reg_tiny.npz")
    conf, quats = filled_oracle("reg")(T(g["pairs"]))
    assert (conf.detach() - T(g["confidence"])).abs().max().item() < TOL
    assert (quats.detach() - T(g["quats"])).abs().max().item() < TOL


def test_oracle_inv_model_vs_reference_golden():
    g = golden("model_inv_tiny.npz")
    desc, attn = filled_oracle("inv")(T(g["pts"]))
    assert (desc.detach() - T(g["descriptor"])).abs().max().item() < TOL
    assert (attn[:, :, ::8].detach() - T(g["attention_sub"])).abs().max().item() < TOL


def test_kanchor20_state_dict_layout_matches_reference():
    """kanchor = 20: the reference builds 'inter_block's (cls_so3net_pn.py:127); the product must build the same tree."""
    from epn_pointcloud_amd import models as M
    g = golden("model_cls_k20_tiny.npz")
    m = M.ClsSO3ConvModel(tiny_layers("cls"), out_mlps=(32,), pooling="attention", kanchor=20)
    mine = sorted(f"{k}:{'x'.join(str(d) for d in v.shape)}" for k, v in m.state_dict().items())
    assert mine == sorted(g["layout"].tolist())


def test_separable_block_rejects_unsupported_kanchor():
    from epn_pointcloud_amd import schedule as S
    with pytest.raises(NotImplementedError):
        S.SeparableBlock(tiny_layers("cls")[1], kanchor=20)
    blk = S.SeparableBlock(tiny_layers("cls")[1], kanchor=1)         # use_intra = kanchor > 1 (base_so3conv.py:177)
    assert not hasattr(blk, "intra_conv")
