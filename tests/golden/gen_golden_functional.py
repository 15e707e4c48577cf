#!/usr/bin/env python3
"""Golden fixtures for the FUNCTIONAL API rows of SURVEY.md 8(a) -- a7 `inter_zpconv_grouping_ball`, a11
`inter_so3conv_grouping` (fresh grouping and the inter_idx-reuse branch), a18 the pooling helpers
`inter_pooling_naive` / `inter_blurring_naive` / `inter_so3conv_blurring` -- by IMPORTING the reference
(build container only; same stand-ins as gen_golden.py, whose docstring describes them).

Run:  python tests/golden/gen_golden_functional.py     -> tests/golden/functional_api.npz (data only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def main():
    G.install_reference()
    import vgtk.so3conv.functional as L        # the reference's modules
    import vgtk.spconv.functional as Z
    rng = np.random.default_rng(4242)
    torch.manual_seed(4242)
    T = torch.from_numpy
    out = {}
    xyz = T(G.unit_ball_cloud(rng, 2, 96))
    # ---- a7: inter_zpconv_grouping_ball, strided with FPS and stride 1 lazily sampled
    for tag, stride, lazy, radius, nn in (("s2", 2, False, 0.45, 12), ("s1", 1, True, 0.4, 9)):
        gx, bidx, sidx, sxyz = Z.inter_zpconv_grouping_ball(xyz, stride, radius, nn, lazy)
        out[f"a7_{tag}_grouped_xyz"], out[f"a7_{tag}_ball_idx"] = gx.numpy(), bidx.numpy().astype(np.int32)
        out[f"a7_{tag}_sample_idx"], out[f"a7_{tag}_sample_xyz"] = sidx.numpy().astype(np.int32), sxyz.numpy()
        out[f"a7_{tag}_args"] = np.array([stride, int(lazy), radius, nn], dtype=np.float64)
    # ---- a11: inter_so3conv_grouping (functional), fresh and with inter_idx / inter_w handed back in
    anchors = T(L.get_anchors(60))
    kernels = T(L.get_sphereical_kernel_points_from_ply(0.7 * 0.45, 1))
    feats = torch.randn(2, 5, 96, 60)
    inter_idx, inter_w, new_xyz, new_feats, sample_idx = L.inter_so3conv_grouping(
        xyz, feats, 2, 12, anchors, kernels, 0.45, 0.1, None, None, False)
    out.update(a11_xyz=xyz.numpy(), a11_feats=feats.numpy(), a11_anchors=anchors.numpy(), a11_kernels=kernels.numpy(),
               a11_inter_idx=inter_idx.numpy().astype(np.int32), a11_inter_w_a6=inter_w[:, :, :6].numpy(),   # first 6 anchors (size)
               a11_new_xyz=new_xyz.numpy(), a11_new_feats=new_feats.numpy(),
               a11_sample_idx=sample_idx.numpy().astype(np.int32))
    feats2 = torch.randn(2, 3, 48, 60)                       # reuse branch: features on the 48 sampled points
    idx1, w1, _, _, _ = L.inter_so3conv_grouping(new_xyz, feats2, 1, 12, anchors, kernels, 0.45, 0.1, None, None, True)
    r_idx, r_w, r_xyz, r_feats, r_sidx = L.inter_so3conv_grouping(
        new_xyz, feats2, 1, 12, anchors, kernels, 0.45, 0.1, idx1, w1, True)
    assert r_sidx is None
    out.update(a11r_feats=feats2.numpy(), a11r_inter_idx=idx1.numpy().astype(np.int32),
               a11r_new_feats=r_feats.numpy(), a11r_new_xyz=r_xyz.numpy())
    # ---- a18: pooling helpers
    pool = Z.inter_pooling_naive(inter_idx, sample_idx, feats)
    blur = Z.inter_blurring_naive(idx1, feats2)
    bl_feats, bl_xyz = L.inter_so3conv_blurring(xyz, feats, 10, 0.45, 2, None, False)
    bl1_feats, bl1_xyz = L.inter_so3conv_blurring(new_xyz, feats2, 10, 0.45, 1, None, True)
    out.update(a18_pool=pool.numpy(), a18_blur=blur.numpy(), a18_blurring_s2_feats=bl_feats.numpy(),
               a18_blurring_s2_xyz=bl_xyz.numpy(), a18_blurring_s1_feats=bl1_feats.numpy(),
               a18_blurring_s1_xyz=bl1_xyz.numpy())
    np.savez_compressed(os.path.join(HERE, "functional_api.npz"), **out)
    print("functional_api.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
