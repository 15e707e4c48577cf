#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference (build container only).

Run:  python tests/golden/gen_golden.py         (needs /root/reference; never runs on the GPU box)

The reference's Python (vgtk.so3conv / vgtk.spconv / SPConvNets blocks) is imported from
/root/reference with small ``sys.modules`` stand-ins for packages that are absent in this image and
for the CUDA-only extensions:

* ``plyfile``   -> a PLY reader (ASCII kpsphere*.ply, binary sphere12.ply).  No arithmetic.
* ``trimesh``   -> load(): faces, face_normals (normalised cross products), face_adjacency (face
                   pairs sharing an edge, ordered by the (v_max, v_min) key of the shared edge),
                   fix_normals() no-op (sphere12.ply is already outward/consistent).  trimesh==3.2.0
                   is the reference's pinned dependency (requirements.txt:7); the adjacency ROW ORDER
                   only permutes the 12 columns of ``intra_idx`` consistently, and no reference test
                   pins it -> "intra_idx column order: parity unpinned".
* ``vgtk.cuda.{grouping,gathering,zpconv}`` -> oracle/index_ref.py (C restatement of the .cu files;
                   parity vs the CUDA binary unpinned, see oracle/epn_oracle.c).

Fixtures are DATA ONLY (inputs + expected outputs as .npz); nothing of the reference's source travels.
Inputs are drawn from fixed seeds and stored explicitly.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


# ----------------------------------------------------------------------------- stand-ins
def _read_ply(path):
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii").splitlines()
    fmt = [l for l in header if l.startswith("format")][0].split()[1]
    nv = int([l for l in header if l.startswith("element vertex")][0].split()[2])
    nf_l = [l for l in header if l.startswith("element face")]
    nf = int(nf_l[0].split()[2]) if nf_l else 0
    body = raw[end:]
    if fmt == "ascii":
        rows = body.decode("ascii").split("\n")
        v = np.array([[float(t) for t in rows[i].split()[:3]] for i in range(nv)], dtype=np.float32)
        return v, None
    # binary_little_endian: vertex = 3 x f32 + 4 x u8 ; face = u8 n + n x i32 + u8 m + m x f32 + 4 x u8
    vdt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("c", "u1", 4)])
    vv = np.frombuffer(body, dtype=vdt, count=nv)
    v = np.stack([vv["x"], vv["y"], vv["z"]], axis=1).astype(np.float32)
    off = nv * vdt.itemsize
    faces = []
    for _ in range(nf):
        n = body[off]; off += 1
        faces.append(np.frombuffer(body, dtype="<i4", count=n, offset=off).copy()); off += 4 * n
        m = body[off]; off += 1 + 4 * m
        off += 4
    return v, np.stack(faces).astype(np.int64)


class _PlyData:
    def __init__(self, v):
        self._v = {"x": v[:, 0], "y": v[:, 1], "z": v[:, 2]}

    @staticmethod
    def read(path):
        v, _ = _read_ply(path)
        return _PlyData(v)

    def __getitem__(self, key):
        assert key == "vertex"
        return self._v


class _Mesh:
    def __init__(self, path):
        v, f = _read_ply(path)
        self.vertices = v.astype(np.float64)
        self.faces = f
        e1 = self.vertices[f[:, 1]] - self.vertices[f[:, 0]]
        e2 = self.vertices[f[:, 2]] - self.vertices[f[:, 0]]
        n = np.cross(e1, e2)
        self.face_normals = n / np.linalg.norm(n, axis=1, keepdims=True)
        edges = {}
        for fi, tri in enumerate(f):
            for a, b in ((0, 1), (1, 2), (2, 0)):
                key = (max(tri[a], tri[b]), min(tri[a], tri[b]))
                edges.setdefault(key, []).append(fi)
        rows = [sorted(fs) for key, fs in sorted(edges.items()) if len(fs) == 2]
        self.face_adjacency = np.array(rows, dtype=np.int64)

    def fix_normals(self):
        pass


def install_reference():
    ply = types.ModuleType("plyfile"); ply.PlyData = _PlyData; ply.PlyElement = object
    sys.modules["plyfile"] = ply
    tm = types.ModuleType("trimesh"); tm.load = lambda p: _Mesh(p)
    sys.modules["trimesh"] = tm
    for name in ("open3d", "colour"):
        sys.modules[name] = types.ModuleType(name)
    parse = types.ModuleType("parse"); parse.parse = lambda *a, **k: None
    sys.modules["parse"] = parse

    from oracle import index_ref
    cuda = types.ModuleType("vgtk.cuda"); cuda.__path__ = []
    grouping = types.ModuleType("vgtk.cuda.grouping")
    grouping.ball_query = index_ref.ball_query
    grouping.furthest_point_sampling = index_ref.furthest_point_sampling
    grouping.initial_anchor_query = None
    grouping.anchor_query = None
    gathering = types.ModuleType("vgtk.cuda.gathering")
    gathering.gather_points_forward = index_ref.gather_points_forward
    gathering.gather_points_backward = index_ref.gather_points_backward
    zp = types.ModuleType("vgtk.cuda.zpconv")
    sys.modules["vgtk.cuda"] = cuda
    sys.modules["vgtk.cuda.grouping"] = grouping
    sys.modules["vgtk.cuda.gathering"] = gathering
    sys.modules["vgtk.cuda.zpconv"] = zp
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "vgtk"))
    import vgtk  # noqa: F401  (the reference's package)
    return vgtk


# ----------------------------------------------------------------------------- synthetic clouds
def unit_ball_cloud(rng, b, n, scale=1.0):
    """uniform in the unit ball, centred, max-norm 1, channel-major [b,3,n] float32."""
    g = rng.standard_normal((b, n, 3))
    u = rng.random((b, n, 1))
    p = g / np.linalg.norm(g, axis=2, keepdims=True) * u ** (1.0 / 3.0)
    p = p - p.mean(axis=1, keepdims=True)
    p = p / np.linalg.norm(p, axis=2).max(axis=1)[:, None, None]
    return np.ascontiguousarray((scale * p).transpose(0, 2, 1).astype(np.float32))


def main():
    vgtk = install_reference()
    import vgtk.so3conv as sptk
    import vgtk.so3conv.functional as L
    import vgtk.spconv as zptk
    from oracle import index_ref
    out = lambda name, **kw: np.savez_compressed(os.path.join(HERE, name), **kw)

    # ---- constant tables (a16, a13 buffers)
    anchors = L.get_anchors(60).astype(np.float32)
    intra_idx = L.get_intra_idx().astype(np.int64)
    kp_raw = vgtk.pc.load_ply(os.path.join(REF, "vgtk/vgtk/data/anchors/kpsphere24.ply")).astype(np.float32)
    kern_r04 = L.get_sphereical_kernel_points_from_ply(0.7 * 0.4, 1).astype(np.float32)
    out("tables.npz", anchors60=anchors, intra_idx=intra_idx, kpsphere24=kp_raw,
        kernels_r0p4=kern_r04, select20=L.get_anchors(20), select40=L.get_anchors(40),
        select1=L.get_anchors(1))

    # ---- FPS / ball query / gather from the C restatement (index parity fixtures)
    rng = np.random.default_rng(2913)
    fps_cases = {}
    for tag, (b, n, m) in {"n256": (2, 256, 128), "n1024": (2, 1024, 512), "n2048": (1, 2048, 512),
                           "n300": (2, 300, 77)}.items():
        x = unit_ball_cloud(rng, b, n)
        if tag == "n256":      # points inside |p|^2 <= 1e-3 and exact duplicates (ties)
            x[0, :, 5] = 0.0; x[0, :, 17] = [0.01, 0.01, 0.01]; x[0, :, 40] = x[0, :, 41]
            x[1, :, 100:104] = x[1, :, 200:204]
        fps_cases[f"{tag}_xyz"] = x
        fps_cases[f"{tag}_m"] = np.int64(m)
        fps_cases[f"{tag}_idx"] = index_ref.furthest_point_sampling(torch.from_numpy(x), m).numpy()
    out("fps.npz", **fps_cases)

    bq = {}
    for tag, (b, n, m, r, k) in {"k16": (2, 256, 128, 0.4, 16), "k32": (2, 1024, 512, 0.2, 32),
                                 "k128": (1, 2048, 512, 0.32, 128), "sparse": (2, 256, 256, 0.12, 16),
                                 "ragged": (3, 77, 33, 0.5, 7)}.items():
        x = unit_ball_cloud(rng, b, n)
        q = x[:, :, :m].copy()
        if tag == "sparse":
            q[0, :, 3] = 5.0      # a query with no neighbour at all -> all-zero row
        bq[f"{tag}_xyz"] = x; bq[f"{tag}_query"] = q
        bq[f"{tag}_r"] = np.float32(r); bq[f"{tag}_k"] = np.int64(k)
        bq[f"{tag}_idx"] = index_ref.ball_query(torch.from_numpy(q), torch.from_numpy(x), r, k).numpy()
    out("ballq.npz", **bq)

    # ---- inter weights (a8) from the reference functional
    torch.manual_seed(2913)
    g = (torch.rand(2, 3, 24, 16) - 0.5) * 0.5
    A = torch.from_numpy(anchors); Kp = torch.from_numpy(kern_r04)
    tet = [3, 4, 5, 27, 28, 29, 39, 40, 41, 48, 49, 50]
    out("interw.npz", grouped_xyz=g.numpy(), anchors60=anchors, kernels=kern_r04, sigma=np.float32(0.08),
        w60=L.inter_so3conv_grouping_anchor(g, A, Kp, 0.08).numpy(),
        tet_index=np.array(tet), w12=L.inter_so3conv_grouping_anchor(g, A[tet], Kp, 0.08).numpy())

    # ---- inter / intra feature grouping + autograd grads (a9, a10, a14, a17)
    b, c, p1, p2, nn, na = 2, 5, 48, 12, 8, 60
    idx = torch.randint(0, p1, (b, p2, nn), dtype=torch.int32)
    w = torch.relu(torch.rand(b, p2, na, 24, nn) - 0.6)
    feats = torch.randn(b, c, p1, na, requires_grad=True)
    G = zptk.inter_zpconv_grouping_naive(idx, w, zptk.add_shadow_feature(feats))
    gG = torch.randn_like(G)
    (dF,) = torch.autograd.grad(G, feats, gG)
    out("inter_group.npz", idx=idx.numpy(), w=w.numpy(), feats=feats.detach().numpy(), G=G.detach().numpy(),
        gG=gG.numpy(), dF=dF.numpy())
    feats2 = torch.randn(2, 6, 24, 60, requires_grad=True)
    G2 = L.intra_so3conv_grouping(torch.from_numpy(intra_idx), feats2)
    gG2 = torch.randn_like(G2)
    (dF2,) = torch.autograd.grad(G2, feats2, gG2)
    out("intra_group.npz", intra_idx=intra_idx, feats=feats2.detach().numpy(), G=G2.detach().numpy(),
        gG=gG2.numpy(), dF=dF2.numpy())

    # ---- module level: BASELINE configs[0] plumbing (B=2, N=256, K=16, A=60), fwd + grads
    x = torch.from_numpy(unit_ball_cloud(rng, 2, 256))
    torch.manual_seed(7)
    for tag, (cin, cout, stride, lazy) in {"s2_fps": (1, 8, 2, False), "s1_lazy": (6, 8, 1, True)}.items():
        conv = sptk.InterSO3Conv(cin, cout, 1, stride, 0.4, 0.08, 16, lazy_sample=lazy, kanchor=60)
        f = (torch.ones(2, 1, 256, 60) if cin == 1 else torch.randn(2, cin, 256, 60)).requires_grad_(True)
        iidx, iw, sidx, y = conv(zptk.SphericalPointCloud(x, f, None))
        gy = torch.randn_like(y.feats)
        dW, dF = torch.autograd.grad(y.feats, [conv.basic_conv.W, f], gy)
        out(f"inter_module_{tag}.npz", xyz=x.numpy(), feats=f.detach().numpy(), W=conv.basic_conv.W.detach().numpy(),
            anchors=conv.anchors.numpy(), kernels=conv.kernels.numpy(), stride=np.int64(stride),
            radius=np.float32(0.4), sigma=np.float32(0.08), n_neighbor=np.int64(16), lazy=np.bool_(lazy),
            inter_idx=iidx.numpy(), inter_w_sub=iw[:, ::16].numpy(), sample_idx=sidx.numpy(), new_xyz=y.xyz.numpy(),
            out=y.feats.detach().numpy(), gy=gy.numpy(), dW=dW.numpy(), dF=dF.numpy(),
            state_keys=np.array(sorted(conv.state_dict().keys())))
    conv = sptk.IntraSO3Conv(8, 8)
    f = torch.randn(2, 8, 128, 60, requires_grad=True)
    y = conv(zptk.SphericalPointCloud(x[:, :, :128].contiguous(), f, None))
    gy = torch.randn_like(y.feats)
    dW, dF = torch.autograd.grad(y.feats, [conv.basic_conv.W, f], gy)
    out("intra_module.npz", feats=f.detach().numpy(), W=conv.basic_conv.W.detach().numpy(),
        intra_idx=conv.intra_idx.numpy(), anchors=conv.anchors.numpy(), out=y.feats.detach().numpy(),
        gy=gy.numpy(), dW=dW.numpy(), dF=dF.numpy(), state_keys=np.array(sorted(conv.state_dict().keys())))

    # ---- block level: SeparableSO3ConvBlock from the unmodified SPConvNets builder semantics (tiny widths)
    import SPConvNets.utils.base_so3conv as M
    torch.manual_seed(11)
    params = dict(dim_in=1, dim_out=8, kernel_size=1, stride=2, radius=0.4, sigma=0.08, n_neighbor=16,
                  lazy_sample=False, dropout_rate=0.0, multiplier=2, activation='leaky_relu', pooling=None,
                  kanchor=60, norm='BatchNorm2d')
    blk = M.SeparableSO3ConvBlock(dict(params)); blk.train()
    xin = M.preprocess_input(x.permute(0, 2, 1).contiguous(), 60, False)
    _, _, sidx, y = blk(xin, None, None)
    sd = {k: v.numpy() for k, v in blk.state_dict().items()}
    out("sepblock_tiny.npz", xyz=x.numpy(), out=y.feats.detach().numpy(), sample_idx=sidx.numpy(),
        **{"sd/" + k: v for k, v in sd.items()})
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
