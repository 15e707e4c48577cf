"""-m gpu: FPS / ball query / gather HIP kernels through the C ABI vs the C oracle and the golden fixtures.
Bar: bit-exact (int32 indices; gathered floats are copies)."""
import numpy as np
import pytest
import torch

from conftest import golden, unit_ball_cloud
from oracle import index_ref

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def ext(gpu, vgtk_alias):
    import vgtk.cuda.grouping as cuda_nn
    import vgtk.cuda.gathering as gather
    return cuda_nn, gather, gpu


def test_fps_golden(ext):
    cuda_nn, _, dev = ext
    g = golden("fps.npz")
    for tag in ("n256", "n1024", "n2048", "n300"):
        x, m = T(g[f"{tag}_xyz"]), int(g[f"{tag}_m"])
        idx = cuda_nn.furthest_point_sampling(x.to(dev), m)
        assert idx.dtype == torch.int32 and idx.is_cuda
        assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_idx"]), tag


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 3, 2), (3, 63, 17), (2, 64, 64), (2, 100, 33), (4, 513, 200),
                                   (2, 4096, 300), (1, 5000, 64), (1, 20000, 40)])
def test_fps_vs_oracle_shapes(ext, b, n, m):
    cuda_nn, _, dev = ext
    rng = np.random.default_rng(n * 131 + m)
    x = T(unit_ball_cloud(rng, b, n)) if n > 3 else T(rng.standard_normal((b, 3, n)).astype(np.float32))
    got = cuda_nn.furthest_point_sampling(x.to(dev), m).cpu()
    assert torch.equal(got, index_ref.furthest_point_sampling(x, m))


def test_fps_beyond_32768_points(ext):
    """Float FPS has no size limit in the reference (strided loop, grouping_cuda_kernel.cu:380-396): clouds beyond what the
    register-resident kernels hold go through epn_fps_temp_f32 (running minima in `temp`, the reference's own structure) --
    bit-exact against the oracle at n = 40000, and the same indices as the register kernels where both apply."""
    cuda_nn, _, dev = ext
    from epn_pointcloud_amd import _lib
    rng = np.random.default_rng(40000)
    x = T(unit_ball_cloud(rng, 2, 40000))
    got = cuda_nn.furthest_point_sampling(x.to(dev), 48).cpu()
    assert got.dtype == torch.int32 and torch.equal(got, index_ref.furthest_point_sampling(x, 48))
    lib = _lib.get_lib()
    for n, m in ((1000, 300), (6, 4), (20000, 30)):
        y = T(unit_ball_cloud(rng, 2, n)).to(dev)
        temp = torch.empty(2, n, device=dev)
        idx = torch.empty(2, m, dtype=torch.int32, device=dev)
        _lib.check(lib.epn_fps_temp_f32(y.data_ptr(), 2, n, m, temp.data_ptr(), idx.data_ptr(), _lib.stream_of(y)), "fps_temp")
        assert torch.equal(idx, cuda_nn.furthest_point_sampling(y, m)), n
    assert lib.epn_fps_f32(x.to(dev).data_ptr(), 2, 40000, 4, got.to(dev).data_ptr(), None) != 0     # EPN_EINVAL, not garbage


def test_fps_ties_and_degenerate(ext):
    cuda_nn, _, dev = ext
    rng = np.random.default_rng(1)
    # lattice points: massive distance ties exercise the reference's reduction-tree tie-break order
    grid = np.stack(np.meshgrid(*[np.arange(8)] * 3, indexing="ij"), 0).reshape(3, -1).astype(np.float32) * 0.25 - 0.8
    x = T(np.stack([grid, grid[:, rng.permutation(512)]], 0))
    assert torch.equal(cuda_nn.furthest_point_sampling(x.to(dev), 256).cpu(), index_ref.furthest_point_sampling(x, 256))
    # all points inside the |p|^2 <= 1e-3 dead zone -> every round returns 0
    z = T((rng.standard_normal((1, 3, 128)) * 1e-3).astype(np.float32))
    got = cuda_nn.furthest_point_sampling(z.to(dev), 16).cpu()
    assert torch.equal(got, index_ref.furthest_point_sampling(z, 16)) and (got == 0).all()
    # duplicated cloud
    d = T(unit_ball_cloud(rng, 1, 256)).repeat(1, 1, 2)
    assert torch.equal(cuda_nn.furthest_point_sampling(d.to(dev), 300).cpu(), index_ref.furthest_point_sampling(d, 300))


def test_ball_query_golden(ext):
    cuda_nn, _, dev = ext
    g = golden("ballq.npz")
    for tag in ("k16", "k32", "k128", "sparse", "ragged"):
        x, q, r, k = T(g[f"{tag}_xyz"]), T(g[f"{tag}_query"]), float(g[f"{tag}_r"]), int(g[f"{tag}_k"])
        idx = cuda_nn.ball_query(q.to(dev), x.to(dev), r, k)
        assert idx.dtype == torch.int32 and tuple(idx.shape) == (x.shape[0], q.shape[2], k)
        assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_idx"]), tag


@pytest.mark.parametrize("b,n,m,r,k", [(1, 1, 1, 0.5, 1), (2, 65, 9, 0.3, 5), (2, 500, 250, 0.15, 64),
                                       (3, 1024, 1024, 0.2828, 16), (1, 2048, 512, 0.08, 128),
                                       (2, 129, 129, 10.0, 130)])
def test_ball_query_vs_oracle(ext, b, n, m, r, k):
    cuda_nn, _, dev = ext
    rng = np.random.default_rng(n + 7 * m + k)
    x = T(unit_ball_cloud(rng, b, n)) if n > 3 else T(rng.standard_normal((b, 3, n)).astype(np.float32))
    q = x[:, :, rng.permutation(n)[:m]].contiguous()
    got = cuda_nn.ball_query(q.to(dev), x.to(dev), r, k).cpu()
    assert torch.equal(got, index_ref.ball_query(q, x, r, k))


def test_ball_query_full_size_properties(ext):
    """BASELINE configs[1] first layer (B=32, N=1024 -> 512 queries, r=0.2, K=32): size-independent
    properties + oracle on a slice."""
    cuda_nn, _, dev = ext
    rng = np.random.default_rng(2913)
    x = T(unit_ball_cloud(rng, 32, 1024))
    xd = x.to(dev)
    sidx = cuda_nn.furthest_point_sampling(xd, 512)
    q = torch.gather(xd, 2, sidx.long()[:, None].expand(-1, 3, -1)).contiguous()
    idx = cuda_nn.ball_query(q, xd, 0.2, 32)
    assert int(idx.min()) >= 0 and int(idx.max()) < 1024
    nb = torch.gather(xd[:, :, None].expand(-1, -1, 512, -1), 3, idx.long()[:, None].expand(-1, 3, -1, -1))
    d2 = ((nb - q[..., None]) ** 2).sum(1)
    # every listed neighbour is inside the ball, except the zero-filled slots of the K-1 / empty quirk
    assert bool(((d2 < 0.2 ** 2 + 1e-6) | (idx == 0)).all())
    assert bool((idx[:, :, 0] == torch.minimum(idx[:, :, 0], sidx)).all())  # first hit is the lowest index; the query itself is a hit
    assert torch.equal(idx[:2].cpu(), index_ref.ball_query(q[:2].cpu(), x[:2], 0.2, 32))
    assert torch.equal(sidx[:2].cpu(), index_ref.furthest_point_sampling(x[:2], 512))


def test_gather_fwd_bwd(ext):
    _, gather, dev = ext
    rng = np.random.default_rng(3)
    for (b, c, n, m) in [(1, 1, 1, 1), (2, 3, 1025, 512 * 32), (3, 7, 300, 1000)]:
        pts = T(rng.standard_normal((b, c, n)).astype(np.float32))
        idx = T(rng.integers(0, n, (b, m)).astype(np.int32))
        out = gather.gather_points_forward(pts.to(dev), idx.to(dev))
        assert out.dtype == torch.float32 and torch.equal(out.cpu(), index_ref.gather_points_forward(pts, idx))
        go = T(rng.standard_normal((b, c, m)).astype(np.float32))
        back = gather.gather_points_backward(go.to(dev), idx.to(dev), n)
        assert torch.allclose(back.cpu(), index_ref.gather_points_backward(go, idx, n), atol=1e-4, rtol=1e-5)


def test_group_nd_and_furthest_sample(ext, vgtk_alias):
    import vgtk.pc as pctk
    _, _, dev = ext
    rng = np.random.default_rng(9)
    x = T(unit_ball_cloud(rng, 2, 256)).to(dev)
    idx, sx = pctk.furthest_sample(x, 128, False)
    assert torch.equal(sx, torch.gather(x, 2, idx.long()[:, None].expand(-1, 3, -1)))
    bi = pctk.ball_query_index(sx, x, 0.4, 16)
    grouped = pctk.group_nd(x, bi)
    assert tuple(grouped.shape) == (2, 3, 128, 16)
    assert torch.equal(grouped[1, :, 5, 3], x[1, :, bi[1, 5, 3].long()])


@pytest.mark.parametrize("b,nc,m,na,ks,radius,sigma", [(2, 16, 500, 60, 24, 0.4, 0.08), (1, 5, 1, 12, 24, 0.3, 0.05),
                                                        (3, 7, 777, 60, 13, 0.25, 0.03), (1, 4, 300, 1, 24, 0.01, 0.02)])
def test_initial_anchor_query_vs_oracle(gpu, vgtk_alias, b, nc, m, na, ks, radius, sigma):
    """vgtk.cuda.grouping.initial_anchor_query (grouping_cuda.cpp:138-158): counts exact, weights to fp32 rounding of
    the same summation order; includes a chunk boundary (m > 256), a single fragment point and an (almost) empty ball."""
    import vgtk.cuda.grouping as cuda_nn
    from oracle import index_ref
    rng = np.random.default_rng(m + nc)
    centers = torch.from_numpy(unit_ball_cloud(rng, b, nc))
    frag = torch.from_numpy(rng.uniform(-0.6, 0.6, (m, 3)).astype(np.float32))              # [m, 3]
    if m == 1:
        frag[0] = centers[0, :, 0] + 0.05                                                  # inside the first ball
    kp = torch.from_numpy((rng.standard_normal((ks, na, 3)) * 0.2).astype(np.float32))
    w_ref, c_ref = index_ref.initial_anchor_query(centers, frag, kp, radius, sigma)
    w, c = cuda_nn.initial_anchor_query(centers.to(gpu), frag.to(gpu), kp.to(gpu), radius, sigma)
    assert tuple(w.shape) == (b, ks, nc, na) and tuple(c.shape) == (b, ks, nc, na)
    assert torch.equal(c.cpu(), c_ref)
    assert (w.cpu() - w_ref).abs().max().item() <= 1e-5 * max(1.0, w_ref.abs().max().item())


@pytest.mark.parametrize("b,nc,m,na,ks,radius,sigma", [(2, 16, 500, 60, 24, 0.4, 0.08), (1, 5, 1, 12, 24, 0.3, 0.05),
                                                        (3, 7, 777, 60, 13, 0.25, 0.03)])
def test_initial_anchor_query_f64_vs_oracle(gpu, vgtk_alias, b, nc, m, na, ks, radius, sigma):
    """The scalar_t = double instantiation (dispatch grouping_cuda_kernel.cu:558-563): outputs are float64, counts
    exact, weights to fp64 rounding of the summation order.  radius / sigma are FLOAT parameters in the reference: a
    fragment point placed between float(radius) and the double value of the same literal tells the two apart."""
    import vgtk.cuda.grouping as cuda_nn
    from oracle import index_ref
    rng = np.random.default_rng(m + nc + 1)
    centers = torch.from_numpy(unit_ball_cloud(rng, b, nc).astype(np.float64))
    frag = torch.from_numpy(rng.uniform(-0.6, 0.6, (m, 3)))
    # float(0.4) = 0.4000000059604645 > 0.4: this point is inside only when the radius is the float value
    frag[0] = centers[0, :, 0] + torch.tensor([0.5 * (float(np.float32(radius)) + radius), 0.0, 0.0], dtype=torch.float64)
    kp = torch.from_numpy(rng.standard_normal((ks, na, 3)) * 0.2)
    w_ref, c_ref = index_ref.initial_anchor_query(centers, frag, kp, radius, sigma)
    w, c = cuda_nn.initial_anchor_query(centers.to(gpu), frag.to(gpu), kp.to(gpu), radius, sigma)
    assert w.dtype == torch.float64 and c.dtype == torch.float64
    assert torch.equal(c.cpu(), c_ref)
    assert (w.cpu() - w_ref).abs().max().item() <= 1e-13 * max(1.0, w_ref.abs().max().item())
    # and it differs from the float32 run beyond float rounding somewhere (the f64 entry is not the f32 one widened)
    w32, _ = cuda_nn.initial_anchor_query(centers.float().to(gpu), frag.float().to(gpu), kp.float().to(gpu), radius, sigma)
    assert w32.dtype == torch.float32


def test_kernel_propagation_module(gpu, vgtk_alias):
    """KernelPropagation.forward (vgtk/vgtk/so3conv/modules.py:57-119) end to end against the oracle pieces."""
    import vgtk.so3conv as sptk
    from oracle import index_ref
    rng = np.random.default_rng(3)
    torch.manual_seed(3)
    clouds = torch.from_numpy(unit_ball_cloud(rng, 2, 64))
    frag = torch.from_numpy(np.ascontiguousarray(unit_ball_cloud(rng, 1, 400)[0].T))
    mod = sptk.KernelPropagation(1, 8, 16, 1, 0.4, 0.08).to(gpu)
    y = mod(frag.to(gpu), clouds.to(gpu))
    assert tuple(y.feats.shape) == (2, 8, 16, 60) and tuple(y.xyz.shape) == (2, 3, 16)
    sidx = index_ref.furthest_point_sampling(clouds, 16).long()
    centers = torch.gather(clouds, 2, sidx[:, None, :].expand(-1, 3, -1)).contiguous()
    assert torch.equal(y.xyz.cpu(), centers)
    w, c = index_ref.initial_anchor_query(centers, frag, mod.kernels.cpu(), 0.4, 0.08)
    g = (w / (c + 1.0)).unsqueeze(1)                                         # [b, 1, ks, nc, na]
    want = torch.matmul(mod.basic_conv.W.detach().cpu(), g.reshape(2, 24, 16 * 60)).view(2, 8, 16, 60)
    assert (y.feats.detach().cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


def test_legacy_zpconv_grouping_functions(gpu, vgtk_alias):
    """vgtk.cuda.zpconv.{inter,intra}_zpconv_{forward,backward} (zpconv_cuda.cpp:41-112) against the oracle's
    restatement and its autograd; out-of-range neighbour indices contribute nothing."""
    import vgtk.cuda.zpconv as zp
    from oracle import so3conv_ref as R
    torch.manual_seed(8)
    b, c, npts, nq, na, ks, ann = 2, 5, 9, 14, 12, 3, 4
    nbr = torch.randint(0, nq, (b, npts, na, ks, ann), dtype=torch.int32)
    nbr[0, 0, 0, 0, 0] = nq                    # a shadow / out-of-range slot
    w = torch.rand(b, npts, na, ks, ann)
    feats = torch.randn(b, c, nq, na, requires_grad=True)
    want = R.zp_inter_forward(nbr, w, feats)
    gy = torch.randn_like(want)
    (dwant,) = torch.autograd.grad(want, feats, gy)
    got = zp.inter_zpconv_forward(nbr.to(gpu), w.to(gpu), feats.detach().to(gpu))
    assert tuple(got.shape) == (b, c, ks, npts, na)
    assert (got.cpu() - want.detach()).abs().max().item() < 1e-5
    dgot = zp.inter_zpconv_backward(nbr.to(gpu), w.to(gpu), gy.to(gpu), nq)
    assert tuple(dgot.shape) == (b, c, nq, na)
    assert (dgot.cpu() - dwant).abs().max().item() < 1e-4

    na_in, na_out = 12, 20
    inbr = torch.randint(0, na_in, (na_out, ann), dtype=torch.int32)
    iw = torch.rand(na_out, ks, ann)
    f2 = torch.randn(b, c, npts, na_in, requires_grad=True)
    want = R.zp_intra_forward(inbr, iw, f2)
    gy = torch.randn_like(want)
    (dwant,) = torch.autograd.grad(want, f2, gy)
    got = zp.intra_zpconv_forward(inbr.to(gpu), iw.to(gpu), f2.detach().to(gpu))
    assert tuple(got.shape) == (b, c, ks, npts, na_out)
    assert (got.cpu() - want.detach()).abs().max().item() < 1e-5
    dgot = zp.intra_zpconv_backward(inbr.to(gpu), iw.to(gpu), gy.to(gpu), na_in)
    assert tuple(dgot.shape) == (b, c, npts, na_in)
    assert (dgot.cpu() - dwant).abs().max().item() < 1e-4


def test_anchor_query_vs_oracle(gpu, vgtk_alias):
    """vgtk.cuda.grouping.anchor_query (legacy ZPConv): HIP kernel vs the numpy restatement of
    grouping_cuda_kernel.cu:180-247, plus two values worked out by hand."""
    import vgtk.cuda.grouping as cuda_nn
    from oracle import index_ref
    rng = np.random.default_rng(5)
    b, p, nn, na, ks = 2, 37, 9, 12, 5
    g = torch.from_numpy(rng.standard_normal((b, 3, p, nn)).astype(np.float32) * 0.3)
    anc = rng.standard_normal((na, 3)).astype(np.float32)
    anc = torch.from_numpy(anc / np.linalg.norm(anc, axis=1, keepdims=True))
    kp = torch.from_numpy(rng.random((ks, 2)).astype(np.float32))
    sidx = torch.zeros(b, p, dtype=torch.int32)
    gidx = torch.zeros(b, p, nn, dtype=torch.int32)
    want = index_ref.anchor_query(sidx, gidx, g, anc, kp, 100)[0]
    got = cuda_nn.anchor_query(sidx.to(gpu), gidx.to(gpu), g.to(gpu), anc.to(gpu), kp.to(gpu), 100)[0]
    assert tuple(got.shape) == (b, p, na, ks, nn)
    assert (got.cpu() - want).abs().max().item() < 1e-5
    # by hand: g = (0, 0, 2), anchor +z, kernel point (kw, kh) = (1, 0): norm = 2 (+1e-6), theta = 0 -> (1-2)^2 + 0 = 1;
    #          anchor +x: theta = pi/2 -> 1 + (pi/2 * 2)^2 = 1 + pi^2
    g1 = torch.tensor([0.0, 0.0, 2.0]).view(1, 3, 1, 1)
    a1 = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    k1 = torch.tensor([[1.0, 0.0]])
    z = torch.zeros(1, 1, dtype=torch.int32, device=gpu)
    w = cuda_nn.anchor_query(z, z.view(1, 1, 1), g1.to(gpu), a1.to(gpu), k1.to(gpu), 1)[0].cpu().flatten()
    assert abs(w[0].item() - 1.0) < 1e-4 and abs(w[1].item() - (1.0 + np.pi ** 2)) < 1e-4


def test_anchor_query_f64_vs_oracle(gpu, vgtk_alias):
    """scalar_t = double (dispatch grouping_cuda_kernel.cu:505-510): float64 output within fp64 rounding of the numpy
    restatement; and the float32 entry's `norm` is float(double(|g|) + 1e-6), the double literal of :221 -- checked on
    the kw - norm term alone (anchor +g direction: theta = 0, kh = 0), where it is the only rounding that differs."""
    import vgtk.cuda.grouping as cuda_nn
    from oracle import index_ref
    rng = np.random.default_rng(6)
    b, p, nn, na, ks = 2, 37, 9, 12, 5
    g = torch.from_numpy(rng.standard_normal((b, 3, p, nn)) * 0.3)
    anc = rng.standard_normal((na, 3))
    anc = torch.from_numpy(anc / np.linalg.norm(anc, axis=1, keepdims=True))
    kp = torch.from_numpy(rng.random((ks, 2)))
    sidx = torch.zeros(b, p, dtype=torch.int32)
    gidx = torch.zeros(b, p, nn, dtype=torch.int32)
    want = index_ref.anchor_query(sidx, gidx, g, anc, kp, 100)[0]
    got = cuda_nn.anchor_query(sidx.to(gpu), gidx.to(gpu), g.to(gpu), anc.to(gpu), kp.to(gpu), 100)[0]
    assert got.dtype == torch.float64 and tuple(got.shape) == (b, p, na, ks, nn)
    assert (got.cpu() - want).abs().max().item() < 1e-12
    # float32: g = (0, 0, z), anchor +x (dot = 0, theta = acos(0)), kernel point (0, float(pi/2)): the angular term
    # vanishes and w = norm * norm in float
    z = np.float32(1e-7) + np.arange(1, 4097, dtype=np.float32) * np.float32(2.0 ** -47)  # consecutive floats near 1e-7
    g1 = torch.zeros(1, 3, 1, z.size)
    g1[0, 2, 0] = torch.from_numpy(z)
    a1 = torch.tensor([[1.0, 0.0, 0.0]])
    k1 = torch.tensor([[0.0, float(np.float32(np.pi / 2))]])
    zi = torch.zeros(1, 1, dtype=torch.int32, device=gpu)
    w = cuda_nn.anchor_query(zi, torch.zeros(1, 1, z.size, dtype=torch.int32, device=gpu), g1.to(gpu), a1.to(gpu),
                             k1.to(gpu), 1)[0].cpu().flatten().numpy()
    norm_d = (z.astype(np.float64) + 1e-6).astype(np.float32)            # double add, one rounding (the reference)
    norm_f = z + np.float32(1e-6)                                        # float add of the float literal
    assert (norm_d != norm_f).sum() > 50                                 # the two differ on this range ...
    assert np.array_equal(w, norm_d * norm_d)                            # ... and the kernel follows the reference's


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 3, 2), (3, 63, 17), (2, 100, 33), (4, 513, 200), (1, 5000, 64)])
def test_fps_f64_vs_oracle(ext, b, n, m):
    """fp64 dispatch of FPS (furthest_point_sampling_cuda_kernel<double>, grouping_cuda_kernel.cu:638-726): bit-exact
    indices against the double oracle, including clouds where float and double sampling differ."""
    cuda_nn, _, dev = ext
    rng = np.random.default_rng(n * 17 + m)
    x = T(rng.standard_normal((b, 3, n))) * 0.5                       # float64
    got = cuda_nn.furthest_point_sampling(x.to(dev), m).cpu()
    assert got.dtype == torch.int32
    assert torch.equal(got, index_ref.furthest_point_sampling(x, m))


@pytest.mark.parametrize("b,n,m,r,k", [(2, 200, 50, 0.4, 16), (1, 1000, 64, 0.2, 32), (3, 77, 77, 0.9, 8), (1, 130, 5, 1e-6, 4)])
def test_ball_query_f64_vs_oracle(ext, b, n, m, r, k):
    cuda_nn, _, dev = ext
    rng = np.random.default_rng(n + m)
    s = T(rng.standard_normal((b, 3, n))) * 0.5
    q = s[:, :, :m].contiguous()
    got = cuda_nn.ball_query(q.to(dev), s.to(dev), r, k).cpu()
    assert torch.equal(got, index_ref.ball_query(q, s, r, k))


def test_gather_f64_and_dtype_errors(ext):
    """gather_points fwd / bwd in double (gathering_cuda_kernel.cu:117,151) and the dtype contract of the wrappers."""
    cuda_nn, gather, dev = ext
    rng = np.random.default_rng(5)
    p = T(rng.standard_normal((2, 7, 90)))
    idx = T(rng.integers(0, 90, size=(2, 40)).astype(np.int32))
    out = gather.gather_points_forward(p.to(dev), idx.to(dev))
    assert out.dtype == torch.float64
    assert torch.equal(out.cpu(), index_ref.gather_points_forward(p, idx))
    g = T(rng.standard_normal((2, 7, 40)))
    gb = gather.gather_points_backward(g.to(dev), idx.to(dev), 90)
    assert gb.dtype == torch.float64
    assert (gb.cpu() - index_ref.gather_points_backward(g, idx, 90)).abs().max().item() < 1e-12
    with pytest.raises(TypeError):
        cuda_nn.furthest_point_sampling(p.half().to(dev), 4)
    with pytest.raises(TypeError):
        cuda_nn.ball_query(p[:, :3, :4].float().contiguous().to(dev), p[:, :3].contiguous().to(dev), 0.1, 4)   # mixed dtypes
