"""Structural known-answer tests of the shipped icosahedral tables (SURVEY.md section 4 "property" rows)."""
import numpy as np
import pytest

from conftest import golden


def _tables():
    import epn_pointcloud_amd  # noqa: F401
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    return L.get_anchors(60), L.get_intra_idx()


def _index_of(R, Rs):
    d = np.abs(Rs - R[None]).reshape(len(Rs), -1).max(1)
    j = int(np.argmin(d))
    assert d[j] < 1e-4
    return j


def test_tables_equal_reference_fixture():
    Rs, idx = _tables()
    g = golden("tables.npz")
    assert Rs.dtype == np.float32 and Rs.shape == (60, 3, 3) and np.array_equal(Rs, g["anchors60"])
    assert idx.dtype == np.int64 and idx.shape == (60, 12) and np.array_equal(idx, g["intra_idx"])


def test_anchor_group_structure():
    Rs, _ = _tables()
    assert np.array_equal(Rs[29], np.eye(3, dtype=np.float32))
    assert np.allclose(np.linalg.det(Rs.astype(np.float64)), 1.0, atol=1e-5)
    assert np.allclose(np.einsum('aij,akj->aik', Rs, Rs), np.eye(3)[None], atol=1e-5)
    for a in range(0, 60, 7):                      # closure: products stay in the set
        for b in range(60):
            _index_of(Rs[a] @ Rs[b], Rs)
    orders = []
    for a in range(60):                            # element orders of A5: {1:1, 2:15, 3:20, 5:24}
        M, k = Rs[a].astype(np.float64), 1
        P = M.copy()
        while not np.allclose(P, np.eye(3), atol=1e-4):
            P = P @ M
            k += 1
        orders.append(k)
    assert {o: orders.count(o) for o in set(orders)} == {1: 1, 2: 15, 3: 20, 5: 24}


def test_intra_idx_is_group_action():
    Rs, idx = _tables()
    G = [Rs[idx[29, k]] for k in range(12)]        # neighbours of the identity anchor
    for a in range(60):
        for k in range(12):
            assert idx[a, k] == _index_of(Rs[a] @ G[k], Rs)
    for k in range(12):                            # every column is a permutation of 0..59
        assert sorted(idx[:, k].tolist()) == list(range(60))
    assert (idx[:, 9] == np.arange(60)).all()      # slot 9 is the identity neighbour


def test_select_anchor():
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    g = golden("tables.npz")
    assert np.array_equal(L.get_anchors(20), g["select20"])
    assert np.array_equal(L.get_anchors(40), g["select40"])
    assert np.array_equal(L.get_anchors(1), g["select1"])
    assert L.get_anchors(12).shape == (60, 3, 3)   # anything else -> all 60 (functional.py:281-289)
    k = L.get_sphereical_kernel_points_from_ply(0.7 * 0.4, 1)
    assert k.dtype == np.float32 and np.array_equal(k, g["kernels_r0p4"])


def test_spectral_basis_block_diagonalises_the_anchor_permutations():
    """so3_fourier.build: derived from intra_idx alone; U orthogonal, U^T P_k U = blockdiag(rho(g_k) repeated d times)
    for the 1 + 3 + 3 + 4 + 5 dimensional irreducibles, and the block-diagonal form reproduces the 12-neighbour
    convolution (float64, 1e-10)."""
    from epn_pointcloud_amd import so3_fourier as sf
    from epn_pointcloud_amd.vgtk.so3conv import functional as L
    idx = L.get_intra_idx()
    bz = sf.build(idx)
    assert bz["dims"] == [1, 3, 3, 4, 5]
    sf.check(bz, tol=1e-9)
    rng = np.random.default_rng(0)
    cin, cout = 3, 2
    F, W = rng.standard_normal((60, cin)), rng.standard_normal((cout, cin, 12))
    direct = sum(F[idx[:, k]] @ W[:, :, k].T for k in range(12))
    Fh, out_h, off = bz["U"].T @ F, np.zeros((60, cout)), 0
    for d, r in zip(bz["dims"], bz["rho"]):
        What = np.einsum('ock,kij->jcio', W, r).reshape(d * cin, d * cout)
        for _ in range(d):
            out_h[off:off + d] = (Fh[off:off + d].reshape(1, d * cin) @ What).reshape(d, cout)
            off += d
    assert np.abs(bz["U"] @ out_h - direct).max() < 1e-10
    # a table that is not a regular group action is refused (callers then keep the 12-neighbour forms)
    bad = idx.copy()
    bad[:, 0] = np.roll(np.arange(60), 1)
    with pytest.raises((ValueError, AssertionError)):
        sf.build(bad)
