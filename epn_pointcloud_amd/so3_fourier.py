"""Block-diagonalising basis of IntraSO3Conv's anchor permutations (the "group Fourier transform" of the icosahedral
rotation group), derived numerically from `intra_idx` alone.

IntraSO3Conv (vgtk/vgtk/so3conv/modules.py:177-200, functional.py:255-268) is  out[a] = sum_k W_k F[intra_idx[a, k]]:
twelve anchor permutations P_k, each a right translation of the 60-element rotation group.  All P_k are block-
diagonalised by ONE orthogonal matrix U: the 60-dimensional (regular) representation splits into the group's irreducible
representations, each of dimension d occurring d times (1 + 3*3 + 3*3 + 4*4 + 5*5 = 60).  In that basis the layer is,
per irreducible rho and per copy m,
    out^rho[m] = sum_k W_k (x) rho(g_k) . F^rho[m]        (a [d*cin] -> [d*cout] linear map shared by the d copies)
i.e. five GEMMs with 2 * cin * cout * (1 + 27 + 27 + 64 + 125) = 244 multiply-adds per point in place of 720 -- the same
result up to fp32 rounding (U is orthogonal), 2.95x fewer flops, and no [cols, 12*cin] grouped tensor.

Nothing here assumes a particular labelling of the anchors: the group closure, the commuting (left-translation)
algebra, the invariant subspaces and their alignment are all computed from the permutation table; `check()` verifies
the result against the permutations to 1e-6.  Host-side float64 numpy, run once per distinct table and cached.
"""
import numpy as np

_CACHE = {}


def _closure(perms):
    """All products of the given permutations (as tuples): the group they generate, acting on range(n)."""
    n = len(perms[0])
    ident = tuple(range(n))
    seen = {ident}
    frontier = [ident]
    while frontier:
        nxt = []
        for g in frontier:
            for p in perms:
                h = tuple(p[g[i]] for i in range(n))       # (p o g)(i)
                if h not in seen:
                    seen.add(h)
                    nxt.append(h)
        frontier = nxt
        if len(seen) > 4 * n:
            raise ValueError("intra_idx columns do not generate a regular (free, transitive) permutation group")
    return sorted(seen)


def build(intra_idx):
    """intra_idx int[na, kn] whose columns are permutations generating a regular permutation group of order na.
    Returns dict(U [na,na] (columns grouped by irreducible, copy, component), dims [(d, n_copies=d)], rho list of
    [kn, d, d], row_of (f -> (irrep index, copy m, component j)))."""
    idx = np.asarray(intra_idx, dtype=np.int64)
    key = idx.tobytes() + bytes(idx.shape)
    if key in _CACHE:
        return _CACHE[key]
    na, kn = idx.shape
    perms = [tuple(int(v) for v in idx[:, k]) for k in range(kn)]
    for p in perms:
        if sorted(p) != list(range(na)):
            raise ValueError("every column of intra_idx must be a permutation of the anchors")
    group = _closure(perms)
    if len(group) != na:
        raise ValueError(f"the anchor permutations generate a group of order {len(group)}, expected {na}")
    # the commutant of a regular action is the opposite regular action: sigma_h(g(0)) = g(h)
    rng = np.random.default_rng(20260928)
    H = np.zeros((na, na))
    for h in range(na):
        sigma = np.zeros(na, dtype=np.int64)
        for g in group:
            sigma[g[0]] = g[h]
        S = np.zeros((na, na))
        S[sigma, np.arange(na)] = 1.0
        H += rng.standard_normal() * (S + S.T)
    evals, evecs = np.linalg.eigh(H)
    # clusters of (numerically) equal eigenvalues = irreducible invariant subspaces
    clusters, start = [], 0
    gap = 1e-8 * max(1.0, np.abs(evals).max())
    for i in range(1, na + 1):
        if i == na or evals[i] - evals[i - 1] > gap:
            clusters.append(evecs[:, start:i])
            start = i
    P = []
    for g in group:
        M = np.zeros((na, na))
        M[np.arange(na), list(g)] = 1.0                     # (M F)[a] = F[g(a)]
        P.append(M)
    Pk = [P[group.index(p)] for p in perms]

    def rep(V):
        return np.stack([V.T @ M @ V for M in P])           # [|G|, d, d]

    # group the clusters by irreducible type (dimension + character), align the copies of a type to its first copy
    types = []
    for V in clusters:
        r = rep(V)
        chi = np.round(np.trace(r, axis1=1, axis2=2), 6)
        for t in types:
            if t["d"] == V.shape[1] and np.allclose(t["chi"], chi, atol=1e-5):
                X = rng.standard_normal((V.shape[1], V.shape[1]))
                T = sum(a @ X @ b.T for a, b in zip(r, t["rep"]))      # intertwiner rho_s T = T rho_1 (Schur)
                u, _, vt = np.linalg.svd(T)
                t["V"].append(V @ (u @ vt))                            # now V^T P V == rep of the first copy
                break
        else:
            types.append({"d": V.shape[1], "chi": chi, "rep": r, "V": [V]})
    types.sort(key=lambda t: (t["d"], -t["chi"].max(), tuple(t["chi"])))
    cols, dims, rho, row_of = [], [], [], []
    for ti, t in enumerate(types):
        d = t["d"]
        if len(t["V"]) != d:
            raise ValueError("unexpected multiplicity in the decomposition of the regular representation")
        dims.append(d)
        rho.append(np.stack([t["V"][0].T @ M @ t["V"][0] for M in Pk]))   # [kn, d, d]
        for m, V in enumerate(t["V"]):
            for j in range(d):
                cols.append(V[:, j])
                row_of.append((ti, m, j))
    out = {"U": np.stack(cols, axis=1), "dims": dims, "rho": rho, "row_of": row_of, "perm_mats": Pk}
    check(out)
    _CACHE[key] = out
    return out


def check(b, tol=1e-6):
    """U orthogonal, and U^T P_k U = blockdiag over (irreducible, copy) of rho(g_k)."""
    U = b["U"]
    na = U.shape[0]
    assert np.abs(U.T @ U - np.eye(na)).max() < tol
    for k, M in enumerate(b["perm_mats"]):
        B = U.T @ M @ U
        want = np.zeros_like(B)
        off = 0
        for d, r in zip(b["dims"], b["rho"]):
            for _ in range(d):
                want[off:off + d, off:off + d] = r[k]
                off += d
        assert np.abs(B - want).max() < tol, (k, np.abs(B - want).max())
