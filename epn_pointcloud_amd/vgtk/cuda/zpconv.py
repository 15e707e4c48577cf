"""`vgtk.cuda.zpconv` -- the four grouping functions of the legacy ZPConv path, same names / signatures / layouts as
vgtk/vgtk/cuda/zpconv_cuda.cpp:41-117.  No shipped model reaches them (their Python callers are commented out in the
reference); provided for API completeness on simple HIP kernels: gather forward passes (atomic-free,
deterministic), atomic scatter backward passes."""
import torch

from ... import _lib


def _i32(t, name):
    if t.dtype != torch.int32:
        raise TypeError(f"{name} must be int32, got {t.dtype}")
    return _lib.dev_ptr(t, name, torch.int32)


def inter_zpconv_forward(anchor_neighbors, anchor_weights, support_point_feats):
    """(nbr i32[b,np,na,ks,ann], w f[b,np,na,ks,ann], feats f[b,c,nq,na]) -> f[b,c,ks,np,na]  (zpconv_cuda.cpp:41-56)."""
    lib = _lib.get_lib()
    b, np_, na, ks, ann = anchor_neighbors.shape
    c, nq = support_point_feats.shape[1], support_point_feats.shape[2]
    out = torch.empty((b, c, ks, np_, na), dtype=torch.float32, device=support_point_feats.device)
    _lib.check(lib.epn_zp_inter_fwd_f32(_i32(anchor_neighbors, "anchor_neighbors"),
                                        _lib.dev_ptr(anchor_weights, "anchor_weights"),
                                        _lib.dev_ptr(support_point_feats, "support_point_feats"), b, c, np_, nq, na, ks, ann,
                                        _lib.dev_ptr(out, "anchor_feats"), _lib.stream_of(out)), "inter_zpconv_forward")
    return out


def inter_zpconv_backward(anchor_neighbors, anchor_weights, grad_anchor_feats, npoint):
    """(nbr, w, grad f[b,c,ks,np,na], npoint) -> f[b,c,npoint,na]  (zpconv_cuda.cpp:58-75)."""
    lib = _lib.get_lib()
    b, np_, na, ks, ann = anchor_neighbors.shape
    c = grad_anchor_feats.shape[1]
    out = torch.empty((b, c, int(npoint), na), dtype=torch.float32, device=grad_anchor_feats.device)
    _lib.check(lib.epn_zp_inter_bwd_f32(_i32(anchor_neighbors, "anchor_neighbors"),
                                        _lib.dev_ptr(anchor_weights, "anchor_weights"),
                                        _lib.dev_ptr(grad_anchor_feats, "grad_anchor_feats"), b, c, np_, int(npoint), na, ks,
                                        ann, _lib.dev_ptr(out, "grad_feats"), _lib.stream_of(out)), "inter_zpconv_backward")
    return out


def intra_zpconv_forward(anchor_neighbors, anchor_weights, support_point_feats):
    """(nbr i32[na_out,ann], w f[na_out,ks,ann], feats f[b,c,np,na_in]) -> f[b,c,ks,np,na_out]  (zpconv_cuda.cpp:77-93)."""
    lib = _lib.get_lib()
    na_out, ann = anchor_neighbors.shape
    ks = anchor_weights.shape[1]
    b, c, np_, na_in = support_point_feats.shape
    out = torch.empty((b, c, ks, np_, na_out), dtype=torch.float32, device=support_point_feats.device)
    _lib.check(lib.epn_zp_intra_fwd_f32(_i32(anchor_neighbors, "anchor_neighbors"),
                                        _lib.dev_ptr(anchor_weights, "anchor_weights"),
                                        _lib.dev_ptr(support_point_feats, "support_point_feats"), b, c, np_, na_in, na_out, ks,
                                        ann, _lib.dev_ptr(out, "anchor_feats"), _lib.stream_of(out)), "intra_zpconv_forward")
    return out


def intra_zpconv_backward(anchor_neighbors, anchor_weights, grad_anchor_feats, anchor_in):
    """(nbr, w, grad f[b,c,ks,np,na_out], anchor_in) -> f[b,c,np,anchor_in]  (zpconv_cuda.cpp:95-112)."""
    lib = _lib.get_lib()
    na_out, ann = anchor_neighbors.shape
    ks = anchor_weights.shape[1]
    b, c, _, np_, _ = grad_anchor_feats.shape
    out = torch.empty((b, c, np_, int(anchor_in)), dtype=torch.float32, device=grad_anchor_feats.device)
    _lib.check(lib.epn_zp_intra_bwd_f32(_i32(anchor_neighbors, "anchor_neighbors"),
                                        _lib.dev_ptr(anchor_weights, "anchor_weights"),
                                        _lib.dev_ptr(grad_anchor_feats, "grad_anchor_feats"), b, c, np_, int(anchor_in),
                                        na_out, ks, ann, _lib.dev_ptr(out, "grad_feats"), _lib.stream_of(out)),
               "intra_zpconv_backward")
    return out
