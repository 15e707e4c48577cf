"""`vgtk.cuda.gathering` -- vgtk/vgtk/cuda/gathering_cuda.cpp:62-65."""
import torch

from ... import _lib


def gather_points_forward(support_points, grouped_indices):
    """(points f[b,c,n], idx int32[b,m]) -> float32 [b,c,m]  (gathering_cuda.cpp:29-43)."""
    lib = _lib.get_lib()
    p = _lib.dev_ptr(support_points, "support_points")
    i = _lib.dev_ptr(grouped_indices, "grouped_indices", torch.int32)
    b, c, n = support_points.shape
    m = grouped_indices.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=support_points.device)
    _lib.check(lib.epn_gather_fwd_f32(p, i, b, c, n, m, _lib.dev_ptr(out, "out"),
                                      _lib.stream_of(support_points)), "gather_points_forward")
    return out


def gather_points_backward(grad_out, grouped_indices, npoint):
    """(grad_out f[b,c,m], idx int32[b,m], int npoint) -> f[b,c,npoint]  (gathering_cuda.cpp:45-60)."""
    lib = _lib.get_lib()
    g = _lib.dev_ptr(grad_out, "grad_out")
    i = _lib.dev_ptr(grouped_indices, "grouped_indices", torch.int32)
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(npoint)), dtype=torch.float32, device=grad_out.device)
    _lib.check(lib.epn_gather_bwd_f32(g, i, b, c, int(npoint), m, _lib.dev_ptr(out, "grad_points"),
                                      _lib.stream_of(grad_out)), "gather_points_backward")
    return out
