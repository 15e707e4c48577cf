"""`vgtk.cuda.gathering` -- vgtk/vgtk/cuda/gathering_cuda.cpp:62-65."""
import torch

from ... import _lib


def gather_points_forward(support_points, grouped_indices):
    """(points f[b,c,n], idx int32[b,m]) -> [b,c,m] of the points' dtype, float32 or float64
    (gathering_cuda.cpp:29-43; AT_DISPATCH_FLOATING_TYPES, gathering_cuda_kernel.cu:117)."""
    lib = _lib.get_lib()
    dt = _lib.float_dtype(support_points, "support_points")
    p = _lib.dev_ptr(support_points, "support_points", dt)
    i = _lib.dev_ptr(grouped_indices, "grouped_indices", torch.int32)
    b, c, n = support_points.shape
    m = grouped_indices.shape[1]
    out = torch.empty((b, c, m), dtype=dt, device=support_points.device)
    fn = lib.epn_gather_fwd_f64 if dt == torch.float64 else lib.epn_gather_fwd_f32
    _lib.check(fn(p, i, b, c, n, m, _lib.dev_ptr(out, "out", dt), _lib.stream_of(support_points)), "gather_points_forward")
    return out


def gather_points_backward(grad_out, grouped_indices, npoint):
    """(grad_out f[b,c,m], idx int32[b,m], int npoint) -> f[b,c,npoint], float32 or float64  (gathering_cuda.cpp:45-60)."""
    lib = _lib.get_lib()
    dt = _lib.float_dtype(grad_out, "grad_out")
    g = _lib.dev_ptr(grad_out, "grad_out", dt)
    i = _lib.dev_ptr(grouped_indices, "grouped_indices", torch.int32)
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(npoint)), dtype=dt, device=grad_out.device)
    fn = lib.epn_gather_bwd_f64 if dt == torch.float64 else lib.epn_gather_bwd_f32
    _lib.check(fn(g, i, b, c, int(npoint), m, _lib.dev_ptr(out, "grad_points", dt), _lib.stream_of(grad_out)),
               "gather_points_backward")
    return out
