"""oracle/index_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end to ``libepn_oracle.so`` (oracle/epn_oracle.c): the CPU
restatement of the reference's CUDA-only index kernels

* ``ball_query``               vgtk/vgtk/cuda/grouping_cuda.cpp:71-86
* ``furthest_point_sampling``  vgtk/vgtk/cuda/grouping_cuda.cpp:160-174
* ``gather_points_forward``    vgtk/vgtk/cuda/gathering_cuda.cpp:29-43
* ``gather_points_backward``   vgtk/vgtk/cuda/gathering_cuda.cpp:45-60
* ``initial_anchor_query``     vgtk/vgtk/cuda/grouping_cuda.cpp:138-158

with the same call signatures as the pybind functions (torch CPU tensors in,
freshly allocated torch CPU tensors out).  Parity vs the CUDA binary is
UNPINNED (see the header of epn_oracle.c); parity HIP-vs-oracle is bit-exact.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libepn_oracle.so")


def build(force=False):
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "epn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libepn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        ci = ctypes.c_int
        lib.epn_oracle_ball_query_f32.argtypes = [f32p, f32p, ci, ci, ci, ctypes.c_float, ci, i32p]
        lib.epn_oracle_fps_f32.argtypes = [f32p, ci, ci, ci, f32p, i32p]
        lib.epn_oracle_gather_fwd_f32.argtypes = [f32p, i32p, ci, ci, ci, ci, f32p]
        lib.epn_oracle_gather_bwd_f32.argtypes = [f32p, i32p, ci, ci, ci, ci, f32p]
        lib.epn_oracle_initial_anchor_query_f32.argtypes = [f32p, f32p, f32p, ci, ci, ci, ci, ci, ctypes.c_float,
                                                            ctypes.c_float, f32p, f32p]
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.epn_oracle_ball_query_f64.argtypes = [f64p, f64p, ci, ci, ci, ctypes.c_float, ci, i32p]
        lib.epn_oracle_fps_f64.argtypes = [f64p, ci, ci, ci, f64p, i32p]
        lib.epn_oracle_gather_fwd_f64.argtypes = [f64p, i32p, ci, ci, ci, ci, f64p]
        lib.epn_oracle_gather_bwd_f64.argtypes = [f64p, i32p, ci, ci, ci, ci, f64p]
        lib.epn_oracle_initial_anchor_query_f64.argtypes = [f64p, f64p, f64p, ci, ci, ci, ci, ci, ctypes.c_float,
                                                            ctypes.c_float, f64p, f64p]
        lib.epn_oracle_set_sq3_order.argtypes = [ci]
        lib.epn_oracle_opt_n_threads.argtypes = [ci]
        lib.epn_oracle_opt_n_threads.restype = ci
        _lib = lib
    return _lib


class sq3_order:
    """with sq3_order(k): the float FPS / ball query evaluate squared distances in one of the alternative orders of
    epn_oracle.c (0 = canonical).  For tests/test_fp_order.py only."""

    def __init__(self, order):
        self.order = int(order)

    def __enter__(self):
        _load().epn_oracle_set_sq3_order(self.order)

    def __exit__(self, *exc):
        _load().epn_oracle_set_sq3_order(0)


def _f32(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f64(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _is64(t):
    """The reference dispatches the index / gather kernels on float and double (AT_DISPATCH_FLOATING_TYPES)."""
    return t.dtype == torch.float64


def _i32(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def ball_query(new_xyz, xyz, radius, nsample):
    """(new_xyz f[b,3,m], xyz f[b,3,n], radius, nsample) -> int32 [b,m,nsample]."""
    lib = _load()
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    out = np.zeros((b, m, nsample), dtype=np.int32)
    if _is64(xyz):
        qa, qp = _f64(new_xyz)
        sa, sp = _f64(xyz)
        lib.epn_oracle_ball_query_f64(qp, sp, b, n, m, float(radius), int(nsample),
                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return torch.from_numpy(out)
    qa, qp = _f32(new_xyz)
    sa, sp = _f32(xyz)
    lib.epn_oracle_ball_query_f32(qp, sp, b, n, m, float(radius), int(nsample),
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return torch.from_numpy(out)


def furthest_point_sampling(xyz, m):
    """(xyz f[b,3,n], m) -> int32 [b,m]; temp = 1e10 as grouping_cuda.cpp:167-168."""
    lib = _load()
    b, _, n = xyz.shape
    out = np.zeros((b, m), dtype=np.int32)
    if _is64(xyz):
        xa, xp = _f64(xyz)
        temp = np.full((b, n), 1e10, dtype=np.float64)
        lib.epn_oracle_fps_f64(xp, b, n, int(m), temp.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return torch.from_numpy(out)
    xa, xp = _f32(xyz)
    temp = np.full((b, n), 1e10, dtype=np.float32)
    lib.epn_oracle_fps_f32(xp, b, n, int(m), temp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return torch.from_numpy(out)


def gather_points_forward(points, idx):
    """(points f[b,c,n], idx int32[b,m]) -> float32 [b,c,m]."""
    lib = _load()
    b, c, n = points.shape
    m = idx.shape[1]
    ia, ip = _i32(idx)
    if _is64(points):
        pa, pp = _f64(points)
        out = np.zeros((b, c, m), dtype=np.float64)
        lib.epn_oracle_gather_fwd_f64(pp, ip, b, c, n, m, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return torch.from_numpy(out)
    pa, pp = _f32(points)
    out = np.zeros((b, c, m), dtype=np.float32)
    lib.epn_oracle_gather_fwd_f32(pp, ip, b, c, n, m, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return torch.from_numpy(out)


def gather_points_backward(grad_out, idx, npoint):
    """(grad_out f[b,c,m], idx int32[b,m], npoint) -> f[b,c,npoint]."""
    lib = _load()
    b, c, m = grad_out.shape
    ia, ip = _i32(idx)
    if _is64(grad_out):
        ga, gp = _f64(grad_out)
        out = np.zeros((b, c, npoint), dtype=np.float64)
        lib.epn_oracle_gather_bwd_f64(gp, ip, b, c, int(npoint), m,
                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return torch.from_numpy(out)
    ga, gp = _f32(grad_out)
    out = np.zeros((b, c, npoint), dtype=np.float32)
    lib.epn_oracle_gather_bwd_f32(gp, ip, b, c, int(npoint), m,
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return torch.from_numpy(out)


def initial_anchor_query(centers, xyz, kernel_points, radius, sigma):
    """(centers f[b,3,nc], xyz f[m,3], kernel_points f[ks,na,3], radius, sigma) -> [weights, counts] f[b,ks,nc,na]
    (vgtk/vgtk/cuda/grouping_cuda.cpp:138-158)."""
    lib = _load()
    b, _, nc = centers.shape
    m = xyz.shape[0]
    ks, na = kernel_points.shape[0], kernel_points.shape[1]
    if _is64(xyz):                       # dispatch on xyz.type(), grouping_cuda_kernel.cu:558
        ca, cp = _f64(centers)
        xa, xp = _f64(xyz)
        ka, kpp = _f64(kernel_points)
        w = np.zeros((b, ks, nc, na), dtype=np.float64)
        c = np.zeros((b, ks, nc, na), dtype=np.float64)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.epn_oracle_initial_anchor_query_f64(cp, xp, kpp, b, nc, m, na, ks, float(radius), float(sigma),
                                                w.ctypes.data_as(f64p), c.ctypes.data_as(f64p))
        return [torch.from_numpy(w), torch.from_numpy(c)]
    ca, cp = _f32(centers)
    xa, xp = _f32(xyz)
    ka, kpp = _f32(kernel_points)
    w = np.zeros((b, ks, nc, na), dtype=np.float32)
    c = np.zeros((b, ks, nc, na), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)
    lib.epn_oracle_initial_anchor_query_f32(cp, xp, kpp, b, nc, m, na, ks, float(radius), float(sigma),
                                            w.ctypes.data_as(f32p), c.ctypes.data_as(f32p))
    return [torch.from_numpy(w), torch.from_numpy(c)]


def opt_n_threads(work_size):
    return _load().epn_oracle_opt_n_threads(int(work_size))


def anchor_query(sample_idx, grouped_indices, grouped_xyz, anchors, kernel_points, nq):
    """vgtk.cuda.grouping.anchor_query restated in numpy, float32 or float64 after grouped_xyz (grouping_cuda.cpp:88-108,
    kernel grouping_cuda_kernel.cu:180-247, dispatch :505-510): w[b,p,a,k,n] = (kw - norm)^2 + ((kh - theta) norm)^2
    with norm = |g| + 1e-6, theta = acos(g . anchor_a / norm).  `+ 1e-6` is a DOUBLE literal in the reference (:221), so
    the float instantiation forms the sum in double and rounds once.  Elementwise float arithmetic (sqrt, acos): the HIP
    kernel is compared within a few ulp, not bit-exactly.  sample_idx / grouped_indices / nq are unused by the reference
    kernel."""
    T = np.float64 if _is64(grouped_xyz) else np.float32
    g = grouped_xyz.detach().cpu().numpy().astype(T)                             # [b,3,p,nn]
    A = anchors.detach().cpu().numpy().astype(T)                                 # [na,3]
    K = kernel_points.detach().cpu().numpy().astype(T)                           # [ks,2]
    norm = (np.sqrt((g * g).sum(1)).astype(np.float64) + 1e-6).astype(T)         # [b,p,nn]
    dot = np.einsum('bcpn,ac->bpan', g, A).astype(T)                             # [b,p,na,nn]
    theta = np.arccos(np.clip(dot / norm[:, :, None, :], -1.0, 1.0)).astype(T)
    d0 = K[None, None, None, :, 0, None] - norm[:, :, None, None, :]             # [b,p,1,ks,nn]
    d1 = (K[None, None, None, :, 1, None] - theta[:, :, :, None, :]) * norm[:, :, None, None, :]
    return [torch.from_numpy((d0 * d0 + d1 * d1).astype(T))]
