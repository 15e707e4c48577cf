"""`vgtk.cuda.grouping` -- same function names/signatures as vgtk/vgtk/cuda/grouping_cuda.cpp:176-181."""
import torch

from ... import _lib


def ball_query(new_xyz, xyz, radius, nsample):
    """(new_xyz f[b,3,m], xyz f[b,3,n], float radius, int nsample) -> int32 [b,m,nsample]; float32 or float64 coordinates
    (grouping_cuda.cpp:71-86; AT_DISPATCH_FLOATING_TYPES, grouping_cuda_kernel.cu:477)."""
    lib = _lib.get_lib()
    dt = _lib.float_dtype(xyz, "xyz")
    q, s = _lib.dev_ptr(new_xyz, "new_xyz", dt), _lib.dev_ptr(xyz, "xyz", dt)
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=xyz.device)
    fn = lib.epn_ball_query_f64 if dt == torch.float64 else lib.epn_ball_query_f32
    _lib.check(fn(q, s, b, n, m, float(radius), int(nsample), _lib.dev_ptr(idx, "idx", torch.int32), _lib.stream_of(xyz)),
               "ball_query")
    return idx


def furthest_point_sampling(source_xyz, m):
    """(xyz f[b,3,n], int m) -> int32 [b,m]; float32 or float64 coordinates (grouping_cuda.cpp:160-174; the fp64
    instantiation keeps its running minima in a temp tensor like the reference, :167-168)."""
    lib = _lib.get_lib()
    dt = _lib.float_dtype(source_xyz, "source_xyz")
    p = _lib.dev_ptr(source_xyz, "source_xyz", dt)
    b, _, n = source_xyz.shape
    idx = torch.empty((b, int(m)), dtype=torch.int32, device=source_xyz.device)
    if dt == torch.float64:
        temp = torch.empty((b, n), dtype=torch.float64, device=source_xyz.device)
        _lib.check(lib.epn_fps_f64(p, b, n, int(m), _lib.dev_ptr(temp, "temp", dt),
                                   _lib.dev_ptr(idx, "sampled_idx", torch.int32), _lib.stream_of(source_xyz)),
                   "furthest_point_sampling")
        return idx
    if n > 32768:        # beyond the register-resident kernels: the reference's own structure, minima in `temp` (:167-168)
        temp = torch.empty((b, n), dtype=torch.float32, device=source_xyz.device)
        _lib.check(lib.epn_fps_temp_f32(p, b, n, int(m), _lib.dev_ptr(temp, "temp", dt),
                                        _lib.dev_ptr(idx, "sampled_idx", torch.int32), _lib.stream_of(source_xyz)),
                   "furthest_point_sampling")
        return idx
    _lib.check(lib.epn_fps_f32(p, b, n, int(m), _lib.dev_ptr(idx, "sampled_idx", torch.int32),
                               _lib.stream_of(source_xyz)), "furthest_point_sampling")
    return idx


def initial_anchor_query(centers, xyz, kernel_points, radius, sigma):
    """(centers f[b,3,nc], xyz f[m,3], kernel_points f[ks,na,3], radius, sigma) ->
    [anchor_weights f[b,ks,nc,na], anchor_ctn f[b,ks,nc,na]]  (grouping_cuda.cpp:138-158; KernelPropagation).  float32 or
    float64 after xyz, outputs in the same dtype (dispatch grouping_cuda_kernel.cu:558-563, outputs :149-154)."""
    lib = _lib.get_lib()
    dt = _lib.float_dtype(xyz, "xyz")
    c, x, k = (_lib.dev_ptr(centers, "centers", dt), _lib.dev_ptr(xyz, "xyz", dt),
               _lib.dev_ptr(kernel_points, "kernel_points", dt))
    b, _, nc = centers.shape
    m = xyz.shape[0]
    ks, na = kernel_points.shape[0], kernel_points.shape[1]
    wts = torch.empty((b, ks, nc, na), dtype=dt, device=xyz.device)
    ctn = torch.empty((b, ks, nc, na), dtype=dt, device=xyz.device)
    fn = lib.epn_initial_anchor_query_f64 if dt == torch.float64 else lib.epn_initial_anchor_query_f32
    _lib.check(fn(c, x, k, b, nc, m, na, ks, float(radius), float(sigma), _lib.dev_ptr(wts, "anchor_weights", dt),
                  _lib.dev_ptr(ctn, "anchor_ctn", dt), _lib.stream_of(xyz)), "initial_anchor_query")
    return [wts, ctn]


def anchor_query(sample_idx, grouped_indices, grouped_xyz, anchors, kernel_points, nq):
    """(sample_idx i[b,p], grouped_indices i[b,p,nn], grouped_xyz f[b,3,p,nn], anchors f[na,3], kernel_points f[ks,2],
    int nq) -> [anchor_weights f[b,p,na,ks,nn]]  (grouping_cuda.cpp:88-108; legacy ZPConv).  float32 or float64 after
    grouped_xyz (dispatch grouping_cuda_kernel.cu:505-510).  sample_idx, grouped_indices and nq are checked like the
    reference does (CHECK_INPUT) and otherwise unused, as in its kernel."""
    lib = _lib.get_lib()
    _lib.dev_ptr(sample_idx, "sample_idx", torch.int32)
    _lib.dev_ptr(grouped_indices, "grouped_indices", torch.int32)
    dt = _lib.float_dtype(grouped_xyz, "grouped_xyz")
    g, a, k = (_lib.dev_ptr(grouped_xyz, "grouped_xyz", dt), _lib.dev_ptr(anchors, "anchors", dt),
               _lib.dev_ptr(kernel_points, "kernel_points", dt))
    b, _, p, nn = grouped_xyz.shape
    na, ks = anchors.shape[0], kernel_points.shape[0]
    w = torch.empty((b, p, na, ks, nn), dtype=dt, device=grouped_xyz.device)
    fn = lib.epn_anchor_query_f64 if dt == torch.float64 else lib.epn_anchor_query_f32
    _lib.check(fn(g, a, k, b, p, nn, na, ks, _lib.dev_ptr(w, "anchor_weights", dt), _lib.stream_of(grouped_xyz)),
               "anchor_query")
    return [w]
