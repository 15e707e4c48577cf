"""BasicSO3Conv's weight contraction (vgtk/vgtk/so3conv/modules.py:48-55) and its autograd transposes on the library's
own MFMA GEMM kernels (csrc/gemm.hip) -- no torch.mm / BLAS on the path.

    gemm_nt(A [M,K], Bt [N,K])  -> A @ Bt.T       activations x weights (fp32, or bf16 with fp32 accumulation)
    gemm_tn(X [R,N1], Y [R,N2]) -> X.T @ Y (fp32) weight gradients: deterministic split over R
    matmul_nt(A, W)             -> autograd Function over the two (dA = dC @ W as NT on W^T, dW = dC^T A as TN)
"""
import ctypes
import os

import torch

from . import _lib
from ._ab import ab


# fp32 contractions (fp32 operands, fp32 accumulation, fp32 result in every mode):
#   "f16x2"  two fp16 pieces per operand, three v_mfma_f32_32x32x16_f16 per block (csrc/gemm.h): measured error vs fp64 at or
#            below the fp32 matrix instruction's; needs max|operand| as a device scalar (`amax`, see absmax)
#   "split"  three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per block: no input bit is dropped (csrc/gemm_x3.hip)
#   "native" v_mfma_f32_32x32x2_f32
FP32_MODES = ("f16x2", "split", "native")
FP32_MODE = os.environ.get("EPN_GEMM_FP32", "f16x2")
if FP32_MODE not in FP32_MODES:
    raise ValueError(f"EPN_GEMM_FP32 must be one of {FP32_MODES}")


def set_fp32_mode(mode):
    global FP32_MODE
    if mode not in FP32_MODES:
        raise ValueError(f"fp32 GEMM mode is one of {FP32_MODES}")
    FP32_MODE = mode


def absmax(t):
    """max |t| of a 2-D row-major (possibly row-strided) or contiguous fp32 tensor as a 1-element device tensor: the scale
    source of the f16x2 GEMMs.  A producer that knows the maximum hands its own scalar to gemm_nt / gemm_tn instead."""
    lib = _lib.get_lib()
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError("absmax: fp32 CUDA tensor expected")
    out = torch.empty(1, dtype=torch.float32, device=t.device)
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    if dense:                              # any dense layout: the maximum does not care about the order
        rows, cols, ld = 1, t.numel(), t.numel()
    elif t.dim() == 2 and t.stride(1) == 1:
        rows, cols, ld = t.shape[0], t.shape[1], t.stride(0)
    else:
        t = t.contiguous()
        rows, cols, ld = 1, t.numel(), t.numel()
    _lib.check(lib.epn_absmax_f32(t.data_ptr(), ld, rows, cols, out.data_ptr(), _lib.stream_of(t)), "absmax")
    return out


def f16x2_on(t):
    """Do the fp32 contractions of this tensor run in the two-piece fp16 form (and so want max|operand| device scalars)?"""
    return FP32_MODE == "f16x2" and t.dtype == torch.float32


def amax_tag(t):
    """The producer-side (or earlier computed) max|t| of a tensor, or None -- never a pass over t."""
    for cand in (t, t._base if t._base is not None and t._base.numel() == t.numel() else None):
        hit = getattr(cand, "_epn_amax", None) if cand is not None else None
        if hit is not None and hit[0] == cand._version:
            return hit[1]
    return None


def absmax_cached(t):
    """absmax(t), remembered ON the tensor object (and dropped when the tensor is written to): a tensor that is an operand of
    several GEMMs -- an output gradient feeds the data-gradient and the weight-gradient contraction, a block input the
    grouping and the skip convolution -- is scanned once, and a producer that knows the maximum tags its output the same way
    (ops._tag_amax: basis change, block tail, norm backward).  A re-layout VIEW of a tagged tensor that covers all of it (what
    autograd's view nodes hand to a 1x1 convolution's backward) has the same maximum."""
    for cand in (t, t._base if t._base is not None and t._base.numel() == t.numel() else None):
        hit = getattr(cand, "_epn_amax", None) if cand is not None else None
        if hit is not None and hit[0] == cand._version:
            return hit[1]
    a = absmax(t)
    try:
        t._epn_amax = (t._version, a)
    except (AttributeError, RuntimeError):
        pass
    return a


def mark_written(t):
    """A tensor that already existed is about to be (or was just) written through its raw pointer by a library kernel -- an
    accumulating scatter onto an incoming gradient, a GEMM into a caller's `out`.  Raw-pointer writes do not touch torch's
    version counter, and the maxima remembered on tensors (`_epn_amax`) are keyed on it: a tagged tensor written this way would
    keep a stale maximum, and the two-piece scale leaves only a factor 2-4 of headroom (advisor finding, round 5).  Bumping the
    counter -- which the tensor shares with every view and detach() alias -- drops every such tag at once, and it is what an
    in-place torch op on the same memory would have done.  Returns t."""
    if t is not None:
        torch._C._increment_version((t,))       # (takes an ITERABLE of tensors: a bare tensor is walked row by row)
        for cand in (t, t._base):
            if cand is not None and getattr(cand, "_epn_amax", None) is not None:
                try:
                    del cand._epn_amax
                except AttributeError:
                    pass
    return t


def f16x2_overflow_count(reset=False, device=None):
    """epn_f16x2_overflow_count of `device` (default: the current one): waves of two-piece fp16 GEMMs that ended a tile with a
    non-finite accumulator since the counter was last cleared.  With finite operands any non-zero value means a reported
    maximum was too small (include/epn_so3conv.h).  Synchronises; not inside a stream capture."""
    lib = _lib.get_lib()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        torch.cuda.synchronize()
        n = int(lib.epn_f16x2_overflow_count(1 if reset else 0))
    if n < 0:
        raise RuntimeError(f"epn_f16x2_overflow_count failed: hipError {-n}")
    return n


def fixed_point_range_count(reset=False, device=None):
    """epn_inter_ungroup_cloud_range_count of `device`: workgroups of the cloud-resident transpose of the grouping that saw a
    contribution beyond the range its reported max|dG| allows (or a non-finite dG) since the counter was last cleared; their
    rows were written as NaN.  Synchronises; not inside a stream capture."""
    lib = _lib.get_lib()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        torch.cuda.synchronize()
        n = int(lib.epn_inter_ungroup_cloud_range_count(1 if reset else 0))
    if n < 0:
        raise RuntimeError(f"epn_inter_ungroup_cloud_range_count failed: hipError {-n}")
    return n


# Debug mode of the scale contract (EPN_AB=1 EPN_CHECK_AMAX=1, or gemm.CHECK_AMAX = True): every maximum a two-piece GEMM is
# about to consume is re-derived by a pass over its operand and compared on the host -- a synchronising, eager-only check that
# turns a stale / under-reported tag into an exception naming the call (tests/test_gpu_bf16.py runs a training step under it).
CHECK_AMAX = ab("EPN_CHECK_AMAX") == "1"


def _check_amax(t, a, what):
    if not CHECK_AMAX or a is None or t is None or t.dtype != torch.float32:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("EPN_CHECK_AMAX synchronises with the host: run eagerly (bench.py --no-graph)")
    true, given = float(absmax(t).item()), float(a.item())
    if not true <= given * (1.0 + 1e-6):
        raise AssertionError(f"{what}: reported max|x| = {given:.9g} but the operand holds {true:.9g} "
                             f"(stale or under-reported maximum: the two-piece scale would overflow at {4 * given:.3g})")


def _use_amax(a):
    """An amax scalar about to be read by a kernel on the current stream.  These 4-byte tensors are produced on one stream (a
    block's main branch) and read on another (its skip branch, forward and backward): without record_stream the caching
    allocator hands the block to the next 4-byte request the moment Python drops the tensor, and a later scalar overwrites it
    under a GEMM that has not read its scale yet (found as non-finite losses at small batch sizes)."""
    if a.dtype != torch.float32 or a.numel() != 1 or not a.is_cuda:
        raise ValueError("amax must be a 1-element fp32 CUDA tensor")
    a.record_stream(torch.cuda.current_stream(a.device))
    return a.data_ptr()


def _amax_array(amaxes):
    """list of 1-element device tensors / None -> (ctypes array of pointers or None, keep-alive list)."""
    if amaxes is None or all(a is None for a in amaxes):
        return None, []
    arr = (ctypes.c_void_p * len(amaxes))()
    for i, a in enumerate(amaxes):
        if a is not None:
            arr[i] = _use_amax(a)
    return arr, [a for a in amaxes if a is not None]


def _is_bf16(t):
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 0
    raise TypeError(f"GEMM operands must be float32 or bfloat16, got {t.dtype}")


def _rowmajor(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _problem(A, Bt, C):
    p = _lib.GemmNtProblem()
    p.A, p.Bt, p.C = A.data_ptr(), Bt.data_ptr(), C.data_ptr()
    p.M, p.N, p.K = A.shape[0], Bt.shape[0], A.shape[1]
    p.lda = A.stride(0) if A.shape[0] > 1 else A.shape[1]
    p.ldb = Bt.stride(0) if Bt.shape[0] > 1 else Bt.shape[1]
    p.ldc = C.stride(0) if C.shape[0] > 1 else C.shape[1]
    return p


def gemm_nt_grouped(problems, out_dtype=None, col_stats=None, a_amax=None, c_amax=None):
    """problems: list of (A [M,K], Bt [N,K], C [M,N] or None); one grouped launch per 6 problems.  Returns the C list.
    c_amax: optional list (one per problem) of ZEROED 1-element fp32 device tensors the kernels raise to max|C| from their
    accumulators (None entries: off).
    a_amax: optional list (one per problem) of 1-element device tensors holding max|A| (f16x2 mode; None = computed here).
    All operands share one dtype (fp32 or bf16); C is that dtype unless out_dtype says float32.  col_stats: optional list
    (one entry per problem) of fp32 [M/32, N, 2] tensors the kernels fill with per-column (sum, sum of squares) of every
    32-row block of C from their accumulators (None entries: off)."""
    lib = _lib.get_lib()
    arr = (_lib.GemmNtProblem * len(problems))()
    outs, keep = [], []
    bf = None
    for i, (A, Bt, C) in enumerate(problems):
        A, Bt = _rowmajor(A, "A"), _rowmajor(Bt, "Bt")
        if A.shape[1] != Bt.shape[1]:
            raise ValueError(f"gemm_nt: K mismatch {tuple(A.shape)} x {tuple(Bt.shape)}^T")
        b = _is_bf16(A)
        if _is_bf16(Bt) != b or (bf is not None and bf != b):
            raise TypeError("gemm_nt: mixed operand dtypes")
        bf = b
        odt = out_dtype or A.dtype
        if C is None:
            C = torch.empty((A.shape[0], Bt.shape[0]), dtype=odt, device=A.device)
        elif C.dtype != odt or C.stride(1) != 1 or tuple(C.shape) != (A.shape[0], Bt.shape[0]):
            raise ValueError("gemm_nt: output must be row-major [M,N] of the output dtype")
        else:
            mark_written(C)
        if a_amax is not None and not b and FP32_MODE == "f16x2":
            _check_amax(A, a_amax[i], f"gemm_nt problem {i} {tuple(A.shape)} x {tuple(Bt.shape)}^T, operand A")
        arr[i] = _problem(A, Bt, C)
        if col_stats is not None and col_stats[i] is not None:
            part = col_stats[i]
            if (part.dtype != torch.float32 or not part.is_contiguous() or A.shape[0] % 32
                    or part.numel() != A.shape[0] // 32 * Bt.shape[0] * 2):
                raise ValueError("gemm_nt: col_stats must be a contiguous fp32 [M/32, N, 2] tensor, M % 32 == 0")
            arr[i].col_stats = part.data_ptr()
        if c_amax is not None and c_amax[i] is not None:
            arr[i].c_amax = _use_amax(c_amax[i])
        outs.append(C)
        keep += [A, Bt]
    st = _lib.stream_of(outs[0])
    if bf:
        out_f32 = 1 if outs[0].dtype == torch.float32 else 0
        _lib.check(lib.epn_gemm_nt_bf16(len(problems), arr, out_f32, st), "gemm_nt_bf16")
    elif FP32_MODE == "f16x2":
        nbytes = int(lib.epn_gemm_nt_f16x2_workspace_bytes(len(problems), arr))
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=outs[0].device)
        parr, keep2 = _amax_array(a_amax)
        _lib.check(lib.epn_gemm_nt_f16x2_f32(len(problems), arr, parr, ws.data_ptr(), ws.numel(), st), "gemm_nt_f16x2_f32")
    elif FP32_MODE == "split":
        nbytes = int(lib.epn_gemm_nt_split_workspace_bytes(len(problems), arr))
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=outs[0].device)
        _lib.check(lib.epn_gemm_nt_split_f32(len(problems), arr, ws.data_ptr(), ws.numel(), st), "gemm_nt_split_f32")
    else:
        _lib.check(lib.epn_gemm_nt_f32(len(problems), arr, st), "gemm_nt_f32")
    return outs


def gemm_nt(A, Bt, out=None, out_dtype=None, col_stats=False, a_amax=None, c_amax=False):
    """col_stats=True: returns (C, partials [M/32, N, 2] or None when M % 32 != 0) -- the per-channel statistics of C
    from the kernel's epilogue (ops.sums_from_partials finishes them).  a_amax: device scalar max|A| (f16x2 mode).
    c_amax=True: returns (C, max|C| as a 1-element device tensor, from the kernel's epilogue; C is tagged with it)."""
    am = None if a_amax is None else [a_amax]
    if c_amax:
        cm = torch.zeros(1, dtype=torch.float32, device=A.device)
        C = gemm_nt_grouped([(A, Bt, out)], out_dtype, a_amax=am, c_amax=[cm])[0]
        try:
            C._epn_amax = (C._version, cm)
        except (AttributeError, RuntimeError):
            pass
        return C, cm
    if not col_stats:
        return gemm_nt_grouped([(A, Bt, out)], out_dtype, a_amax=am)[0]
    M, N = A.shape[0], Bt.shape[0]
    part = torch.empty((M // 32, N, 2), dtype=torch.float32, device=A.device) if M % 32 == 0 and M > 0 else None
    return gemm_nt_grouped([(A, Bt, out)], out_dtype, [part], a_amax=am)[0], part


def gemm_tn(X, Y, out=None, x_amax=None, y_amax=None, fp32_mode=None):
    """X [R,N1], Y [R,N2] (same dtype) -> X^T Y fp32 [N1,N2].  x_amax / y_amax: device scalars max|X|, max|Y| (f16x2 mode).
    fp32_mode: the form of THIS call for fp32 operands (default: the process-wide FP32_MODE) -- a bandwidth-bound contraction
    whose operands carry no maximum runs as fast in the lossless form, without the two passes."""
    lib = _lib.get_lib()
    X, Y = _rowmajor(X, "X"), _rowmajor(Y, "Y")
    if X.shape[0] != Y.shape[0]:
        raise ValueError(f"gemm_tn: row mismatch {tuple(X.shape)} vs {tuple(Y.shape)}")
    bf = _is_bf16(X)
    if _is_bf16(Y) != bf:
        raise TypeError("gemm_tn: mixed operand dtypes")
    R, N1, N2 = X.shape[0], X.shape[1], Y.shape[1]
    if out is None:
        out = torch.empty((N1, N2), dtype=torch.float32, device=X.device)
    elif out.dtype != torch.float32 or out.stride(1) != 1 or tuple(out.shape) != (N1, N2):
        raise ValueError("gemm_tn: output must be row-major fp32 [N1,N2]")
    else:
        mark_written(out)
    mode = fp32_mode or FP32_MODE
    split = not bf and mode == "split"
    f2 = not bf and mode == "f16x2"
    if f2:
        _check_amax(X, x_amax, f"gemm_tn {tuple(X.shape)}^T x {tuple(Y.shape)}, operand X")
        _check_amax(Y, y_amax, f"gemm_tn {tuple(X.shape)}^T x {tuple(Y.shape)}, operand Y")
    nbytes = int(lib.epn_gemm_tn_workspace_bytes(3 if f2 else (2 if split else bf), R, N1, N2))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=X.device)
    fn = lib.epn_gemm_tn_bf16 if bf else (lib.epn_gemm_tn_split_f32 if split else lib.epn_gemm_tn_f32)
    ldx = X.stride(0) if R > 1 else N1
    ldy = Y.stride(0) if R > 1 else N2
    ldc = out.stride(0) if N1 > 1 else N2
    if f2:
        _lib.check(lib.epn_gemm_tn_f16x2_f32(X.data_ptr(), ldx, Y.data_ptr(), ldy, out.data_ptr(), ldc, R, N1, N2,
                                             None if x_amax is None else _use_amax(x_amax),
                                             None if y_amax is None else _use_amax(y_amax), ws.data_ptr(), ws.numel(),
                                             _lib.stream_of(X)), "gemm_tn_f16x2")
        return out
    _lib.check(fn(X.data_ptr(), ldx, Y.data_ptr(), ldy, out.data_ptr(), ldc, R, N1, N2, ws.data_ptr(), ws.numel(),
                  _lib.stream_of(X)), "gemm_tn")
    return out


def gemm_tn_grouped(problems, outs_into=None, x_amax=None, y_amax=None):
    """problems: list of (X [R,N1], Y [R,N2]) of one dtype -> list of fp32 X^T Y, ONE launch (+ one reduction launch).
    outs_into: optional list of row-major fp32 [N1,N2] tensors to write the results into (e.g. slices of one buffer).
    x_amax / y_amax: optional lists of device scalars (f16x2 mode)."""
    lib = _lib.get_lib()
    arr = (_lib.GemmTnProblem * len(problems))()
    outs, keep, bf = [], [], None
    for i, (X, Y) in enumerate(problems):
        X, Y = _rowmajor(X, "X"), _rowmajor(Y, "Y")
        if X.shape[0] != Y.shape[0]:
            raise ValueError(f"gemm_tn: row mismatch {tuple(X.shape)} vs {tuple(Y.shape)}")
        b = _is_bf16(X)
        if _is_bf16(Y) != b or (bf is not None and bf != b):
            raise TypeError("gemm_tn: mixed operand dtypes")
        bf = b
        if outs_into is not None:
            C = outs_into[i]
            if C.dtype != torch.float32 or tuple(C.shape) != (X.shape[1], Y.shape[1]) or not C.is_contiguous():
                raise ValueError("gemm_tn_grouped: outputs must be contiguous fp32 [N1,N2]")
            mark_written(C)
        else:
            C = torch.empty((X.shape[1], Y.shape[1]), dtype=torch.float32, device=X.device)
        p = arr[i]
        p.X, p.Y, p.C = X.data_ptr(), Y.data_ptr(), C.data_ptr()
        p.R, p.N1, p.N2 = X.shape[0], X.shape[1], Y.shape[1]
        p.ldx = X.stride(0) if X.shape[0] > 1 else X.shape[1]
        p.ldy = Y.stride(0) if Y.shape[0] > 1 else Y.shape[1]
        p.ldc = C.shape[1]
        outs.append(C)
        keep += [X, Y]
    mode = bf if bf else {"split": 2, "f16x2": 3}.get(FP32_MODE, 0)
    nbytes = int(lib.epn_gemm_tn_grouped_workspace_bytes(mode, len(problems), arr))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=outs[0].device)
    if mode == 3:
        for i, (X, Y) in enumerate(problems):
            if x_amax is not None:
                _check_amax(_rowmajor(X, "X"), x_amax[i], f"gemm_tn_grouped problem {i}, operand X {tuple(X.shape)}")
            if y_amax is not None:
                _check_amax(_rowmajor(Y, "Y"), y_amax[i], f"gemm_tn_grouped problem {i}, operand Y {tuple(Y.shape)}")
        xa, k1 = _amax_array(x_amax)
        ya, k2 = _amax_array(y_amax)
        _lib.check(lib.epn_gemm_tn_grouped_f16x2(len(problems), arr, xa, ya, ws.data_ptr(), ws.numel(), _lib.stream_of(outs[0])),
                   "gemm_tn_grouped_f16x2")
        return outs
    _lib.check(lib.epn_gemm_tn_grouped(mode, len(problems), arr, ws.data_ptr(), ws.numel(), _lib.stream_of(outs[0])),
               "gemm_tn_grouped")
    return outs


def transpose_cast(W, dtype):
    """[r,c] -> [c,r] contiguous in `dtype` (fp32 / bf16) with the library's kernel (weights only)."""
    lib = _lib.get_lib()
    W = W.contiguous()
    out = torch.empty((W.shape[1], W.shape[0]), dtype=dtype, device=W.device)
    _lib.check(lib.epn_transpose_cast(W.data_ptr(), out.data_ptr(), W.shape[0], W.shape[1], _is_bf16(W), _is_bf16(out),
                                      _lib.stream_of(W)), "transpose_cast")
    return out


def cast(t, dtype):
    """fp32 <-> bf16 copy of a contiguous tensor with the library's kernel."""
    if t.dtype == dtype:
        return t
    lib = _lib.get_lib()
    t = t.contiguous()
    out = torch.empty_like(t, dtype=dtype)
    _lib.check(lib.epn_cast(t.data_ptr(), out.data_ptr(), t.numel(), _is_bf16(t), _is_bf16(out), _lib.stream_of(t)), "cast")
    return out


class MatmulNT(torch.autograd.Function):
    """C = A @ W^T with A [M,K] activations (fp32 or bf16) and W [N,K] fp32 master weights (cast per call for bf16).
    dA = dC @ W (NT against W^T), dW = dC^T @ A (TN, fp32)."""

    @staticmethod
    def forward(ctx, A, W, col_stats=False, a_amax=None):
        from . import ops
        ctx.set_materialize_grads(False)   # (C, part): no zero-filled gradient for the statistics partials
        Wc = W if W.dtype == A.dtype else cast(W, A.dtype)
        ctx.save_for_backward(A, W)
        M, K, N = A.shape[0], A.shape[1], W.shape[0]
        if f16x2_on(A):                   # max|A| (or a bound the caller has): shared with the weight gradient, where A is the Y operand
            a_amax = a_amax if a_amax is not None else absmax_cached(A)
        else:
            a_amax = None
        ctx.a_amax = a_amax
        if not col_stats:
            return ops._launch("conv1x1_gemm", ("nt", M, N, K), 2.0 * M * N * K, A.device, lambda: gemm_nt(A, Wc, a_amax=a_amax))
        # (C, partial column statistics of C from the kernel's epilogue; an empty tensor when M % 32 != 0)
        C, part = ops._launch("conv1x1_gemm", ("nt", M, N, K), 2.0 * M * N * K, A.device,
                              lambda: gemm_nt(A, Wc, col_stats=True, a_amax=a_amax))
        part = part if part is not None else torch.empty(0, dtype=torch.float32, device=A.device)
        ctx.mark_non_differentiable(part)
        return C, part

    @staticmethod
    def backward(ctx, dC, _dpart=None):
        if dC is None:
            return None, None, None, None
        A, W = ctx.saved_tensors
        dC = dC if dC.dtype == A.dtype else dC.to(A.dtype)
        from . import ops
        dA = dW = None
        M, K, N = A.shape[0], A.shape[1], W.shape[0]
        dc_amax = absmax_cached(dC) if f16x2_on(dC) else None       # (tagged by the block tail's backward that produced it)
        if ctx.needs_input_grad[0]:
            Wt = transpose_cast(W, A.dtype)
            dA = ops._launch("conv1x1_gemm", ("nt", M, K, N), 2.0 * M * N * K, A.device, lambda: gemm_nt(dC, Wt, a_amax=dc_amax))
            # a fresh buffer that is the gradient of exactly one tensor (A): a consumer that receives a re-layout VIEW of it
            # through autograd's view nodes may accumulate into it (ops.InterSO3ConvSplitFn._may_write_into)
            dA._epn_private = True
        if ctx.needs_input_grad[1]:
            dW = ops._launch("conv1x1_gemm_dw", ("tn", M, N, K), 2.0 * M * N * K, A.device,
                             lambda: gemm_tn(dC, A, x_amax=dc_amax, y_amax=ctx.a_amax))
        return dA, dW, None, None


def matmul_nt(A, W, col_stats=False, a_amax=None):
    return MatmulNT.apply(A, W, col_stats, a_amax)
