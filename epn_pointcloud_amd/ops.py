"""Autograd Functions over the C ABI (include/epn_so3conv.h) for the fused SO(3) convolutions.

Public feature tensors keep the reference's logical shape [b, c, p, a]; physically they are kept
channels-last ([b][p][a][c], torch.channels_last), which is what the HIP kernels consume.  A tensor in
any other layout is converted once on entry.
"""
import ctypes
import os
import weakref

import torch

from ._ab import ab

from . import _lib, gemm


# ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream ------------
# Every launching C-ABI call is bracketed (the library proxy's CALL_HOOK); calls made inside _launch(...) are timed as ONE
# record carrying the caller's algorithmic flops.  The device kernel of a record is what the library itself reports
# (epn_last_kernel: the exact template instance the launcher picked), never a guess made here.
_PROFILE = None
_DEPTH = 0


def _last_kernel():
    return _lib.get_lib().epn_last_kernel().decode()


def _hook(name, fn, args):
    if _PROFILE is None or _DEPTH > 0:
        return fn(*args)
    st = torch.cuda.current_stream()            # stream_of() made the tensors' device current
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    rc = fn(*args)
    e1.record(st)
    _PROFILE.append((name[4:], (), 0.0, e0, e1, _last_kernel()))
    return rc


def profile_begin():
    global _PROFILE
    _PROFILE = []
    _lib.get_lib().epn_last_kernel()            # clear
    _lib.CALL_HOOK = _hook


def profile_end():
    """-> list of (kind, shape_key, algorithmic_flops, start_event, end_event, device_kernel_name)"""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    _lib.CALL_HOOK = None
    return rec


def _launch(kind, key, flops, device, fn):
    global _DEPTH
    if _PROFILE is None or _DEPTH > 0:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(device)   # the stream the C ABI launches on (stream_of)
    _lib.get_lib().epn_last_kernel()         # clear
    e0.record(st)
    _DEPTH += 1
    try:
        rc = fn()
    finally:
        _DEPTH -= 1
    e1.record(st)
    _PROFILE.append((kind, key, flops, e0, e1, _last_kernel()))
    return rc


def _inter_flops(d):
    cols = float(d.b) * d.p2 * d.na
    return 9.0 * cols * d.ks * d.nn + 2.0 * cols * d.cin * d.ks * d.nn + 2.0 * cols * d.cout * d.cin * d.ks


def _inter_key(d):
    return (d.b, d.p1, d.p2, d.nn, d.na, d.ks, d.cin, d.cout)


FEATURE_DTYPES = (torch.float32, torch.bfloat16)


def to_cl(t, name="feats"):
    """Logical [b,c,p,a] float32 / bfloat16 device tensor -> channels-last contiguous (no copy if already so)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype not in FEATURE_DTYPES:
        raise TypeError(f"{name} must be float32 or bfloat16, got {t.dtype}")
    if t.dim() != 4:
        raise ValueError(f"{name} must be [b,c,p,a]")
    return t.contiguous(memory_format=torch.channels_last)


def _cl_ptr(t):
    assert t.is_contiguous(memory_format=torch.channels_last) or t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def empty_cl(b, c, p, a, device, dtype=torch.float32):
    return torch.empty((b, c, p, a), dtype=dtype, device=device, memory_format=torch.channels_last)


def _entry(lib, base, dtype):
    """C entry point of `base` for a feature dtype: epn_<base>_f32 | epn_<base>_bf16; the fp32 change of basis has a
    split form (bf16 matrix pipe, fp32 accuracy) that follows the GEMMs' switch (gemm.FP32_MODE)."""
    if dtype != torch.bfloat16 and base in ("so3_basis", "so3_basis_norm", "so3_basis_stats", "so3_basis_dstats") and gemm.FP32_MODE != "native":
        return getattr(lib, f"epn_{base}_split_f32")
    return getattr(lib, f"epn_{base}_{'bf16' if dtype == torch.bfloat16 else 'f32'}")


class CastFn(torch.autograd.Function):
    """fp32 <-> bf16 copy on the library's cast kernel (the seams of the bf16 feature path: after the fp32 first layer,
    before the fp32 PointnetSO3Conv / heads); the gradient is cast back."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dim() == 4:
            x = to_cl(x)
            out = torch.empty_like(x, dtype=dtype)      # preserves channels-last strides
            _lib.check(_lib.get_lib().epn_cast(x.data_ptr(), out.data_ptr(), x.numel(), int(x.dtype == torch.bfloat16),
                                               int(dtype == torch.bfloat16), _lib.stream_of(x)), "cast")
            return out
        return gemm.cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, g):
        return CastFn.apply(g, ctx.src), None


def cast_feats(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


class InterGeometry:
    """Lazy stand-in for the reference's materialised `inter_w` [b,p2,na,ks,nn] (3 GB at B=32 for the
    first ModelNet layer).  It carries what the fused kernels need to regenerate the weights on the fly
    (vgtk/vgtk/so3conv/functional.py:180-218); `dense()` materialises the reference tensor on demand so
    `conv(x, inter_idx, inter_w)` round-trips."""

    def __init__(self, xyz, new_xyz, ball_idx, anchors, kernels, sigma):
        self.xyz, self.new_xyz, self.ball_idx = xyz, new_xyz, ball_idx
        self.anchors, self.kernels, self.sigma = anchors.contiguous(), kernels.contiguous(), float(sigma)
        self._dense = None
        self._inverse = None

    @property
    def shape(self):
        b, p2, nn = self.ball_idx.shape
        return torch.Size((b, p2, self.anchors.shape[0], self.kernels.shape[0], nn))

    @property
    def device(self):
        return self.ball_idx.device

    def desc(self, cin, cout, dense_w=None):
        b, p2, nn = self.ball_idx.shape
        d = _lib.InterDesc()
        d.xyz = _lib.dev_ptr(self.xyz, "xyz")
        d.new_xyz = _lib.dev_ptr(self.new_xyz, "new_xyz")
        d.ball_idx = _lib.dev_ptr(self.ball_idx, "inter_idx", torch.int32)
        d.anchors = _lib.dev_ptr(self.anchors, "anchors")
        d.kernels = _lib.dev_ptr(self.kernels, "kernels")
        d.dense_w = _lib.dev_ptr(dense_w, "inter_w") if dense_w is not None else None
        d.sigma = self.sigma
        d.b, d.p1, d.p2, d.nn = b, self.xyz.shape[2], p2, nn
        d.na, d.ks, d.cin, d.cout = self.anchors.shape[0], self.kernels.shape[0], int(cin), int(cout)
        return d

    def inverse_list(self):
        """CSR inverse of the ball query (offsets [b, p1+1], entries [b, p2*nn], entries of a destination in increasing
        (p, n) order) for the deterministic data gradient; built once per geometry by epn_inter_inverse_list."""
        if self._inverse is None:
            lib = _lib.get_lib()
            b, p2, nn = self.ball_idx.shape
            p1 = self.xyz.shape[2]
            off = torch.empty((b, p1 + 1), dtype=torch.int32, device=self.device)
            ent = torch.empty((b, p2 * nn), dtype=torch.int32, device=self.device)
            _lib.check(lib.epn_inter_inverse_list(_lib.dev_ptr(self.ball_idx, "inter_idx", torch.int32), b, p1, p2, nn,
                                                  _lib.dev_ptr(off, "offsets", torch.int32),
                                                  _lib.dev_ptr(ent, "entries", torch.int32), _lib.stream_of(off)),
                       "inter_inverse_list")
            self._inverse = (off, ent)
        return self._inverse

    def dense(self):
        """Materialise w[b,p2,na,ks,nn] with the HIP kernel (API compatibility only)."""
        if self._dense is None:
            lib = _lib.get_lib()
            w = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.device)
            d = self.desc(1, 1)
            _lib.check(lib.epn_inter_weights_f32(ctypes.byref(d), _lib.dev_ptr(w, "w"), _lib.stream_of(w)),
                       "inter_weights")
            self._dense = w
        return self._dense


class DenseInterWeights:
    """User-supplied dense inter_w + inter_idx (the reference's reuse path, functional.py:170-172)."""

    def __init__(self, ball_idx, inter_w, p1):
        self.ball_idx, self.w, self.p1 = ball_idx, inter_w.contiguous(), int(p1)

    def desc(self, cin, cout, dense_w=None):
        b, p2, na, ks, nn = self.w.shape
        d = _lib.InterDesc()
        d.ball_idx = _lib.dev_ptr(self.ball_idx, "inter_idx", torch.int32)
        d.dense_w = _lib.dev_ptr(self.w, "inter_w")
        d.sigma = 1.0
        d.b, d.p1, d.p2, d.nn, d.na, d.ks, d.cin, d.cout = b, self.p1, p2, nn, na, ks, int(cin), int(cout)
        return d


def _workspace(lib, d, device):
    nbytes = lib.epn_inter_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return ws, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())


class InterSO3ConvFn(torch.autograd.Function):
    """out[b,o,p,a] = sum_{c,k} W[o,c*ks+k] sum_n feats[b,c,idx[b,p,n],a] w[b,p,a,k,n]
    (InterSO3Conv.forward, vgtk/vgtk/so3conv/modules.py:157-174), fused; backward = SURVEY a17."""

    @staticmethod
    def forward(ctx, feats, W, geo):
        lib = _lib.get_lib()
        f = to_cl(feats)
        Wc = W.contiguous()
        cout, ck = Wc.shape
        cin = f.shape[1]
        d = geo.desc(cin, cout)
        if ck != cin * d.ks or f.shape[2] != d.p1 or f.shape[3] != d.na or f.shape[0] != d.b:
            raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, W {tuple(Wc.shape)}, geometry "
                             f"b={d.b} p1={d.p1} na={d.na} ks={d.ks}")
        out = empty_cl(d.b, cout, d.p2, d.na, f.device)
        ws, wsp, wsn = _workspace(lib, d, f.device)
        grouped = None
        if cin == 1 and lib.epn_inter_c1_ok(ctypes.byref(d)) and ab("EPN_C1_SAVE") == "1":
            # first layer: keep the 24 grouped values per column for the weight gradient (96 B per column) instead of
            # regenerating the ks x nn weights there -- only when a weight gradient can be asked for
            if ctx.needs_input_grad[1]:
                grouped = torch.empty((d.b * d.p2 * d.na, d.ks), dtype=torch.float32, device=f.device)
            import contextlib
            with (_lib.kernel_policy(2) if ab("EPN_C1_MFMA") == "0" else contextlib.nullcontext()):   # A/B: the VALU kernel
                _lib.check(_launch("inter_fwd", _inter_key(d), _inter_flops(d), f.device,
                                   lambda: lib.epn_inter_so3conv_fwd_c1_f32(ctypes.byref(d), _cl_ptr(f), _lib.dev_ptr(Wc, "W"),
                                                                            _cl_ptr(out), grouped.data_ptr() if grouped is not None
                                                                            else None, wsp, wsn, _lib.stream_of(f))),
                           "inter_so3conv_fwd_c1")
        else:
            _lib.check(_launch("inter_fwd", _inter_key(d), _inter_flops(d), f.device,
                               lambda: lib.epn_inter_so3conv_fwd_f32(ctypes.byref(d), _cl_ptr(f), _lib.dev_ptr(Wc, "W"),
                                                                     _cl_ptr(out), wsp, wsn, _lib.stream_of(f))),
                       "inter_so3conv_fwd")
        # (the saved grouped values go through save_for_backward like any saved activation: saved-tensor hooks, version
        # checks, release with the graph -- advisor finding, round 4)
        if grouped is not None:
            ctx.save_for_backward(f, Wc, grouped)
        else:
            ctx.save_for_backward(f, Wc)
        ctx.geo, ctx.has_grouped = geo, grouped is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.get_lib()
        if ctx.has_grouped:
            f, Wc, grouped = ctx.saved_tensors
        else:
            (f, Wc), grouped = ctx.saved_tensors, None
        geo = ctx.geo
        g = to_cl(grad_out, "grad_out")
        cout = Wc.shape[0]
        cin = f.shape[1]
        d = geo.desc(cin, cout)
        ws, wsp, wsn = _workspace(lib, d, f.device)
        gf = gW = None
        if ctx.needs_input_grad[0]:
            gf = empty_cl(d.b, cin, d.p1, d.na, f.device)
            _lib.check(_launch("inter_bwd_data", _inter_key(d), _inter_flops(d), f.device,
                               lambda: lib.epn_inter_so3conv_bwd_data_f32(ctypes.byref(d), _cl_ptr(g),
                                                                          _lib.dev_ptr(Wc, "W"), _cl_ptr(gf), wsp,
                                                                          wsn, _lib.stream_of(f))),
                       "inter_so3conv_bwd_data")
        if ctx.needs_input_grad[1] and grouped is not None:
            gW = torch.empty_like(Wc)
            fl = 2.0 * d.b * d.p2 * d.na * cout * d.ks
            if ab("EPN_C1_DW") == "gemm" and g.dtype == torch.float32:
                # dW[o][k] = sum_col dOut[col][o] G[col][k] is a tall-skinny weight-gradient GEMM (2e6 x 32 x 24): the
                # library's TN kernels stream it at 3-4.5 TB/s, the dedicated kernel (one unpipelined stage per
                # workgroup, 1536 atomics each) ran at 0.7 TB/s
                # two-piece fp16 mode: the maxima without passes over the operands -- max|dOut| from its producer's tag (the
                # norm backward), |grouped| <= K max|feats| (0 <= w <= 1; the features are 32 x 1024 x 60 values): the two
                # passes cost 0.22 ms beside a 0.07 ms GEMM
                xa = ya = form = None
                if gemm.f16x2_on(g):
                    xa = gemm.amax_tag(g)
                    if xa is None:          # bf16 networks hand in an fp32 COPY of the gradient: no tag -- the lossless form
                        form = "split"      # needs no maximum and this contraction is bound by its operand stream anyway
                    else:
                        ya = gemm.absmax_cached(f) * float(d.nn)
                _launch("inter_bwd_weight_c1", _inter_key(d), fl, f.device,
                        lambda: gemm.gemm_tn(g.permute(0, 2, 3, 1).reshape(-1, cout), grouped, out=gW, x_amax=xa, y_amax=ya,
                                             fp32_mode=form))
            else:
                _lib.check(_launch("inter_bwd_weight_c1", _inter_key(d), fl, f.device,
                                   lambda: lib.epn_inter_so3conv_bwd_weight_c1_f32(ctypes.byref(d), grouped.data_ptr(),
                                                                                   _cl_ptr(g), _lib.dev_ptr(gW, "grad_W"),
                                                                                   _lib.stream_of(f))),
                           "inter_so3conv_bwd_weight_c1")
        elif ctx.needs_input_grad[1]:
            gW = torch.empty_like(Wc)
            _lib.check(_launch("inter_bwd_weight", _inter_key(d), _inter_flops(d), f.device,
                               lambda: lib.epn_inter_so3conv_bwd_weight_f32(ctypes.byref(d), _cl_ptr(f), _cl_ptr(g),
                                                                            _lib.dev_ptr(gW, "grad_W"), wsp, wsn,
                                                                            _lib.stream_of(f))),
                       "inter_so3conv_bwd_weight")
        return gf, gW, None


def _group_workspace(lib, d, device):
    nbytes = lib.epn_inter_group_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return ws, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())


class InterGroupFn(torch.autograd.Function):
    """inter_so3conv_grouping's feature part (vgtk/vgtk/so3conv/functional.py:118-140 ->
    inter_zpconv_grouping_naive, vgtk/vgtk/spconv/functional.py:372-421) as a tensor: [b, c, p1, a] features ->
    grouped [b*p2*na, c*ks] (the reference's [b, c, ks, p2, na] with the (c, ks) axes last); backward = the transpose
    (epn_inter_ungroup_f32)."""

    @staticmethod
    def forward(ctx, feats, geo):
        lib = _lib.get_lib()
        f = to_cl(feats)
        cin = f.shape[1]
        d = geo.desc(cin, 16)
        if f.shape[2] != d.p1 or f.shape[3] != d.na or f.shape[0] != d.b:
            raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, geometry b={d.b} p1={d.p1} na={d.na}")
        cols = d.b * d.p2 * d.na
        G = torch.empty((cols, cin * d.ks), dtype=f.dtype, device=f.device)
        ws, wsp, wsn = _group_workspace(lib, d, f.device)
        _lib.check(_entry(lib, "inter_group", f.dtype)(ctypes.byref(d), _cl_ptr(f), ctypes.c_void_p(G.data_ptr()), wsp,
                                                       wsn, _lib.stream_of(f)), "inter_group")
        ctx.geo, ctx.cin = geo, cin
        return G

    @staticmethod
    def backward(ctx, dG):
        lib = _lib.get_lib()
        d = ctx.geo.desc(ctx.cin, 16)
        dG = dG.contiguous()
        gf = empty_cl(d.b, ctx.cin, d.p1, d.na, dG.device)          # the scatter target is fp32 for either dtype
        ws, wsp, wsn = _group_workspace(lib, d, dG.device)
        _lib.check(_entry(lib, "inter_ungroup", dG.dtype)(ctypes.byref(d), ctypes.c_void_p(dG.data_ptr()), _cl_ptr(gf),
                                                         wsp, wsn, _lib.stream_of(dG)), "inter_ungroup")
        return cast_feats(gf, dG.dtype), None


def inter_group(feats, geo):
    return InterGroupFn.apply(feats, geo)


class InterSO3ConvSplitFn(torch.autograd.Function):
    """The same convolution as InterSO3ConvFn in the reference's own two steps -- inter_so3conv_grouping, then
    BasicSO3Conv's matmul (vgtk/vgtk/so3conv/modules.py:38-52,157-174) -- with the grouping as ONE HIP kernel that
    writes only the grouped features G[col][cin*ks] (no inter_w, no gathered neighbours) and the three weight
    contractions (out = G W^T, dW = dOut^T G, dG = dOut W) on this library's own MFMA GEMM kernels (csrc/gemm.hip,
    csrc/gemm_x3.hip: fp32 operands in the lossless 3 x bf16 split form by default, 160-200 fp32-equivalent TFLOP/s on the
    schedule's shapes; no BLAS library is involved).  G (cin*ks*4 bytes per column) is kept for the backward pass: the
    training-time choice on a 288 GB part; InterSO3ConvFn / InterSO3ConvOnChipFn are the forms that never write it."""

    @staticmethod
    def forward(ctx, feats, W, geo, share_input=False):
        """share_input: also return `feats` itself as a second output.  A caller that feeds the same tensor to another
        branch (the skip path of a SeparableSO3ConvBlock) uses THAT output there: the other branch's gradient then arrives
        here, and the transpose of the grouping accumulates onto it (epn_inter_ungroup_acc_*) -- no zero-fill of the
        scatter target and no separate addition pass over the two [b, cin, p1, na] gradients.  That accumulation writes INTO
        the incoming gradient tensor, which autograd only tolerates for a buffer nobody else can see: backward() checks what
        it can (not a view; the shared output neither retains its gradient nor carries hooks -- `shared_ref`, set by
        ops.inter_so3conv) and otherwise adds out of place.  What it cannot see is a consumer whose backward hands ONE tensor
        to two inputs (`shared + other`): callers of share_input=True promise a consumer with a private gradient buffer, as
        this library's row gather / 1x1 convolution are (EPN_SHARE_INPUT_GRAD=0 turns the fold off).
        share_input="stats": only the epilogue statistics are wanted -> returns (out, part)."""
        lib = _lib.get_lib()
        ctx.set_materialize_grads(False)
        ctx.share_input = share_input
        f = to_cl(feats)
        Wc = W.contiguous()
        cout, ck = Wc.shape
        cin = f.shape[1]
        d = geo.desc(cin, cout)
        if ck != cin * d.ks or f.shape[2] != d.p1 or f.shape[3] != d.na or f.shape[0] != d.b:
            raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, W {tuple(Wc.shape)}, geometry "
                             f"b={d.b} p1={d.p1} na={d.na} ks={d.ks}")
        cols = d.b * d.p2 * d.na
        G = torch.empty((cols, ck), dtype=f.dtype, device=f.device)
        ws, wsp, wsn = _group_workspace(lib, d, f.device)
        gflops = 9.0 * cols * d.ks * d.nn + 2.0 * cols * cin * d.ks * d.nn
        # packed column order of G (contiguous stores in the grouping kernel, include/epn_so3conv.h): W's columns follow
        packed = group_packed() and Wc.dtype == torch.float32 and bool(lib.epn_inter_group_packed_ok(ctypes.byref(d)))
        grp = _entry(lib, "inter_group_packed" if packed else "inter_group", f.dtype)
        _lib.check(_launch("inter_group", _inter_key(d), gflops, f.device,
                           lambda: grp(ctypes.byref(d), _cl_ptr(f), ctypes.c_void_p(G.data_ptr()), wsp, wsn,
                                       _lib.stream_of(f))), "inter_group")
        if packed:
            Wd = torch.empty((cout, ck), dtype=f.dtype, device=f.device)
            _lib.check(_entry(lib, "inter_pack_weights", f.dtype)(_lib.dev_ptr(Wc, "W"), cout, cin, d.ks, Wd.data_ptr(),
                                                                  _lib.stream_of(f)), "inter_pack_weights")
        else:
            Wd = gemm.cast(Wc, f.dtype)                              # fp32 master weights; bf16 copy per call
        # two-piece fp16 contractions want max|G| as a device scalar.  Reading the 3-6 GB of G back for it would cost what the
        # form saves; |G[col][c, k]| = |sum_n w[k][n] F[idx[n], a, c]| <= K max|F| (0 <= w <= 1) is a bound from a pass over the
        # 24 x smaller feature tensor -- an over-estimated maximum only narrows the window of full relative precision from
        # 2^17 to 2^17 / K below the true maximum (tests/test_gpu_bf16.py::test_f16x2_gemm_has_fp32_accuracy pins 64 x)
        g_amax = gemm.absmax_cached(f) * float(d.nn) if gemm.f16x2_on(G) else None
        if share_input:        # a block asks: its norm follows -- per-channel statistics from the GEMM's epilogue
            out2d, part = _launch("inter_gemm", _inter_key(d), 2.0 * cols * cout * ck, f.device,
                                  lambda: gemm.gemm_nt(G, Wd, col_stats=True, a_amax=g_amax))
            part = part if part is not None else torch.empty(0, dtype=torch.float32, device=f.device)
        else:
            out2d = _launch("inter_gemm", _inter_key(d), 2.0 * cols * cout * ck, f.device,
                            lambda: gemm.gemm_nt(G, Wd, a_amax=g_amax))
        ctx.g_amax = g_amax
        ctx.save_for_backward(G, Wc)
        ctx.geo, ctx.cin, ctx.packed = geo, cin, packed
        out = out2d.view(d.b, d.p2, d.na, cout).permute(0, 3, 1, 2)
        if share_input == "stats":
            ctx.mark_non_differentiable(part)
            return out, part
        if share_input:
            ctx.mark_non_differentiable(part)
            return out, feats, part
        return out

    # Tensor handles alive on a gradient that ONLY this backward can see: the engine's argument list + the Python wrapper.
    # A consumer whose backward hands ONE tensor to two inputs (`shared + other`) leaves a third handle in the other input's
    # buffer until that node has run -- writing in place would corrupt its gradient (advisor finding, round 4).
    # That baseline is a detail of the autograd engine of the torch that is running (2 in torch 2.10), so it is MEASURED once
    # per process instead of assumed (advisor finding, round 5: one handle fewer in another version would accept a tensor with
    # a real second owner): a probe Function receives a gradient nobody else holds and records its handle count; a second
    # probe receives one that IS shared with another node's input buffer and must see more.  No usable measurement (no
    # counter, or the two probes do not separate) -> never in place.
    _SOLE_OWNER_HANDLES = None

    @staticmethod
    def _calibrate_sole_owner():
        seen = {}

        class _Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, key):
                ctx.key = key
                return x.view_as(x)

            @staticmethod
            def backward(ctx, g):
                seen[ctx.key] = g._use_count()
                return g, None
        try:
            with torch.enable_grad():
                a = torch.zeros(4, requires_grad=True)
                (_Probe.apply(a, "sole") * 2.0).sum().backward()               # mul's backward hands over a fresh tensor
                b = torch.zeros(4, requires_grad=True)
                # add's backward hands ONE tensor to both inputs: the probe's gradient has a second owner (c's buffer) while it runs
                c = torch.zeros(4, requires_grad=True)
                (_Probe.apply(b, "shared") + c).backward(torch.ones(4))
            sole, shared = seen.get("sole"), seen.get("shared")
            return sole if sole is not None and shared is not None and shared > sole else 0
        except Exception:                                         # no _use_count, or an engine that works differently
            return 0

    @staticmethod
    def _sole_owner(t):
        cls = InterSO3ConvSplitFn
        if cls._SOLE_OWNER_HANDLES is None:
            cls._SOLE_OWNER_HANDLES = cls._calibrate_sole_owner()
        try:
            return cls._SOLE_OWNER_HANDLES > 0 and t._use_count() <= cls._SOLE_OWNER_HANDLES
        except AttributeError:                                  # a torch without the counter: never in place
            return False

    @staticmethod
    def _may_write_into(ctx, grad_shared):
        """May the scatter accumulate in place into the incoming gradient of the shared output?  (see forward)  Only when
        nobody else can observe that memory: the tensor (and, for the one accepted view, its base) has no other live handle,
        no retained gradient and no hook."""
        own = InterSO3ConvSplitFn._sole_owner
        if not own(grad_shared):
            return False
        base = grad_shared._base
        if base is not None:
            # a view: the base belongs to somebody else -- unless it is the fresh data gradient of the skip branch's 1x1
            # convolution (gemm.MatmulNT.backward marks it), re-laid-out by autograd's view nodes ([rows, c] -> [b, c, p, a]):
            # the whole buffer, seen by this Function alone (stride-1 blocks: three additions + three zero fills of a
            # [32, c, p, 60] tensor per classification step otherwise)
            if not (getattr(base, "_epn_private", False) and base.numel() == grad_shared.numel()
                    and base.data_ptr() == grad_shared.data_ptr() and base.is_contiguous() and own(base)):
                return False
        ref = getattr(ctx, "shared_ref", None)
        shared = ref() if ref is not None else None
        if shared is not None and (shared.retains_grad or shared._backward_hooks):
            return False                                        # the very tensor would be observed as `.grad` / by a hook
        return True

    @staticmethod
    def backward(ctx, grad_out, grad_shared=None, _grad_part=None):
        if ctx.share_input == "stats":
            grad_shared = None
        lib = _lib.get_lib()
        G, Wc = ctx.saved_tensors
        geo, cin = ctx.geo, ctx.cin
        cout, ck = Wc.shape
        d = geo.desc(cin, cout)
        cols = d.b * d.p2 * d.na
        if grad_out is None:                   # only the shared input was used downstream
            return (grad_shared if ctx.needs_input_grad[0] else None), None, None, None
        g = cast_feats(to_cl(grad_out, "grad_out"), G.dtype)
        g2d = g.permute(0, 2, 3, 1).reshape(cols, cout)   # view of the channels-last buffer
        need_f, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gemm_fl = 2.0 * cols * cout * ck
        gf = gW = None
        # two-piece fp16 contractions: max|dOut| once for both GEMMs it feeds (a pass over the narrow operand), max|G| as the
        # forward pass bounded it
        go_amax = gemm.absmax_cached(g) if gemm.f16x2_on(G) else None      # (tagged by the norm backward that produced it)
        if need_w:
            gW = _launch("inter_gemm_dw", _inter_key(d), gemm_fl, G.device,
                         lambda: gemm.gemm_tn(g2d, G, x_amax=go_amax, y_amax=ctx.g_amax))
            if ctx.packed:                                           # computed against packed G: columns back in c*ks + k order
                gWp, gW = gW, torch.empty_like(gW)
                _lib.check(lib.epn_inter_unpack_weight_grad_f32(gWp.data_ptr(), cout, cin, d.ks, gW.data_ptr(),
                                                                _lib.stream_of(G)), "inter_unpack_weight_grad")
        if need_f:
            mode = os.environ.get("EPN_INTER_BWD_DATA", "auto")
            if mode in ("auto", "cloud") and _ungroup_cloud_takes(lib, d, geo, G.dtype, mode):
                # dG GEMM (its epilogue leaves max|dG|) + the transpose of the grouping with the cloud's gradient rows resident
                # in LDS (csrc/inter_ungroup_cloud.hip: 64-bit fixed-point accumulators, no global atomics): no zero fill of a
                # scatter target, the gradient is written in its own dtype, the other branch's gradient of a shared input is
                # folded into the write-out, and the result is bitwise repeatable (so this is also the deterministic form)
                Wt = gemm.transpose_cast(Wc, G.dtype)
                dG, dg_amax = _launch("inter_gemm_dg", _inter_key(d), gemm_fl, G.device,
                                      lambda: gemm.gemm_nt(g2d, Wt, a_amax=go_amax, c_amax=True))
                add = None
                if grad_shared is not None:
                    add = cast_feats(to_cl(grad_shared, "grad_shared"), G.dtype)
                gf = empty_cl(d.b, cin, d.p1, d.na, G.device, G.dtype)
                nb = int(lib.epn_inter_ungroup_cloud_workspace_bytes(ctypes.byref(d)))
                ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=G.device)
                gflops = 9.0 * cols * d.ks * d.nn + 2.0 * cols * cin * d.ks * d.nn
                args = [ctypes.byref(d), ctypes.c_void_p(dG.data_ptr()), gemm._use_amax(dg_amax), _cl_ptr(gf),
                        None if add is None else _cl_ptr(add)]
                if G.dtype == torch.bfloat16:
                    args.append(0)                       # out_f32 = 0: the gradient leaves in bf16
                fn = _entry(lib, "inter_ungroup_cloud", G.dtype)
                _lib.check(_launch("inter_ungroup", _inter_key(d), gflops, G.device,
                                   lambda: fn(*args, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                                              _lib.stream_of(G))), "inter_ungroup_cloud")
                return gf, gW, None, None
            if mode == "cloud":
                mode = "split"
            # the other branch's gradient of the shared input: fp32 -> the scatter accumulates onto it; otherwise added below
            # (bf16 features: starting the fp32 scatter target from the converted gradient instead of zeros measured no gain --
            # 1444 vs 1455 point-clouds/s on the rotation network -- so that path keeps the plain addition)
            onto = (grad_shared is not None and grad_shared.dtype == torch.float32 and G.dtype == torch.float32
                    and mode != "fused" and not deterministic_bwd(G.dtype)
                    and InterSO3ConvSplitFn._may_write_into(ctx, grad_shared))
            if onto:
                # the scatter writes this tensor through its raw pointer: bump its version so that no maximum remembered on
                # it (or on a view / alias of it) survives the write (gemm.mark_written)
                gf, grad_shared = gemm.mark_written(to_cl(grad_shared, "grad_shared").detach()), None
            else:
                gf = empty_cl(d.b, cin, d.p1, d.na, G.device)       # fp32: the scatter target of either dtype
            if G.dtype != torch.float32 or deterministic_bwd(G.dtype):
                mode = "split"          # bf16 features / deterministic mode: dG GEMM + (atomic-free) transpose of the grouping
            elif mode == "auto":
                # Three forms.  Measured per layer of the ModelNet schedule at B = 32 (ms; pair = dG GEMM + LDS-pre-reduced
                # transpose | on-chip kernel, csrc/inter_bwd_f2.hip: dG never written):
                #   stand-alone (tools/bwd_onchip_probe.py, profiles/r06_bwd_onchip_probe.txt)
                #     K = 16:  64->64 3.09 | 2.50   128->128 3.40 | 2.94   256->256 4.56 | 4.26
                #     K = 32:  64->128 2.17 | 2.80  128->256 2.52 | 3.24   256->256 2.58 | 3.16
                #   inside the training step (bench per_call, one box): K = 16: 2.75 | 2.72, 3.38 | 3.33, 4.39 | 4.81 -- the pair is
                #   HBM-bound and keeps its speed on a chip at its power limit, the on-chip kernel is latency-bound and does not;
                #   the step: 536-537 point-clouds/s (pair everywhere) vs 530-532 (on-chip at K = 16) vs 505-510 (on-chip everywhere).
                # So the pair stays the default; EPN_INTER_BWD_DATA=onchip is the form that moves 36-54 GB less per step and
                # allocates no [cols, cin*ks] gradient; round 1's fused exact-f32 kernel (=fused) writes no such tensor either.
                mode = "split"
            if (mode == "onchip" and gemm.f16x2_on(G) and go_amax is not None and isinstance(geo, InterGeometry)
                    and lib.epn_inter_bwd_data_f16x2_ok(ctypes.byref(d))):
                # dG never written (csrc/inter_bwd_f2.hip): the two-piece contraction dOut . W runs inside the workgroup of the
                # LDS-reduced scatter, its D fragments reach the tail through in-register row transposes
                nb = int(lib.epn_inter_bwd_data_f16x2_workspace_bytes(ctypes.byref(d)))
                ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=G.device)
                gemm._check_amax(g2d, go_amax, "inter_bwd_data_f16x2, operand dOut")
                _lib.check(_launch("inter_bwd_data_f2", _inter_key(d), _inter_flops(d), G.device,
                                   lambda: lib.epn_inter_bwd_data_f16x2_f32(ctypes.byref(d), _cl_ptr(g), _lib.dev_ptr(Wc, "W"),
                                                                            gemm._use_amax(go_amax), _cl_ptr(gf), 1 if onto else 0,
                                                                            ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                                                                            _lib.stream_of(G))), "inter_bwd_data_f16x2")
            elif mode == "fused" and lib.epn_inter_is_fused(ctypes.byref(d)) and cin >= 16:
                # The fused data-gradient kernel (W^T dOut + per-column tail in one pass, no dG tensor): its fp32 atomic
                # scatter -- cols*K*cin = 1.0e9 atomics per layer -- hides under the MFMA phases.  Also measured and
                # dropped: an atomic-free CSR-gather transpose (re-reads every 96-byte dG row K times: 3x slower) and
                # running the weight-gradient GEMM on a side stream underneath the scatter (no overlap: +-1 ms).
                ws, wsp, wsn = _workspace(lib, d, G.device)
                _lib.check(_launch("inter_bwd_data", _inter_key(d), _inter_flops(d), G.device,
                                   lambda: lib.epn_inter_so3conv_bwd_data_f32(ctypes.byref(d), _cl_ptr(g),
                                                                              _lib.dev_ptr(Wc, "W"), _cl_ptr(gf), wsp,
                                                                              wsn, _lib.stream_of(G))),
                           "inter_so3conv_bwd_data")
            else:
                Wt = gemm.transpose_cast(Wc, G.dtype)                        # [ck, cout]: dG = dOut W as an NT GEMM
                dG = _launch("inter_gemm_dg", _inter_key(d), gemm_fl, G.device, lambda: gemm.gemm_nt(g2d, Wt, a_amax=go_amax))
                ws, wsp, wsn = _group_workspace(lib, d, G.device)
                gflops = 9.0 * cols * d.ks * d.nn + 2.0 * cols * cin * d.ks * d.nn
                if deterministic_bwd(G.dtype) and isinstance(geo, InterGeometry) and d.na >= 16:
                    # atomic-free: per-slot slab + ordered reduction over the inverse neighbour list (bitwise repeatable;
                    # for bf16 also faster than the fp32 atomic scatter + conversion at K >= 32)
                    off, ent = geo.inverse_list()
                    gf = empty_cl(d.b, cin, d.p1, d.na, G.device, G.dtype)
                    # + one byte per (point, neighbour slot) behind the slab: marks of the pre-reduced form
                    extra = (d.b * d.p2 * d.nn + 512 + G.element_size() - 1) // G.element_size()
                    slab = torch.empty(d.b * d.p2 * d.nn * d.na * cin + extra, dtype=G.dtype, device=G.device)
                    det = _entry(lib, "inter_ungroup_det", G.dtype)
                    _lib.check(_launch("inter_ungroup_det", _inter_key(d), gflops, G.device,
                                       lambda: det(ctypes.byref(d), ctypes.c_void_p(dG.data_ptr()), _cl_ptr(gf),
                                                   _lib.dev_ptr(off, "offsets", torch.int32),
                                                   _lib.dev_ptr(ent, "entries", torch.int32),
                                                   ctypes.c_void_p(slab.data_ptr()), slab.numel() * slab.element_size(),
                                                   wsp, wsn, _lib.stream_of(G))), "inter_ungroup_det")
                    if grad_shared is not None:
                        gf = gf + grad_shared.to(gf.dtype)
                    return gf, gW, None, None
                ungrp = _entry(lib, "inter_ungroup_acc" if onto else "inter_ungroup", G.dtype)
                _lib.check(_launch("inter_ungroup", _inter_key(d), gflops, G.device,
                                   lambda: ungrp(ctypes.byref(d), ctypes.c_void_p(dG.data_ptr()), _cl_ptr(gf), wsp, wsn,
                                                 _lib.stream_of(G))), "inter_ungroup")
            if (grad_shared is not None and G.dtype == torch.bfloat16 and gf.dtype == torch.float32
                    and grad_shared.dtype == torch.bfloat16):
                gs = to_cl(grad_shared, "grad_shared")           # bf16(gf) + the other branch's gradient in one pass
                out = empty_cl(d.b, cin, d.p1, d.na, G.device, torch.bfloat16)
                _lib.check(lib.epn_cast_add_bf16(gf.data_ptr(), gs.data_ptr(), out.data_ptr(), gf.numel(), _lib.stream_of(gf)),
                           "cast_add")
                return out, gW, None, None
            gf = cast_feats(gf, G.dtype)
            if grad_shared is not None:
                gf = gf + grad_shared.to(gf.dtype)
        return gf, gW, None, None


def _ungroup_cloud_takes(lib, d, geo, dtype, mode):
    """Does the cloud-resident transpose of the grouping (epn_inter_ungroup_cloud_*) take this layer?  mode "cloud": whenever
    the kernel can; "auto": where it is measured faster INSIDE the training step than the LDS-pre-reduced atomic scatter
    (profiles/r06_ab_ungroup_cloud.txt, A/B on one box) -- bf16 features: rotation network 1957-1966 -> 2095-2115
    point-clouds/s, 3DMatch 1890-1919 -> 2033-2066 (no fp32 scatter target, zero fill or conversion pass either); fp32
    features up to K = 32: cls 530-532 -> 537-538 (until its table kernel was fixed -- 91 -> 20 us per call -- this was a tie);
    fp32 at K = 64 regenerates its weights per 16 channels under the register cap and loses (rotation network in fp32 with the
    form everywhere: 1033 -> 1004) -- and always in deterministic mode, which it satisfies by construction (the slab-based
    kernels cost 3.6-8 % of a step)."""
    if not isinstance(geo, InterGeometry) or not lib.epn_inter_ungroup_cloud_ok(ctypes.byref(d)):
        return False
    if mode == "cloud" or deterministic_bwd(dtype):
        return True
    return dtype == torch.bfloat16 or d.nn <= 32


def _lib_generic():
    return False      # the generic-kernel policy is only ever set by the cross-check tests, around whole calls


def _onchip_workspace(lib, d, bf16, device):
    nbytes = lib.epn_inter_onchip_workspace_bytes(ctypes.byref(d), int(bf16))
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return ws, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())


def inter_onchip_ok(feats, W, geo):
    """Does the on-chip form (csrc/inter_fx.hip: grouping = A-tile producer of the weight contraction) take this layer?"""
    if isinstance(geo, DenseInterWeights) or not feats.is_cuda or feats.dtype not in FEATURE_DTYPES:
        return False
    if feats.dtype == torch.float32 and gemm.FP32_MODE == "native":
        return False                      # the on-chip fp32 form IS a split (3 x bf16) contraction: not under the exact-f32 switch
    d = geo.desc(feats.shape[1], W.shape[0])
    return bool(_lib.get_lib().epn_inter_onchip_ok(ctypes.byref(d), int(feats.dtype == torch.bfloat16)))


def inter_onchip_fwd(f, Wc, geo):
    """out_cl [b, cout, p2, na] (channels-last) of InterSO3Conv with no [cols, cin*ks] tensor (epn_inter_so3conv_fwd_onchip_f32 /
    epn_inter_so3conv_fwd_bf16)."""
    lib = _lib.get_lib()
    cout, ck = Wc.shape
    cin = f.shape[1]
    d = geo.desc(cin, cout)
    if ck != cin * d.ks or f.shape[2] != d.p1 or f.shape[3] != d.na or f.shape[0] != d.b:
        raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, W {tuple(Wc.shape)}, geometry "
                         f"b={d.b} p1={d.p1} na={d.na} ks={d.ks}")
    bf = f.dtype == torch.bfloat16
    out = empty_cl(d.b, cout, d.p2, d.na, f.device, f.dtype)
    ws, wsp, wsn = _onchip_workspace(lib, d, bf, f.device)
    fn = lib.epn_inter_so3conv_fwd_bf16 if bf else lib.epn_inter_so3conv_fwd_onchip_f32
    _lib.check(_launch("inter_fwd_onchip", _inter_key(d), _inter_flops(d), f.device,
                       lambda: fn(ctypes.byref(d), _cl_ptr(f), _lib.dev_ptr(Wc, "W"), _cl_ptr(out), wsp, wsn,
                                  _lib.stream_of(f))), "inter_so3conv_fwd_onchip")
    return out


class InterSO3ConvOnChipFn(torch.autograd.Function):
    """InterSO3Conv with the grouped features kept on chip (csrc/inter_fx.hip): forward saves only its inputs."""

    @staticmethod
    def forward(ctx, feats, W, geo):
        f = to_cl(feats)
        Wc = W.contiguous()
        out = inter_onchip_fwd(f, Wc, geo)
        ctx.save_for_backward(f, Wc)
        ctx.geo = geo
        return out

    @staticmethod
    def backward(ctx, grad_out):
        f, Wc = ctx.saved_tensors
        geo = ctx.geo
        need_f, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        lib = _lib.get_lib()
        d = geo.desc(f.shape[1], Wc.shape[0])
        if f.dtype == torch.float32 and lib.epn_inter_is_fused(ctypes.byref(d)) and not _lib_generic():
            # fp32: the fused transposes of csrc/inter_mfma.hip (exact-f32 MFMAs; grouped features and their gradient stay
            # on chip there too) -- the whole layer then writes no [cols, cin*ks] tensor in either direction
            g = to_cl(grad_out, "grad_out")
            ws, wsp, wsn = _workspace(lib, d, f.device)
            gf = gW = None
            if need_f:
                gf = empty_cl(d.b, f.shape[1], d.p1, d.na, f.device)
                _lib.check(_launch("inter_bwd_data", _inter_key(d), _inter_flops(d), f.device,
                                   lambda: lib.epn_inter_so3conv_bwd_data_f32(ctypes.byref(d), _cl_ptr(g), _lib.dev_ptr(Wc, "W"),
                                                                              _cl_ptr(gf), wsp, wsn, _lib.stream_of(f))),
                           "inter_so3conv_bwd_data")
            if need_w:
                gW = torch.empty_like(Wc)
                _lib.check(_launch("inter_bwd_weight", _inter_key(d), _inter_flops(d), f.device,
                                   lambda: lib.epn_inter_so3conv_bwd_weight_f32(ctypes.byref(d), _cl_ptr(f), _cl_ptr(g),
                                                                                _lib.dev_ptr(gW, "grad_W"), wsp, wsn,
                                                                                _lib.stream_of(f))),
                           "inter_so3conv_bwd_weight")
            return gf, gW, None
        # bf16 features: transposes through the split form (grouped features recomputed, not saved)
        with torch.enable_grad():
            fd = f.detach().requires_grad_(need_f)
            Wd = Wc.detach().requires_grad_(need_w)
            y = InterSO3ConvSplitFn.apply(fd, Wd, geo)
        ins = [t for t, n in ((fd, need_f), (Wd, need_w)) if n]
        grads = list(torch.autograd.grad(y, ins, grad_out))
        gf = grads.pop(0) if need_f else None
        gW = grads.pop(0) if need_w else None
        return gf, gW, None


_SIDE = {}


def _side_stream(device):
    key = str(torch.device(device))
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def side_stream_if_any(device):
    """The second stream of `device` if one was ever created (dp.GradBuckets orders its collectives after it)."""
    return _SIDE.get(str(torch.device(device)))


class _TensorCache:
    """Values derived from an index tensor, cached per LIVE tensor object: an entry holds a weak reference to the tensor it
    was derived from and its in-place version, and is used only while that very object is alive and unmodified (a key
    made of data_ptr() alone would hand a recycled address the previous table's entry).  Bounded."""

    def __init__(self, maxlen=32):
        self._d, self._maxlen = {}, maxlen

    def get(self, t, make):
        import weakref
        k = id(t)
        e = self._d.get(k)
        if e is not None and e[0]() is t and e[1] == t._version:
            return e[2]
        if len(self._d) >= self._maxlen:
            self._d = {kk: ee for kk, ee in self._d.items() if ee[0]() is not None}
            if len(self._d) >= self._maxlen:
                self._d.clear()
        v = make(t)
        self._d[k] = (weakref.ref(t), t._version, v)
        return v


_INV_CACHE = _TensorCache()


def _make_inverse(intra_idx32):
    host = intra_idx32.cpu().long()
    na, kn = host.shape
    if not all(sorted(host[:, k].tolist()) == list(range(na)) for k in range(kn)):
        return None
    inv = torch.empty_like(host)
    inv.scatter_(0, host, torch.arange(na).view(-1, 1).expand(-1, kn))
    return inv.int().to(intra_idx32.device)


def inverse_intra_idx(intra_idx32):
    """inv[idx[a,k], k] = a when every column of intra_idx is a permutation of the anchors (true for the
    icosahedral table, tests/test_tables.py); None otherwise.  Cached per live index tensor."""
    return _INV_CACHE.get(intra_idx32, _make_inverse)


def _intra_ws(lib, na, kn, cin, cout, device):
    nbytes = lib.epn_intra_workspace_bytes(na, kn, cin, cout)
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return ws, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel())


class IntraSO3ConvFn(torch.autograd.Function):
    """out[b,o,p,a] = sum_{c,k} W[o,c*kn+k] feats[b,c,p,intra_idx[a,k]]
    (IntraSO3Conv.forward, vgtk/vgtk/so3conv/modules.py:197-200), fused."""

    @staticmethod
    def forward(ctx, feats, W, intra_idx32):
        lib = _lib.get_lib()
        f = to_cl(feats)
        Wc = W.contiguous()
        b, cin, p, na = f.shape
        cout = Wc.shape[0]
        kn = intra_idx32.shape[1]
        if Wc.shape[1] != cin * kn or intra_idx32.shape[0] != na:
            raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, W {tuple(Wc.shape)}, "
                             f"intra_idx {tuple(intra_idx32.shape)}")
        out = empty_cl(b, cout, p, na, f.device)
        ws, wsp, wsn = _intra_ws(lib, na, kn, cin, cout, f.device)
        fl = 2.0 * b * p * na * cout * cin * kn
        _lib.check(_launch("intra_fwd", (b, p, na, kn, cin, cout), fl, f.device,
                           lambda: lib.epn_intra_so3conv_fwd_f32(
                               _cl_ptr(f), _lib.dev_ptr(intra_idx32, "intra_idx", torch.int32), _lib.dev_ptr(Wc, "W"),
                               b, p, na, kn, cin, cout, _cl_ptr(out), wsp, wsn, _lib.stream_of(f))),
                   "intra_so3conv_fwd")
        ctx.save_for_backward(f, Wc, intra_idx32)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.get_lib()
        f, Wc, iidx = ctx.saved_tensors
        g = to_cl(grad_out, "grad_out")
        b, cin, p, na = f.shape
        cout = Wc.shape[0]
        kn = iidx.shape[1]
        ip = _lib.dev_ptr(iidx, "intra_idx", torch.int32)
        fl = 2.0 * b * p * na * cout * cin * kn
        gf = gW = None
        if ctx.needs_input_grad[0]:
            gf = empty_cl(b, cin, p, na, f.device)
            inv = inverse_intra_idx(iidx)
            ws, wsp, wsn = _intra_ws(lib, na, kn, cin, cout, f.device)
            _lib.check(_launch("intra_bwd_data", (b, p, na, kn, cin, cout), fl, f.device,
                               lambda: lib.epn_intra_so3conv_bwd_data_f32(
                                   _cl_ptr(g), ip, _lib.dev_ptr(inv, "inv_idx", torch.int32), _lib.dev_ptr(Wc, "W"),
                                   b, p, na, kn, cin, cout, _cl_ptr(gf), wsp, wsn, _lib.stream_of(f))),
                       "intra_so3conv_bwd_data")
        if ctx.needs_input_grad[1]:
            gW = torch.empty_like(Wc)
            _lib.check(_launch("intra_bwd_weight", (b, p, na, kn, cin, cout), fl, f.device,
                               lambda: lib.epn_intra_so3conv_bwd_weight_f32(
                                   _cl_ptr(f), _cl_ptr(g), ip, b, p, na, kn, cin, cout, _lib.dev_ptr(gW, "grad_W"),
                                   _lib.stream_of(f))),
                       "intra_so3conv_bwd_weight")
        return gf, gW, None


def _intra_group(lib, f, idx32, b, p, na, kn, c):
    """grouped[col][k*c + ci] = f[b][p][idx[a,k]][ci] (epn_intra_group_*), any feature dtype."""
    G = torch.empty((b * p * na, kn * c), dtype=f.dtype, device=f.device)
    _lib.check(_launch("intra_group", (b, p, na, kn, c), 0.0, f.device,
                       lambda: _entry(lib, "intra_group", f.dtype)(
                           _cl_ptr(f), _lib.dev_ptr(idx32, "intra_idx", torch.int32), ctypes.c_void_p(G.data_ptr()),
                           b, p, na, kn, c, _lib.stream_of(f))), "intra_group")
    return G


class IntraSO3ConvSplitFn(torch.autograd.Function):
    """IntraSO3Conv in the reference's two steps (intra_so3conv_grouping, then BasicSO3Conv's matmul): the anchor
    gather as one streaming HIP kernel writing grouped[col][kn*cin], `out = grouped Wp^T` and `dW = dOut^T grouped` on
    the library's GEMM kernels.  Data gradient: fp32 on the fused kernel (no grouped-gradient tensor); bf16 as the same
    two steps through the inverse anchor permutation.  `grouped` is kept for the backward pass."""

    @staticmethod
    def forward(ctx, feats, W, intra_idx32):
        lib = _lib.get_lib()
        f = to_cl(feats)
        Wc = W.contiguous()
        b, cin, p, na = f.shape
        cout = Wc.shape[0]
        kn = intra_idx32.shape[1]
        if Wc.shape[1] != cin * kn or intra_idx32.shape[0] != na:
            raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, W {tuple(Wc.shape)}, "
                             f"intra_idx {tuple(intra_idx32.shape)}")
        cols = b * p * na
        G = _intra_group(lib, f, intra_idx32, b, p, na, kn, cin)
        Wp = gemm.cast(Wc.view(cout, cin, kn).permute(0, 2, 1).reshape(cout, kn * cin), f.dtype)     # [o][k*cin + c]
        fl = 2.0 * cols * cout * cin * kn
        out2d = _launch("intra_gemm", (b, p, na, kn, cin, cout), fl, f.device, lambda: gemm.gemm_nt(G, Wp))
        ctx.save_for_backward(G, Wc, intra_idx32)
        ctx.dims = (b, cin, p, na)
        return out2d.view(b, p, na, cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.get_lib()
        G, Wc, iidx = ctx.saved_tensors
        b, cin, p, na = ctx.dims
        cout, kn = Wc.shape[0], iidx.shape[1]
        cols = b * p * na
        g = cast_feats(to_cl(grad_out, "grad_out"), G.dtype)
        fl = 2.0 * cols * cout * cin * kn
        gf = gW = None
        if ctx.needs_input_grad[1]:
            g2d = g.permute(0, 2, 3, 1).reshape(cols, cout)
            gWp = _launch("intra_gemm_dw", (b, p, na, kn, cin, cout), fl, G.device, lambda: gemm.gemm_tn(g2d, G))
            gW = gWp.view(cout, kn, cin).permute(0, 2, 1).reshape(cout, cin * kn)
        if ctx.needs_input_grad[0]:
            inv = inverse_intra_idx(iidx)
            if G.dtype == torch.float32 or inv is None:
                gf = empty_cl(b, cin, p, na, G.device)
                g32 = cast_feats(g, torch.float32)
                ws, wsp, wsn = _intra_ws(lib, na, kn, cin, cout, G.device)
                _lib.check(_launch("intra_bwd_data", (b, p, na, kn, cin, cout), fl, G.device,
                                   lambda: lib.epn_intra_so3conv_bwd_data_f32(
                                       _cl_ptr(g32), _lib.dev_ptr(iidx, "intra_idx", torch.int32),
                                       _lib.dev_ptr(inv, "inv_idx", torch.int32), _lib.dev_ptr(Wc, "W"),
                                       b, p, na, kn, cin, cout, _cl_ptr(gf), wsp, wsn, _lib.stream_of(G))),
                           "intra_so3conv_bwd_data")
                gf = cast_feats(gf, G.dtype)
            else:
                # dF[col][ci] = sum_{k,o} dOut[pt, inv[a,k], o] W[o, ci*kn + k]: gather through the inverse permutation,
                # then an NT GEMM against W re-ordered to [ci][k*cout + o]
                Gd = _intra_group(lib, g, inv, b, p, na, kn, cout)
                Wq = gemm.cast(Wc.view(cout, cin, kn).permute(1, 2, 0).reshape(cin, kn * cout), G.dtype)
                gf2d = _launch("intra_gemm", (b, p, na, kn, cout, cin), fl, G.device, lambda: gemm.gemm_nt(Gd, Wq))
                gf = gf2d.view(b, p, na, cin).permute(0, 3, 1, 2)
        return gf, gW, None


# ------------------------------------------------------------------------------------------------------------------
# IntraSO3Conv in the block-diagonalising anchor basis (so3_fourier.py): U^T, one GEMM per irreducible block, U.
_BASIS_CACHE = _TensorCache()


class _SpectralBasis:
    """Device-side tables of so3_fourier.build() for one intra_idx tensor."""

    def __init__(self, intra_idx32):
        from . import so3_fourier
        import numpy as np
        bz = so3_fourier.build(intra_idx32.detach().cpu().numpy())
        dev = intra_idx32.device
        U = bz["U"].astype(np.float32)
        self.na = U.shape[0]
        self.U = torch.from_numpy(np.ascontiguousarray(U)).to(dev)            # [a][f]
        self.Ut = torch.from_numpy(np.ascontiguousarray(U.T)).to(dev)         # [f][a]
        self.dims = list(bz["dims"])
        self.bases, blocks, base = [], [], 0
        for d in self.dims:
            self.bases.append(base)
            blocks += [(base, d * d)] * (d * d)
            base += d * d
        self.blocks = torch.tensor(blocks, dtype=torch.int32, device=dev)     # [na][2]
        self.rho = [torch.from_numpy(r.astype(np.float32)).to(dev) for r in bz["rho"]]   # [kn, d, d] each
        self.rho_all = torch.cat([r.reshape(r.shape[0], -1) for r in self.rho], dim=1).contiguous()   # [kn, na], (i, j)
        self.rho_all_t = self.rho_all.t().contiguous()                                                 # [na, kn]


def _make_basis(intra_idx32):
    na = intra_idx32.shape[0]
    if not (na <= 64 and na % 4 == 0):
        return None
    try:
        return _SpectralBasis(intra_idx32)
    except ValueError:
        return None


def spectral_basis(intra_idx32):
    """Cached per live index tensor; None when the table is not a regular permutation group action (then the
    12-neighbour forms are used) or the anchor count does not fit the transform kernel."""
    return _BASIS_CACHE.get(intra_idx32, _make_basis)


def _tag_amax(t, amax):
    """Remember a producer-side max|t| on the tensor object (what gemm.absmax_cached looks for)."""
    try:
        t._epn_amax = (t._version, amax)
    except (AttributeError, RuntimeError):
        pass


def _basis_call(lib, src, M, basis, pts, c, in_spec, out_spec, dst, kind):
    if src.dtype != dst.dtype or src.dtype not in FEATURE_DTYPES:
        raise TypeError(f"so3_basis: {src.dtype} -> {dst.dtype}")
    if out_spec and gemm.f16x2_on(dst):
        # a spectral buffer is the operand of two-piece fp16 GEMMs: its maximum from this kernel's accumulators, not from a
        # pass over the 250 MB it writes
        amax = torch.empty(1, dtype=torch.float32, device=dst.device)
        _lib.check(_launch(kind, ("so3_basis", pts, c), 2.0 * pts * basis.na * basis.na * c, src.device,
                           lambda: lib.epn_so3_basis_amax_split_f32(ctypes.c_void_p(src.data_ptr()), _lib.dev_ptr(M, "M"),
                                                                    _lib.dev_ptr(basis.blocks, "blocks", torch.int32),
                                                                    ctypes.c_longlong(pts), basis.na, c, in_spec, out_spec,
                                                                    ctypes.c_void_p(dst.data_ptr()), amax.data_ptr(),
                                                                    _lib.stream_of(src))), "so3_basis_amax")
        _tag_amax(dst, amax)
        return
    fn = _entry(lib, "so3_basis", src.dtype)
    _lib.check(_launch(kind, ("so3_basis", pts, c), 2.0 * pts * basis.na * basis.na * c, src.device,
                       lambda: fn(ctypes.c_void_p(src.data_ptr()), _lib.dev_ptr(M, "M"),
                                                     _lib.dev_ptr(basis.blocks, "blocks", torch.int32),
                                                     ctypes.c_longlong(pts), basis.na, c, in_spec, out_spec,
                                                     ctypes.c_void_p(dst.data_ptr()), _lib.stream_of(src))),
               "so3_basis")


class ToSpectralFn(torch.autograd.Function):
    """[b,c,p,a] channels-last -> flat spectral buffer [na * b*p * c] (block rho at offset base_rho * pts * c, laid out
    [pts * d][d * c]); backward is the inverse transform (U is orthogonal)."""

    @staticmethod
    def forward(ctx, feats, basis):
        lib = _lib.get_lib()
        f = to_cl(feats)
        b, c, p, na = f.shape
        y = torch.empty(na * b * p * c, dtype=f.dtype, device=f.device)
        _basis_call(lib, f, basis.Ut, basis, b * p, c, 0, 1, y, "so3_basis")
        ctx.basis, ctx.dims = basis, (b, c, p, na)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.get_lib()
        b, c, p, na = ctx.dims
        gf = empty_cl(b, c, p, na, gy.device, gy.dtype)
        _basis_call(lib, gy.contiguous(), ctx.basis.U, ctx.basis, b * p, c, 1, 0, gf, "so3_basis")
        return gf, None


class FromSpectralFn(torch.autograd.Function):
    """flat spectral buffer -> [b,c,p,a] channels-last (out[a] = sum_f U[a,f] y[f]); backward is ToSpectral."""

    @staticmethod
    def forward(ctx, y, basis, b, p, c, out_stats=False):
        lib = _lib.get_lib()
        ctx.set_materialize_grads(False)   # else autograd zero-fills a gradient for every non-differentiable output, every step
        out = empty_cl(b, c, p, basis.na, y.device, y.dtype)
        ctx.basis, ctx.dims = basis, (b, c, p)
        if not out_stats:
            _basis_call(lib, y.contiguous(), basis.U, basis, b * p, c, 1, 0, out, "so3_basis")
            return out
        # + per-point partials of out's per-channel statistics, from the accumulators (epn_so3_basis_stats_*)
        part = torch.empty((b * p, c, 2), dtype=torch.float32, device=y.device)
        src = y.contiguous()
        fn = _entry(lib, "so3_basis_stats", src.dtype)
        _lib.check(_launch("so3_basis", ("so3_basis", b * p, c), 2.0 * b * p * basis.na * basis.na * c, src.device,
                           lambda: fn(ctypes.c_void_p(src.data_ptr()), _lib.dev_ptr(basis.U, "M"),
                                      _lib.dev_ptr(basis.blocks, "blocks", torch.int32), ctypes.c_longlong(b * p), basis.na,
                                      c, 1, 0, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(part.data_ptr()),
                                      _lib.stream_of(src))), "so3_basis_stats")
        ctx.mark_non_differentiable(part)
        return out, part

    @staticmethod
    def backward(ctx, gout, _gpart=None):
        if gout is None:
            return None, None, None, None, None, None
        lib = _lib.get_lib()
        b, c, p = ctx.dims
        g = to_cl(gout, "grad_out")
        gy = torch.empty(ctx.basis.na * b * p * c, dtype=g.dtype, device=g.device)
        _basis_call(lib, g, ctx.basis.Ut, ctx.basis, b * p, c, 0, 1, gy, "so3_basis")
        return gy, None, None, None, None, None


class SpectralWeightsFn(torch.autograd.Function):
    """IntraSO3Conv's fp32 weights [cout, cin*kn] -> the five block matrices What^rho [d*cin, d*cout] (views of one flat
    buffer) and, without gradient, their transposes (the Bt operands of the forward GEMMs): ONE pass of
    epn_spectral_weights_f32 per layout instead of a small GEMM + five slice / permute / clone chains, and one
    epn_spectral_weights_bwd_f32 instead of their autograd transposes."""

    @staticmethod
    def forward(ctx, W, basis, cin, cout, bf16_ops=False):
        """Returns (whats fp32 x5 [differentiable], forward operands What^T x5, backward operands What x5): the operand sets
        are fp32 (the second = the differentiable blocks themselves) or, bf16_ops, bf16 copies written by the same kernel."""
        ctx.set_materialize_grads(False)   # else autograd zero-fills a gradient for every non-differentiable output, every step
        lib = _lib.get_lib()
        Wc = W.contiguous()
        na, kn = basis.rho_all_t.shape
        n = na * cin * cout
        cc = cin * cout
        dev = W.device
        st = _lib.stream_of(Wc)
        R, blk = basis.rho_all_t.data_ptr(), basis.blocks.data_ptr()

        def views(buf, t):
            return [buf[b0 * cc:(b0 + d * d) * cc].view(*((d * cout, d * cin) if t else (d * cin, d * cout)))
                    for d, b0 in zip(basis.dims, basis.bases)]
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        if bf16_ops:
            fb, fbt = (torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(2))
            _lib.check(lib.epn_spectral_weights_f32(Wc.data_ptr(), R, blk, cout, cin, kn, na, flat.data_ptr(), None, st),
                       "spectral_weights")
            _lib.check(lib.epn_spectral_weights_bf16(Wc.data_ptr(), R, blk, cout, cin, kn, na, fb.data_ptr(), fbt.data_ptr(), st),
                       "spectral_weights_bf16")
            ops_t, ops_n = views(fbt, True), views(fb, False)
        else:
            flat_t = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(lib.epn_spectral_weights_f32(Wc.data_ptr(), R, blk, cout, cin, kn, na, flat.data_ptr(), flat_t.data_ptr(),
                                                    st), "spectral_weights")
            ops_t, ops_n = views(flat_t, True), None
        whats = views(flat, False)
        ctx.basis, ctx.cfg = basis, (cin, cout, tuple(W.shape))
        if ops_n is None:
            ctx.mark_non_differentiable(*ops_t)
            return (*whats, *ops_t)
        ctx.mark_non_differentiable(*ops_t, *ops_n)
        return (*whats, *ops_t, *ops_n)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.get_lib()
        basis = ctx.basis
        cin, cout, wshape = ctx.cfg
        na, kn = basis.rho_all_t.shape
        cc = cin * cout
        gs = grads[:len(basis.dims)]
        dev = basis.rho_all_t.device
        # the weight-gradient GEMMs write their results as consecutive slices of one buffer (_BlockGemmsFn.backward): use it
        flat = None
        if all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for g in gs):
            p0 = gs[0].data_ptr()
            if all(g.data_ptr() == p0 + 4 * b0 * cc for g, b0 in zip(gs, basis.bases)) and \
                    gs[0].untyped_storage().nbytes() - gs[0].storage_offset() * 4 >= 4 * na * cc:
                flat = gs[0]
        if flat is None:
            flat = torch.cat([(g.float().reshape(-1) if g is not None else torch.zeros(d * d * cc, device=dev))
                              for g, d in zip(gs, basis.dims)])
        gW = torch.empty(cout * cin * kn, dtype=torch.float32, device=dev)
        _lib.check(lib.epn_spectral_weights_bwd_f32(flat.data_ptr(), basis.rho_all_t.data_ptr(), basis.blocks.data_ptr(), cout,
                                                    cin, kn, na, gW.data_ptr(), _lib.stream_of(flat)), "spectral_weights_bwd")
        return gW.view(wshape), None, None, None, None


class _BlockGemmsFn(torch.autograd.Function):
    """All irreducible blocks of one layer: Z^rho = Y^rho @ What^rho as ONE grouped launch of the library's NT GEMM
    kernel (csrc/gemm.hip), written straight into the slices of one spectral output buffer (no concatenation), and
    likewise for the data gradient; the weight gradients are TN GEMMs (deterministic split over the points).
    whats[i]: What^rho [d*cin, d*cout]."""

    @staticmethod
    def forward(ctx, y, basis, pts, cin, cout, whats_t, *whats):
        """whats_t: None, or (What^T x5, What x5 or None) in y's dtype when the caller already has them (SpectralWeightsFn):
        the Bt operands of the forward and of the data-gradient GEMMs."""
        whats_n = None
        if whats_t is not None:
            whats_t, whats_n = whats_t
        z = torch.empty(basis.na * pts * cout, dtype=y.dtype, device=y.device)
        probs, fl = [], 0.0
        for bi, (d, base, wh) in enumerate(zip(basis.dims, basis.bases, whats)):
            A = y[base * pts * cin:(base + d * d) * pts * cin].view(pts * d, d * cin)
            O = z[base * pts * cout:(base + d * d) * pts * cout].view(pts * d, d * cout)
            wt = whats_t[bi] if whats_t is not None and whats_t[bi].dtype == y.dtype else gemm.transpose_cast(wh, y.dtype)
            probs.append((A, wt, O))                                        # Bt = What^T [d*cout, d*cin]
            fl += 2.0 * pts * d * d * cin * d * cout
        # two-piece fp16 contractions: ONE maximum for the whole spectral buffer (its five slices are the operands)
        y_amax = gemm.absmax_cached(y) if gemm.f16x2_on(y) else None      # (tagged by the basis change that wrote y)
        _launch("intra_gemm", ("spectral", pts, cin, cout), fl, y.device,
                lambda: gemm.gemm_nt_grouped(probs, a_amax=None if y_amax is None else [y_amax] * len(probs)))
        ctx.y_amax = y_amax
        ctx.save_for_backward(y, *whats)
        ctx.cfg = (basis, pts, cin, cout)
        ctx.ops_n = whats_n if whats_n is not None and whats_n[0].dtype == y.dtype else None
        return z

    @staticmethod
    def backward(ctx, gz):
        y, *whats = ctx.saved_tensors
        basis, pts, cin, cout = ctx.cfg
        gz_amax = gemm.absmax_cached(gz) if gemm.f16x2_on(gz) else None  # (tagged by the basis change that wrote gz)
        gz = gz.contiguous()
        gy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        gws, probs, tprobs, tidx, fl, flw = [None] * len(whats), [], [], [], 0.0, 0.0
        for bi, (d, base, wh) in enumerate(zip(basis.dims, basis.bases, whats)):
            A = y[base * pts * cin:(base + d * d) * pts * cin].view(pts * d, d * cin)
            G = gz[base * pts * cout:(base + d * d) * pts * cout].view(pts * d, d * cout)
            f1 = 2.0 * pts * d * d * cin * d * cout
            if gy is not None:
                gA = gy[base * pts * cin:(base + d * d) * pts * cin].view(pts * d, d * cin)
                wn = ctx.ops_n[bi] if ctx.ops_n is not None else gemm.cast(wh, y.dtype)
                probs.append((G, wn, gA))                                   # dY = dZ What^T: Bt = What [d*cin, d*cout]
                fl += f1
            if ctx.needs_input_grad[6 + bi]:
                tprobs.append((A, G))
                tidx.append(bi)
                flw += f1
        if tprobs:      # the five weight-gradient GEMMs, each too small to fill the chip alone: ONE grouped TN launch,
            cc = cin * cout                        # results as consecutive slices of one buffer (SpectralWeightsFn.backward)
            gw_flat = torch.empty(basis.na * cc, dtype=torch.float32, device=y.device)
            into = [gw_flat[basis.bases[bi] * cc:(basis.bases[bi] + basis.dims[bi] ** 2) * cc].view(
                basis.dims[bi] * cin, basis.dims[bi] * cout) for bi in tidx]
            outs = _launch("intra_gemm_dw", ("spectral_dw", pts, cin, cout), flw, y.device,
                           lambda: gemm.gemm_tn_grouped(tprobs, into,
                                                        x_amax=None if ctx.y_amax is None else [ctx.y_amax] * len(tprobs),
                                                        y_amax=None if gz_amax is None else [gz_amax] * len(tprobs)))
            for bi, o in zip(tidx, outs):
                gws[bi] = o
        if probs:
            _launch("intra_gemm", ("spectral_dA", pts, cin, cout), fl, y.device,
                    lambda: gemm.gemm_nt_grouped(probs, a_amax=None if gz_amax is None else [gz_amax] * len(probs)))
        return (gy, None, None, None, None, None, *gws)


def intra_so3conv_spectral(feats, W, intra_idx32, basis, pre_norm=None, pre_slope=0.01, pre_part=None, out_stats=False):
    """IntraSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:197-200) in the block-diagonal anchor basis: same result up to
    fp32 rounding, 244 instead of 720 multiply-adds per (point, cin, cout), no [cols, 12*cin] grouped tensor; gradients
    by autograd through the same pieces (GEMMs and transforms on this library's HIP kernels)."""
    f = to_cl(feats)
    b, cin, p, na = f.shape
    cout, kn = W.shape[0], intra_idx32.shape[1]
    if W.shape[1] != cin * kn or intra_idx32.shape[0] != na:
        raise ValueError(f"shape mismatch: feats {tuple(f.shape)}, W {tuple(W.shape)}, intra_idx {tuple(intra_idx32.shape)}")
    pts = b * p
    if pre_norm is not None:
        # the block's preceding norm + leaky_relu, folded into the transform's loads (training-mode statistics)
        import torch.nn as nn
        inst = isinstance(pre_norm, nn.InstanceNorm2d)
        y, sums = NormToSpectralFn.apply(f, getattr(pre_norm, "weight", None), getattr(pre_norm, "bias", None), None,
                                         inst, pre_norm.eps, pre_slope, basis, pre_part)
        _update_running_stats(pre_norm, sums, b * p * na)
    else:
        y = ToSpectralFn.apply(f, basis)
    # What^rho[(j, c), (i, o)] = sum_k W[o, c, k] rho(g_k)[i, j]: one small GEMM for all blocks, then a re-layout each
    if W.dtype == torch.float32 and kn <= 16 and ab("EPN_SPECTRAL_WEIGHTS") == "fused":
        nb_ = len(basis.dims)
        outs = SpectralWeightsFn.apply(W, basis, cin, cout, y.dtype == torch.bfloat16)   # one kernel per layout and dtype
        whats = outs[:nb_]
        whats_t = (outs[nb_:2 * nb_], outs[2 * nb_:] if len(outs) > 2 * nb_ else None)
    else:
        wh_all = gemm.matmul_nt(W.reshape(cout * cin, kn), basis.rho_all_t)     # [cout*cin, na]
        whats = [wh_all[:, base:base + d * d].reshape(cout, cin, d, d).permute(3, 1, 2, 0).reshape(d * cin, d * cout)
                 for d, base in zip(basis.dims, basis.bases)]
        whats_t = None
    z = _BlockGemmsFn.apply(y, basis, pts, cin, cout, whats_t, *whats)
    return FromSpectralFn.apply(z, basis, b, p, cout, out_stats)      # out_stats: (out, per-point statistics partials)


def norm_act_supported(c):
    """Channel counts the fused norm kernels take (4 channels per lane, C/4 lanes dividing a 256-thread block)."""
    return c >= 4 and c % 4 == 0 and c <= 1024 and 256 % (c // 4) == 0


class NormActFn(torch.autograd.Function):
    """y = leaky_relu(norm(x)) (+ residual) on a [b,c,p,a] tensor, norm = BatchNorm2d (groups=1, optional affine) or
    InstanceNorm2d(affine=False) (groups=b) -- `relu(norm(x))` of SPConvNets/utils/base_so3conv.py:116-126,52-62,208-211
    -- two streaming HIP passes forward, two backward (include/epn_so3conv.h, "block glue").
    Returns (y, sums) with sums[g][c] = (sum x, sum x^2) for the caller's running-statistics update."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, conv_bias, instance, eps, slope):
        ctx.set_materialize_grads(False)   # else autograd zero-fills a gradient for every non-differentiable output, every step
        lib = _lib.get_lib()
        xc = to_cl(x, "x")
        b, c, p, a = xc.shape
        groups, rows = (b, p * a) if instance else (1, b * p * a)
        st = _lib.stream_of(xc)
        dt = xc.dtype
        sums = _chan_stats(xc, groups, rows, c)
        y = empty_cl(b, c, p, a, xc.device, dt)
        g = gamma.contiguous() if gamma is not None else None
        bt = beta.contiguous() if beta is not None else None
        r = cast_feats(to_cl(residual, "residual"), dt) if residual is not None else None
        _lib.check(_entry(lib, "norm_act_fwd", dt)(_cl_ptr(xc), groups, rows, c, _lib.dev_ptr(sums, "sums"),
                                            _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta"),
                                            _cl_ptr(r) if r is not None else ctypes.c_void_p(0), float(eps),
                                            float(slope), _cl_ptr(y), st), "norm_act_fwd")
        ctx.save_for_backward(xc, sums, g, bt)
        ctx.cfg = (groups, rows, c, float(eps), float(slope), residual is not None, conv_bias is not None)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, grad_y, _grad_sums):
        if grad_y is None:
            return (None,) * 8
        xc, sums, g, bt = ctx.saved_tensors
        groups, rows, c, eps, slope, has_res, has_cb = ctx.cfg
        dy = cast_feats(to_cl(grad_y, "grad_y"), xc.dtype)
        dx, dg, db = _norm_act_backward(xc, dy, sums, g, bt, groups, rows, c, eps, slope, ctx.needs_input_grad[0])
        # conv_bias (a bias the normalisation cancels, see norm_act): exact gradient = 0
        dcb = torch.zeros(c, dtype=torch.float32, device=xc.device) if has_cb else None
        return dx, dg, db, (dy if has_res else None), dcb, None, None, None


def _pair_side(sums, g, bt, eps, instance):
    sd = _lib.NormPairSide()
    sd.sums = _lib.dev_ptr(sums, "sums")
    sd.gamma = _lib.dev_ptr(g, "gamma")
    sd.beta = _lib.dev_ptr(bt, "beta")
    sd.eps, sd.instance = float(eps), int(instance)
    return sd


class NormActPairFn(torch.autograd.Function):
    """y = leaky_relu(norm_a(xa)) + leaky_relu(norm_b(xb)): the tail of a separable block (IntraSO3Conv output with its
    InstanceNorm + the skip branch with the block's norm, SPConvNets/utils/base_so3conv.py:204-211) as ONE streaming pass
    forward and two backward (epn_norm_act_pair_*): the skip branch's normalised tensor is never written and the common
    output gradient is read once per pass.  Returns (y, sums_a, sums_b)."""

    @staticmethod
    def forward(ctx, xa, xb, gamma_a, beta_a, gamma_b, beta_b, conv_bias_b, inst_a, inst_b, eps_a, eps_b, slope,
                part_b=None, part_a=None):
        ctx.set_materialize_grads(False)   # else autograd zero-fills a gradient for every non-differentiable output, every step
        lib = _lib.get_lib()
        xac = to_cl(xa, "xa")
        xbc = to_cl(xb, "xb")
        if xbc.dtype != xac.dtype:
            xbc, part_b = cast_feats(xbc, xac.dtype), None      # statistics of the values the kernel will read
        b, c, p, a = xac.shape
        rows = p * a
        sums_a = _stats(xac, b if inst_a else 1, rows if inst_a else b * rows, c, part_a, a)   # one block per point
        sums_b = _stats(xbc, b if inst_b else 1, rows if inst_b else b * rows, c, part_b)
        ga, ba = (t.contiguous() if t is not None else None for t in (gamma_a, beta_a))
        gb, bb = (t.contiguous() if t is not None else None for t in (gamma_b, beta_b))
        y = empty_cl(b, c, p, a, xac.device, xac.dtype)
        sa, sb = _pair_side(sums_a, ga, ba, eps_a, inst_a), _pair_side(sums_b, gb, bb, eps_b, inst_b)
        if gemm.f16x2_on(y):       # the block output is the next block's GEMM operand source: its maximum from this pass
            amax = torch.empty(1, dtype=torch.float32, device=y.device)
            _lib.check(lib.epn_norm_act_pair_fwd_amax(_cl_ptr(xac), _cl_ptr(xbc), b, rows, c, ctypes.byref(sa), ctypes.byref(sb),
                                                      float(slope), _cl_ptr(y), 0, amax.data_ptr(), _lib.stream_of(xac)),
                       "norm_act_pair_fwd_amax")
            _tag_amax(y, amax)
        else:
            _lib.check(lib.epn_norm_act_pair_fwd(_cl_ptr(xac), _cl_ptr(xbc), b, rows, c, ctypes.byref(sa), ctypes.byref(sb),
                                                 float(slope), _cl_ptr(y), int(xac.dtype == torch.bfloat16),
                                                 _lib.stream_of(xac)), "norm_act_pair_fwd")
        ctx.save_for_backward(xac, xbc, sums_a, sums_b, ga, ba, gb, bb)
        ctx.cfg = (b, rows, c, bool(inst_a), bool(inst_b), float(eps_a), float(eps_b), float(slope), conv_bias_b is not None)
        ctx.mark_non_differentiable(sums_a, sums_b)
        return y, sums_a, sums_b

    @staticmethod
    def backward(ctx, grad_y, _ga, _gb):
        if grad_y is None:
            return (None,) * 14
        lib = _lib.get_lib()
        xac, xbc, sums_a, sums_b, ga, ba, gb, bb = ctx.saved_tensors
        b, rows, c, inst_a, inst_b, eps_a, eps_b, slope, has_cb = ctx.cfg
        dy = cast_feats(to_cl(grad_y, "grad_y"), xac.dtype)
        dev = xac.device
        bf = int(xac.dtype == torch.bfloat16)
        sa, sb = _pair_side(sums_a, ga, ba, eps_a, inst_a), _pair_side(sums_b, gb, bb, eps_b, inst_b)
        dsa, dsb = torch.empty_like(sums_a), torch.empty_like(sums_b)
        f32 = dict(dtype=torch.float32, device=dev)
        dga = torch.empty(c, **f32) if ga is not None else None
        dba = torch.empty(c, **f32) if ba is not None else None
        dgb = torch.empty(c, **f32) if gb is not None else None
        dbb = torch.empty(c, **f32) if bb is not None else None
        ws = torch.empty(max(int(lib.epn_norm_pair_workspace_bytes(b, rows, c)), 16), dtype=torch.uint8, device=dev)
        st = _lib.stream_of(xac)
        _lib.check(lib.epn_norm_act_pair_bwd_reduce(_cl_ptr(xac), _cl_ptr(xbc), _cl_ptr(dy), b, rows, c, ctypes.byref(sa),
                                                    ctypes.byref(sb), slope, _lib.dev_ptr(dsa, "dsums_a"),
                                                    _lib.dev_ptr(dga, "dgamma_a"), _lib.dev_ptr(dba, "dbeta_a"),
                                                    _lib.dev_ptr(dsb, "dsums_b"), _lib.dev_ptr(dgb, "dgamma_b"),
                                                    _lib.dev_ptr(dbb, "dbeta_b"), ctypes.c_void_p(ws.data_ptr()),
                                                    ctypes.c_size_t(ws.numel()), bf, st), "norm_act_pair_bwd_reduce")
        dxa = torch.empty_like(xac) if ctx.needs_input_grad[0] else None
        dxb = torch.empty_like(xbc) if ctx.needs_input_grad[1] else None
        if dxa is not None or dxb is not None:
            pa = _cl_ptr(dxa) if dxa is not None else ctypes.c_void_p(0)
            pb = _cl_ptr(dxb) if dxb is not None else ctypes.c_void_p(0)
            if dxb is not None and gemm.f16x2_on(dxb):      # side b's gradient feeds the skip convolution's backward GEMMs
                amax = torch.empty(1, dtype=torch.float32, device=dxb.device)
                _lib.check(lib.epn_norm_act_pair_bwd_apply_amax(_cl_ptr(xac), _cl_ptr(xbc), _cl_ptr(dy), b, rows, c, ctypes.byref(sa),
                                                                ctypes.byref(sb), slope, _lib.dev_ptr(dsa, "dsums_a"),
                                                                _lib.dev_ptr(dsb, "dsums_b"), pa, pb, bf, amax.data_ptr(),
                                                                _lib.stream_of(xac)), "norm_act_pair_bwd_apply_amax")
                _tag_amax(dxb, amax)
            else:
                _lib.check(lib.epn_norm_act_pair_bwd_apply(_cl_ptr(xac), _cl_ptr(xbc), _cl_ptr(dy), b, rows, c, ctypes.byref(sa),
                                                           ctypes.byref(sb), slope, _lib.dev_ptr(dsa, "dsums_a"),
                                                           _lib.dev_ptr(dsb, "dsums_b"), pa, pb, bf,
                                                           _lib.stream_of(xac)), "norm_act_pair_bwd_apply")
        dcb = torch.zeros(c, **f32) if has_cb else None       # a bias the normalisation cancels: exact gradient 0
        return dxa, dxb, dga, dba, dgb, dbb, dcb, None, None, None, None, None, None, None


def norm_act_pair(xa, norm_a, xb, norm_b, conv_bias_b=None, slope=0.01, part_b=None, part_a=None):
    """leaky_relu(norm_a(xa)) + leaky_relu(norm_b(xb)) in one pass (training mode); norm_* an nn.BatchNorm2d or
    nn.InstanceNorm2d(affine=False) whose parameters / running statistics are used and updated as the module would.
    conv_bias_b: bias of the convolution that produced xb, NOT added to xb (see norm_act).  part_b: block partials of xb's
    per-channel statistics from the epilogue of the GEMM that produced it (conv1x1(..., col_stats=True)); part_a: per-point
    partials of xa's from the inverse basis change (intra_so3conv(..., out_stats=True))."""
    import torch.nn as nn
    ia, ib = isinstance(norm_a, nn.InstanceNorm2d), isinstance(norm_b, nn.InstanceNorm2d)
    y, sums_a, sums_b = NormActPairFn.apply(xa, xb, getattr(norm_a, "weight", None), getattr(norm_a, "bias", None),
                                            getattr(norm_b, "weight", None), getattr(norm_b, "bias", None), conv_bias_b,
                                            ia, ib, norm_a.eps, norm_b.eps, slope, part_b, part_a)
    n = xa.shape[0] * xa.shape[2] * xa.shape[3]
    _update_running_stats(norm_a, sums_a, n)
    _update_running_stats(norm_b, sums_b, n, conv_bias_b)
    return y


def _chan_stats(xc, groups, rows, c):
    """sums[g][c] = (sum x, sum x^2) of a channels-last tensor (epn_chan_stats_*: block partials + finishing kernel)."""
    lib = _lib.get_lib()
    sums = torch.empty((groups, c, 2), dtype=torch.float32, device=xc.device)
    ws = torch.empty(max(int(lib.epn_norm_workspace_bytes(groups, rows, c)), 16), dtype=torch.uint8, device=xc.device)
    _lib.check(_entry(lib, "chan_stats", xc.dtype)(_cl_ptr(xc), groups, rows, c, _lib.dev_ptr(sums, "sums"),
                                                   ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                                                   _lib.stream_of(xc)), "chan_stats")
    return sums


def sums_from_partials(part, groups, rows, c, block_rows=32):
    """sums[g][c][2] from the [rows_total / block_rows, c, 2] block partials a producer wrote from its accumulators
    (gemm_nt(..., col_stats=True): 32-row blocks; the inverse basis change: one block per point = na rows); None when they
    do not apply (no partials, or a group's rows are not whole blocks)."""
    if part is None or part.numel() == 0 or rows % block_rows or part.numel() != groups * (rows // block_rows) * c * 2:
        return None
    lib = _lib.get_lib()
    nb = rows // block_rows
    sums = torch.empty((groups, c, 2), dtype=torch.float32, device=part.device)
    ws = torch.empty(max(int(lib.epn_stats_finish_workspace_bytes(groups, nb, c)), 16), dtype=torch.uint8, device=part.device)
    _lib.check(lib.epn_stats_finish(part.data_ptr(), groups, nb, c, sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _lib.stream_of(part)), "stats_finish")
    return sums


def _stats(xc, groups, rows, c, part=None, block_rows=32):
    """Per-channel statistics of xc: finished from its producer's epilogue partials when there are any, else a pass over xc."""
    sums = sums_from_partials(part, groups, rows, c, block_rows) if part is not None else None
    return sums if sums is not None else _chan_stats(xc, groups, rows, c)


def _norm_act_backward(xc, dy, sums, g, bt, groups, rows, c, eps, slope, need_dx):
    """Backward of y = leaky(norm(x)): (dx, dgamma, dbeta) from x, dy and the forward statistics (two streaming passes)."""
    lib = _lib.get_lib()
    st = _lib.stream_of(xc)
    dsums = torch.empty_like(sums)
    dg = torch.empty(c, dtype=torch.float32, device=xc.device) if g is not None else None
    db = torch.empty(c, dtype=torch.float32, device=xc.device) if bt is not None else None
    gp, bp = _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta")
    ws = torch.empty(max(int(lib.epn_norm_workspace_bytes(groups, rows, c)), 16), dtype=torch.uint8, device=xc.device)
    _lib.check(_entry(lib, "norm_act_bwd_reduce", xc.dtype)(_cl_ptr(xc), _cl_ptr(dy), groups, rows, c,
                                               _lib.dev_ptr(sums, "sums"), gp, bp, eps, slope,
                                               _lib.dev_ptr(dsums, "dsums"), _lib.dev_ptr(dg, "dgamma"),
                                               _lib.dev_ptr(db, "dbeta"), ctypes.c_void_p(ws.data_ptr()),
                                               ctypes.c_size_t(ws.numel()), st), "norm_act_bwd_reduce")
    dx = None
    if need_dx:
        dx = torch.empty_like(xc)
        _lib.check(_entry(lib, "norm_act_bwd_apply", xc.dtype)(_cl_ptr(xc), _cl_ptr(dy), groups, rows, c,
                                                  _lib.dev_ptr(sums, "sums"), _lib.dev_ptr(dsums, "dsums"), gp, bp,
                                                  eps, slope, _cl_ptr(dx), st), "norm_act_bwd_apply")
    return dx, dg, db


class NormToSpectralFn(torch.autograd.Function):
    """ToSpectral(leaky_relu(norm(x))) with the normalisation applied as the basis-change kernel loads its rows
    (epn_so3_basis_norm_*): the normalised tensor is never written or re-read (SURVEY 8f.1 "norm on load").  Backward:
    inverse transform of the spectral gradient, then the two norm passes of NormActFn.  Returns (y_spectral, sums)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, conv_bias, instance, eps, slope, basis, part=None):
        ctx.set_materialize_grads(False)   # else autograd zero-fills a gradient for every non-differentiable output, every step
        lib = _lib.get_lib()
        xc = to_cl(x, "x")
        b, c, p, na = xc.shape
        groups, rows = (b, p * na) if instance else (1, b * p * na)
        sums = _stats(xc, groups, rows, c, part)           # part: block partials from the epilogue of x's producer
        g = gamma.contiguous() if gamma is not None else None
        bt = beta.contiguous() if beta is not None else None
        y = torch.empty(na * b * p * c, dtype=xc.dtype, device=xc.device)
        if gemm.f16x2_on(y):                                 # + max|y| for the two-piece GEMMs that read it (see _basis_call)
            amax = torch.empty(1, dtype=torch.float32, device=y.device)
            _lib.check(_launch("so3_basis", ("so3_basis", b * p, c), 2.0 * b * p * na * na * c, xc.device,
                               lambda: lib.epn_so3_basis_norm_amax_split_f32(
                                   _cl_ptr(xc), _lib.dev_ptr(basis.Ut, "M"), _lib.dev_ptr(basis.blocks, "blocks", torch.int32),
                                   ctypes.c_longlong(b * p), na, c, 1, ctypes.c_void_p(y.data_ptr()), _lib.dev_ptr(sums, "sums"),
                                   groups, ctypes.c_longlong(p), _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta"), float(eps),
                                   float(slope), amax.data_ptr(), _lib.stream_of(xc))), "so3_basis_norm_amax")
            _tag_amax(y, amax)
        else:
            fn = _entry(lib, "so3_basis_norm", xc.dtype)
            _lib.check(_launch("so3_basis", ("so3_basis", b * p, c), 2.0 * b * p * na * na * c, xc.device,
                               lambda: fn(_cl_ptr(xc), _lib.dev_ptr(basis.Ut, "M"), _lib.dev_ptr(basis.blocks, "blocks", torch.int32),
                                          ctypes.c_longlong(b * p), na, c, 1, ctypes.c_void_p(y.data_ptr()),
                                          _lib.dev_ptr(sums, "sums"), groups, ctypes.c_longlong(p),
                                          _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta"), float(eps), float(slope),
                                          _lib.stream_of(xc))), "so3_basis_norm")
        ctx.save_for_backward(xc, sums, g, bt)
        ctx.basis = basis
        ctx.cfg = (groups, rows, c, float(eps), float(slope), conv_bias is not None, (b, c, p, na))
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, gy, _grad_sums):
        if gy is None:
            return (None,) * 9
        lib = _lib.get_lib()
        xc, sums, g, bt = ctx.saved_tensors
        groups, rows, c, eps, slope, has_cb, (b, _, p, na) = ctx.cfg
        gf = empty_cl(b, c, p, na, gy.device, xc.dtype)
        fused = (ab("EPN_NORM_BWD_EPILOGUE") == "1" and ctx.needs_input_grad[0]
                 and xc.numel() * xc.element_size() < 0x7fffff00 and b * p < (1 << 24))
        if fused:
            # the norm's backward reduction from the accumulators of the inverse basis change that produces its output gradient
            # (epn_so3_basis_dstats_*): no norm_act_bwd_reduce pass over x and dy
            gyc = cast_feats(gy.contiguous(), xc.dtype)
            pd = torch.empty((b * p, c, 2), dtype=torch.float32, device=gy.device)
            fn = _entry(lib, "so3_basis_dstats", xc.dtype)
            _lib.check(_launch("so3_basis", ("so3_basis", b * p, c), 2.0 * b * p * na * na * c, gy.device,
                               lambda: fn(ctypes.c_void_p(gyc.data_ptr()), _lib.dev_ptr(ctx.basis.U, "M"),
                                          _lib.dev_ptr(ctx.basis.blocks, "blocks", torch.int32), ctypes.c_longlong(b * p), na, c,
                                          _cl_ptr(gf), _cl_ptr(xc), _lib.dev_ptr(sums, "sums"), groups, ctypes.c_longlong(p),
                                          _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta"), float(eps), float(slope),
                                          pd.data_ptr(), _lib.stream_of(gf))), "so3_basis_dstats")
            dsums = torch.empty_like(sums)
            dg = torch.empty(c, dtype=torch.float32, device=gy.device) if g is not None else None
            db = torch.empty(c, dtype=torch.float32, device=gy.device) if bt is not None else None
            bpg = b * p // groups
            ws = torch.empty(max(int(lib.epn_stats_finish_workspace_bytes(groups, bpg, c)), 16), dtype=torch.uint8, device=gy.device)
            _lib.check(lib.epn_norm_bwd_finish(pd.data_ptr(), groups, ctypes.c_longlong(bpg), c, _lib.dev_ptr(g, "gamma"),
                                               _lib.dev_ptr(dsums, "dsums"), _lib.dev_ptr(dg, "dgamma"), _lib.dev_ptr(db, "dbeta"),
                                               ws.data_ptr(), ws.numel(), _lib.stream_of(gf)), "norm_bwd_finish")
            dx = torch.empty_like(xc)
            if gemm.f16x2_on(dx):      # dx is the inter convolution's output gradient: the narrow operand of two backward GEMMs
                amax = torch.empty(1, dtype=torch.float32, device=dx.device)
                _lib.check(lib.epn_norm_act_bwd_apply_amax_f32(_cl_ptr(xc), _cl_ptr(gf), groups, rows, c,
                                                               _lib.dev_ptr(sums, "sums"), _lib.dev_ptr(dsums, "dsums"),
                                                               _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta"), eps, slope,
                                                               _cl_ptr(dx), amax.data_ptr(), _lib.stream_of(gf)),
                           "norm_act_bwd_apply_amax")
                _tag_amax(dx, amax)
            else:
                _lib.check(_entry(lib, "norm_act_bwd_apply", xc.dtype)(_cl_ptr(xc), _cl_ptr(gf), groups, rows, c,
                                                                      _lib.dev_ptr(sums, "sums"), _lib.dev_ptr(dsums, "dsums"),
                                                                      _lib.dev_ptr(g, "gamma"), _lib.dev_ptr(bt, "beta"), eps, slope,
                                                                      _cl_ptr(dx), _lib.stream_of(gf)), "norm_act_bwd_apply")
            dcb = torch.zeros(c, dtype=torch.float32, device=xc.device) if has_cb else None
            return dx, dg, db, dcb, None, None, None, None, None
        _basis_call(lib, cast_feats(gy.contiguous(), xc.dtype), ctx.basis.U, ctx.basis, b * p, c, 1, 0, gf, "so3_basis")
        dx, dg, db = _norm_act_backward(xc, gf, sums, g, bt, groups, rows, c, eps, slope, ctx.needs_input_grad[0])
        dcb = torch.zeros(c, dtype=torch.float32, device=xc.device) if has_cb else None
        return dx, dg, db, dcb, None, None, None, None, None


def _update_running_stats(norm, sums, n, conv_bias=None):
    """BatchNorm's running statistics from (sum x, sum x^2), exactly as the torch module updates them in training mode."""
    import torch.nn as nn
    if isinstance(norm, nn.InstanceNorm2d) or not norm.track_running_stats or norm.running_mean is None:
        return
    c = norm.running_mean.numel()
    rm, rv, nb = norm.running_mean, norm.running_var, norm.num_batches_tracked
    if (sums.is_cuda and c <= 1024 and rm.dtype == torch.float32 and rm.is_contiguous() and rv.is_contiguous()
            and nb.dtype == torch.int64 and sums.dtype == torch.float32 and sums.is_contiguous()):
        # one launch instead of ~10 elementwise torch kernels on c-sized vectors
        with torch.no_grad():
            bias = conv_bias.detach().float().contiguous() if conv_bias is not None else None
            _lib.check(_lib.get_lib().epn_bn_running_update_f32(
                sums.data_ptr(), float(n), bias.data_ptr() if bias is not None else None, rm.data_ptr(), rv.data_ptr(),
                nb.data_ptr(), float(norm.momentum) if norm.momentum is not None else -1.0, c, _lib.stream_of(sums)),
                "bn_running_update")
        return
    with torch.no_grad():
        mean = sums[0, :, 0] / n
        var = (sums[0, :, 1] / n - mean * mean).clamp_min_(0) * (n / max(n - 1, 1))   # unbiased, as BatchNorm stores
        if conv_bias is not None:
            mean = mean + conv_bias
        norm.num_batches_tracked += 1
        # momentum=None: cumulative moving average, as torch.nn.modules.batchnorm._BatchNorm.forward
        if norm.momentum is not None:
            norm.running_mean.mul_(1 - norm.momentum).add_(mean, alpha=norm.momentum)
            norm.running_var.mul_(1 - norm.momentum).add_(var, alpha=norm.momentum)
        else:                                                  # device-side weight: no host sync (graph capture)
            m = norm.num_batches_tracked.to(mean.dtype).reciprocal()
            norm.running_mean.lerp_(mean, m)
            norm.running_var.lerp_(var, m)


def norm_act(x, norm, residual=None, slope=0.01, conv_bias=None):
    """leaky_relu(norm(x)) (+ residual) with `norm` an nn.BatchNorm2d or nn.InstanceNorm2d(affine=False) module whose
    parameters / running statistics are used and updated exactly as the module would (training mode).
    conv_bias: per-channel bias of the convolution that produced x, NOT yet added to x.  Both norms subtract the
    per-channel mean, so norm(x + b) == norm(x) and d/db == 0 exactly: the add (a full read + write of the tensor) and
    the bias-gradient reduction are skipped, only BatchNorm's running_mean needs b."""
    import torch.nn as nn
    instance = isinstance(norm, nn.InstanceNorm2d)
    gamma = getattr(norm, "weight", None)
    beta = getattr(norm, "bias", None)
    y, sums = NormActFn.apply(x, gamma, beta, residual, conv_bias, instance, norm.eps, slope)
    _update_running_stats(norm, sums, x.shape[0] * x.shape[2] * x.shape[3], conv_bias)
    return y


def deterministic_bwd(dtype):
    """Atomic-free, bitwise repeatable InterSO3Conv data gradient?  EPN_DETERMINISTIC = 1 | 0 (default 0).  Measured on
    MI355X: the per-slot slab + ordered reduction costs 12 % of a bf16 step (rotation network, 673 vs 767 clouds/s) and
    9 % of an fp32 step (classification network, 231 vs 254 clouds/s) against the fp32 atomic scatter -- the slab is K
    times the size of the gradient it reduces to -- so it is an option, not the default."""
    return os.environ.get("EPN_DETERMINISTIC", "0") == "1"


def group_packed():
    """Grouped features of the split convolution in the packed column order (epn_inter_group_packed_*)?  EPN_GROUP_PACKED =
    1 | 0 (default 1).  The grouping kernel's stores become contiguous (8-20 % less kernel time per layer, measured on MI355X);
    the weights are permuted to match, so the product is the same up to fp32 summation order."""
    return ab("EPN_GROUP_PACKED") != "0"


def inter_mode():
    """EPN_INTER_MODE = fused | split | auto (default).  auto: the split form (grouped features to HBM + library GEMMs)
    for every layer the MFMA grouping kernel takes (cin % 16 == 0) -- measured faster for training and for inference;
    the fused kernels are the memory-lean choice (no [cols, cin*ks] tensor) and serve cin = 1 and dense inter_w."""
    return os.environ.get("EPN_INTER_MODE", "auto")


def inter_so3conv(feats, W, geo, out_dtype=None, share_input=False):
    """InterSO3Conv's compute.  out_dtype (default: the dtype of feats) lets the fp32 first layer (cin = 1, all-ones
    occupancy features) hand bf16 features to the rest of a bf16 network.  share_input: return (out, feats', part) where
    feats' is `feats` for the caller's OTHER uses of it (InterSO3ConvSplitFn.forward: its gradient is then folded into the
    convolution's own data gradient; the forms without that fold return `feats` itself) and part the block partials of
    out's per-channel statistics from the GEMM epilogue (or None)."""
    if share_input:
        mode = inter_mode()
        plain = (feats.shape[1] % 16 != 0 or isinstance(geo, DenseInterWeights) or not feats.is_cuda or mode not in ("auto", "split")
                 or feats.dtype not in FEATURE_DTYPES or ab("EPN_SHARE_INPUT_GRAD") != "1")
        if not plain and feats.dtype == torch.float32:
            g_bytes = (geo.ball_idx.shape[0] * geo.ball_idx.shape[1] * geo.anchors.shape[0] * feats.shape[1] *
                       geo.kernels.shape[0] * 4)
            plain = mode == "auto" and g_bytes > _device_bytes(feats.device) // 8
        if plain:
            return inter_so3conv(feats, W, geo, out_dtype), feats, None
        if not feats.requires_grad:
            # nothing flows back into `feats` (frozen / detached trunk): handing it out as a differentiable output would make
            # it require grad through W and cost a discarded dA GEMM (+ scatter) in the skip branch's backward
            out, part = InterSO3ConvSplitFn.apply(feats, W, geo, "stats")
            shared = feats
        else:
            out, shared, part = InterSO3ConvSplitFn.apply(feats, W, geo, True)
            if out.grad_fn is not None:
                out.grad_fn.shared_ref = weakref.ref(shared)    # backward(): is anybody watching this tensor's gradient?
        if (out_dtype or feats.dtype) != out.dtype:
            return cast_feats(out, out_dtype), shared, None      # statistics must be those of the values handed on
        return out, shared, (part if part.numel() else None)
    mode = inter_mode()
    out_dtype = out_dtype or feats.dtype
    split_ok = feats.shape[1] % 16 == 0 and not isinstance(geo, DenseInterWeights)
    if mode == "onchip" and split_ok and inter_onchip_ok(feats, W, geo):
        return cast_feats(InterSO3ConvOnChipFn.apply(feats, W, geo), out_dtype)
    if feats.dtype == torch.bfloat16 and split_ok:
        return cast_feats(InterSO3ConvSplitFn.apply(feats, W, geo), out_dtype)
    if feats.dtype != torch.float32:         # shapes only the fp32 fused / generic kernels take
        feats = cast_feats(feats, torch.float32)
    if split_ok and mode == "auto" and feats.is_cuda:
        # the split form keeps grouped[cols, cin*ks] (and its gradient) alive: 6 GB per layer at B=32 on the ModelNet
        # schedule.  A layer whose grouped tensor would take more than an eighth of the device falls back to the fused
        # kernels, which never materialise it (small-memory parts, very large batches).
        d_b, d_p2 = geo.ball_idx.shape[0], geo.ball_idx.shape[1]
        g_bytes = d_b * d_p2 * geo.anchors.shape[0] * feats.shape[1] * geo.kernels.shape[0] * 4
        if g_bytes > _device_bytes(feats.device) // 8:
            split_ok = False
    if split_ok and mode in ("split", "auto"):
        return cast_feats(InterSO3ConvSplitFn.apply(feats, W, geo), out_dtype)
    return cast_feats(InterSO3ConvFn.apply(feats, W, geo), out_dtype)


_DEVICE_BYTES = {}


def _device_bytes(device):
    key = torch.device(device).index or 0
    if key not in _DEVICE_BYTES:
        _DEVICE_BYTES[key] = torch.cuda.get_device_properties(key).total_memory
    return _DEVICE_BYTES[key]


def intra_mode():
    """EPN_INTRA_MODE = fused | split | spectral | auto (default).  auto: the block-diagonal ("spectral") form when both
    widths are multiples of 32 and the index table is a regular group action, else the split form (gather kernel +
    library GEMMs) for multiples of 16, else the fused / generic kernels."""
    return os.environ.get("EPN_INTRA_MODE", "auto")


def intra_so3conv_fused(feats, W, intra_idx32):
    return IntraSO3ConvFn.apply(feats, W, intra_idx32)


def intra_takes_spectral(cin, cout, intra_idx32, is_cuda=True):
    """Will intra_so3conv run the block-diagonal form for these widths / this table?  (Then a preceding norm + leaky_relu
    can be folded into its basis change: intra_so3conv(..., pre_norm=).)"""
    return (intra_mode() in ("auto", "spectral") and is_cuda and cin % 32 == 0 and cout % 32 == 0
            and intra_idx32.shape[1] > 1 and spectral_basis(intra_idx32) is not None)


def intra_so3conv(feats, W, intra_idx32, pre_norm=None, pre_part=None, out_stats=False):
    """pre_norm: an nn.BatchNorm2d / nn.InstanceNorm2d(affine=False) whose leaky_relu(norm(feats)) is the actual input
    (training mode); only with intra_takes_spectral(...) -- other forms get the normalised tensor from ops.norm_act.
    pre_part: block partials of feats' per-channel statistics from its producer's epilogue (spectral form only).
    out_stats: return (out, part) -- part = per-point partials of out's statistics (spectral form) or None."""
    if out_stats:
        mode = intra_mode()
        cin, cout = feats.shape[1], W.shape[0]
        if mode in ("auto", "spectral") and feats.is_cuda and cin % 32 == 0 and cout % 32 == 0 and intra_idx32.shape[1] > 1:
            basis = spectral_basis(intra_idx32)
            if basis is not None:
                return intra_so3conv_spectral(feats, W, intra_idx32, basis, pre_norm=pre_norm, pre_part=pre_part,
                                              out_stats=True)
        return intra_so3conv(feats, W, intra_idx32, pre_norm, pre_part), None
    mode = intra_mode()
    cin, cout = feats.shape[1], W.shape[0]
    bf = feats.dtype == torch.bfloat16
    if mode in ("auto", "spectral") and feats.is_cuda and cin % 32 == 0 and cout % 32 == 0 and intra_idx32.shape[1] > 1:
        basis = spectral_basis(intra_idx32)
        if basis is not None:
            return intra_so3conv_spectral(feats, W, intra_idx32, basis, pre_norm=pre_norm, pre_part=pre_part)
    if pre_norm is not None:
        feats = norm_act(feats, pre_norm)
    if (bf and cin % 8 == 0 and cout % 8 == 0) or mode == "split" or \
            (mode in ("auto", "spectral") and cin % 16 == 0 and cout % 16 == 0):
        return IntraSO3ConvSplitFn.apply(feats, W, intra_idx32)
    if bf:                                   # odd widths: fp32 kernels between two casts
        return cast_feats(IntraSO3ConvFn.apply(cast_feats(feats, torch.float32), W, intra_idx32), torch.bfloat16)
    return IntraSO3ConvFn.apply(feats, W, intra_idx32)


class PointnetSO3ConvFn(torch.autograd.Function):
    """max_p(embed(cat(feats, R_a^T (xyz - mean)))) of PointnetSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:219-235)
    in one fused HIP pass; returns [b, co, a] (a channels-last view of the [b][a][co] buffer the kernel writes).
    Gradients flow to feats / weight / bias through the arg-max point, as torch.max's backward does."""

    @staticmethod
    def forward(ctx, feats, xyz, anchors, weight, bias):
        lib = _lib.get_lib()
        fc = to_cl(feats, "feats")
        if fc.dtype != torch.float32:
            raise TypeError("PointnetSO3ConvFn takes fp32 features (pointnet_so3conv casts bf16 ones)")
        b, c, p, a = fc.shape
        co = weight.shape[0]
        if weight.numel() != co * (c + 3):
            raise ValueError(f"embed weight must be [co, c+3, 1, 1] with c={c}, got {tuple(weight.shape)}")
        if tuple(xyz.shape) != (b, 3, p):
            raise ValueError(f"xyz must be [b,3,p]=({b},3,{p}), got {tuple(xyz.shape)}")
        w = weight.reshape(co, c + 3).contiguous()
        xyz = xyz.contiguous()
        anc = anchors.contiguous() if (anchors is not None and a > 1) else None
        out = torch.empty((b, a, co), dtype=torch.float32, device=fc.device)
        arg = torch.empty((b, a, co), dtype=torch.int32, device=fc.device)
        ctr = torch.empty((b, 3), dtype=torch.float32, device=fc.device)
        bs = bias.contiguous() if bias is not None else None
        _launch("pointnet_fwd", ("pointnet", b, p, a, c, co), 2.0 * b * p * a * co * (c + 3), fc.device,
                lambda: _lib.check(lib.epn_pointnet_so3conv_fwd_f32(
                    _cl_ptr(fc), _lib.dev_ptr(xyz, "xyz"), _lib.dev_ptr(anc, "anchors"), _lib.dev_ptr(w, "weight"),
                    _lib.dev_ptr(bs, "bias"), _lib.dev_ptr(out, "out"), _lib.dev_ptr(arg, "argmax", torch.int32),
                    _lib.dev_ptr(ctr, "centre"), b, p, a, c, co, _lib.stream_of(fc)), "pointnet_so3conv_fwd"))
        ctx.save_for_backward(fc, xyz, anc, w, arg, ctr)
        ctx.cfg = (b, p, a, c, co, tuple(weight.shape), bias is not None)
        return out.permute(0, 2, 1)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.get_lib()
        fc, xyz, anc, w, arg, ctr = ctx.saved_tensors
        b, p, a, c, co, wshape, has_bias = ctx.cfg
        g = grad_out.permute(0, 2, 1).contiguous()           # [b][a][co]
        st = _lib.stream_of(fc)
        dF = dW = db = None
        if ctx.needs_input_grad[0]:
            dF = torch.empty_like(fc)
            _launch("pointnet_bwd_data", ("pointnet", b, p, a, c, co), 2.0 * b * a * co * c, fc.device,
                    lambda: _lib.check(lib.epn_pointnet_so3conv_bwd_data_f32(
                        _lib.dev_ptr(g, "grad_out"), _lib.dev_ptr(arg, "argmax", torch.int32), _lib.dev_ptr(w, "weight"),
                        _cl_ptr(dF), b, p, a, c, co, st), "pointnet_so3conv_bwd_data"))
        if ctx.needs_input_grad[3] or (has_bias and ctx.needs_input_grad[4]):
            dW = torch.empty((co, c + 3), dtype=torch.float32, device=fc.device)
            db = torch.empty(co, dtype=torch.float32, device=fc.device) if has_bias else None
            _launch("pointnet_bwd_weight", ("pointnet", b, p, a, c, co), 2.0 * b * a * co * (c + 3), fc.device,
                    lambda: _lib.check(lib.epn_pointnet_so3conv_bwd_weight_f32(
                        _lib.dev_ptr(g, "grad_out"), _lib.dev_ptr(arg, "argmax", torch.int32), _cl_ptr(fc),
                        _lib.dev_ptr(xyz, "xyz"), _lib.dev_ptr(anc, "anchors"), _lib.dev_ptr(ctr, "centre"),
                        _lib.dev_ptr(dW, "grad_W"), _lib.dev_ptr(db, "grad_bias"), b, p, a, c, co, st),
                        "pointnet_so3conv_bwd_weight"))
            dW = dW.view(wshape)
        return dF, None, None, dW, db


_IDENT = {}


def _identity_index(na, device):
    """[na, 1] identity neighbour table, one tensor per (na, device): the tables derived from an index tensor are
    cached per tensor."""
    key = (na, str(device))
    if key not in _IDENT:
        _IDENT[key] = torch.arange(na, dtype=torch.int32, device=device).view(na, 1)
    return _IDENT[key]


class GatherRowsFn(torch.autograd.Function):
    """batched_index_select(feats, 2, sample_idx) of the strided skip connection (SPConvNets/utils/base_so3conv.py:206-207)
    on channels-last data: whole [a][c] rows move (epn_gather_rows); backward adds them back into a zeroed tensor
    (epn_scatter_rows_add: FPS indices are distinct for ordinary clouds, but repeat index 0 when a cloud has fewer live
    points than samples -- those rows must receive the SUM, as torch.gather's backward gives)."""

    @staticmethod
    def forward(ctx, feats, sample_idx):
        lib = _lib.get_lib()
        f = to_cl(feats)
        b, c, p1, a = f.shape
        idx = sample_idx.int().contiguous()
        p2 = idx.shape[1]
        out = empty_cl(b, c, p2, a, f.device, f.dtype)
        _lib.check(lib.epn_gather_rows(_cl_ptr(f), _lib.dev_ptr(idx, "sample_idx", torch.int32), _cl_ptr(out), b, p1, p2,
                                       a * c * f.element_size(), _lib.stream_of(f)), "gather_rows")
        ctx.save_for_backward(idx)
        ctx.dims = (b, c, p1, a)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.get_lib()
        (idx,) = ctx.saved_tensors
        b, c, p1, a = ctx.dims
        gc = to_cl(g, "grad")
        gs = empty_cl(b, c, p1, a, gc.device, gc.dtype)
        _lib.check(lib.epn_scatter_rows_add(_cl_ptr(gc), _lib.dev_ptr(idx, "sample_idx", torch.int32), _cl_ptr(gs), b, p1,
                                            idx.shape[1], a * c * gc.element_size(), int(gc.dtype == torch.bfloat16),
                                            _lib.stream_of(gc)), "scatter_rows_add")
        return gs, None


def gather_rows(feats, sample_idx):
    """Rows of a [b,c,p,a] tensor selected along p; needs a*c*elem_size % 16 == 0 (else torch.gather)."""
    if feats.is_cuda and (feats.shape[1] * feats.shape[3] * feats.element_size()) % 16 == 0 and \
            feats.dtype in FEATURE_DTYPES and sample_idx.shape[1] <= 8192:
        return GatherRowsFn.apply(feats, sample_idx)
    idx = sample_idx.long().view(feats.shape[0], 1, -1, 1).expand(-1, feats.shape[1], -1, feats.shape[3])
    return torch.gather(feats, 2, idx)


class AnchorSoftmaxPoolFn(torch.autograd.Function):
    """(attn, pooled) = (softmax(logits, dim=3), (feats * attn).sum(3)) of InvOutBlockMVD.forward
    (SPConvNets/utils/base_so3conv.py:603-606) in one pass over the channels-last tensors, and one pass backward
    (epn_anchor_softmax_pool_{fwd,bwd}_f32).  feats, logits [b, c, p, a] fp32 -> attn [b, c, p, a], pooled [b, c, p, 1]."""

    @staticmethod
    def forward(ctx, feats, logits):
        ctx.set_materialize_grads(False)
        lib = _lib.get_lib()
        f, l = to_cl(feats), to_cl(logits, "logits")
        b, c, p, a = f.shape
        attn = empty_cl(b, c, p, a, f.device)
        pooled = torch.empty((b, p, c), dtype=torch.float32, device=f.device)
        _lib.check(lib.epn_anchor_softmax_pool_fwd_f32(_cl_ptr(f), _cl_ptr(l), _cl_ptr(attn), pooled.data_ptr(), b * p, a, c,
                                                       _lib.stream_of(f)), "anchor_softmax_pool_fwd")
        ctx.save_for_backward(f, attn)
        return attn, pooled.permute(0, 2, 1).unsqueeze(-1)

    @staticmethod
    def backward(ctx, g_attn, g_pooled):
        f, attn = ctx.saved_tensors
        if g_attn is None and g_pooled is None:
            return None, None
        lib = _lib.get_lib()
        b, c, p, a = f.shape
        ga = to_cl(g_attn.float(), "grad_attn") if g_attn is not None else None
        gp = g_pooled.float().reshape(b, c, p).permute(0, 2, 1).contiguous() if g_pooled is not None else None
        gf = empty_cl(b, c, p, a, f.device) if ctx.needs_input_grad[0] and gp is not None else None
        gl = empty_cl(b, c, p, a, f.device)
        _lib.check(lib.epn_anchor_softmax_pool_bwd_f32(_cl_ptr(f), _cl_ptr(attn), gp.data_ptr() if gp is not None else None,
                                                       _cl_ptr(ga) if ga is not None else None,
                                                       _cl_ptr(gf) if gf is not None else None, _cl_ptr(gl), b * p, a, c,
                                                       _lib.stream_of(f)), "anchor_softmax_pool_bwd")
        return gf, gl


def anchor_softmax_pool(feats, logits):
    """-> (attn [b,c,p,a], pooled [b,c,p,1]): the 3DMatch head's attention pooling over the anchors; CUDA fp32 tensors with at
    most 64 anchors run the fused kernels, anything else the reference's torch composition."""
    if (feats.is_cuda and feats.dtype == torch.float32 and logits.dtype == torch.float32 and feats.dim() == 4
            and feats.shape == logits.shape and feats.shape[3] <= 64 and ab("EPN_ATTN_POOL") == "1"):
        return AnchorSoftmaxPoolFn.apply(feats, logits)
    attn = torch.nn.functional.softmax(logits, dim=3)
    return attn, (feats * attn).sum(-1, keepdim=True)


class Conv1x1C1Fn(torch.autograd.Function):
    """nn.Conv2d(1, cout, 1) on channels-last rows: y[row][c] = x[row] * w[c] (epn_conv1x1_c1_f32); the weight gradient is
    one streaming reduction (epn_conv1x1_c1_bwd_weight_f32).  x is the occupancy feature of the first block (an input)."""

    @staticmethod
    def forward(ctx, x_rows, w):
        lib = _lib.get_lib()
        x_rows, w = x_rows.contiguous(), w.contiguous()
        y = torch.empty((x_rows.numel(), w.numel()), dtype=torch.float32, device=x_rows.device)
        _lib.check(lib.epn_conv1x1_c1_f32(x_rows.data_ptr(), w.data_ptr(), y.data_ptr(), x_rows.numel(), w.numel(),
                                          _lib.stream_of(x_rows)), "conv1x1_c1")
        ctx.save_for_backward(x_rows, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_rows, w = ctx.saved_tensors
        lib = _lib.get_lib()
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            _lib.check(lib.epn_conv1x1_c1_bwd_weight_f32(x_rows.data_ptr(), gy.data_ptr(), gw.data_ptr(), x_rows.numel(),
                                                         w.numel(), _lib.stream_of(gy)), "conv1x1_c1_bwd_weight")
        if ctx.needs_input_grad[0]:
            gx = gy @ w
        return gx, gw


def conv1x1(x, weight, bias=None, col_stats=False, x_amax=None):
    """nn.Conv2d(cin, cout, 1) on a [b,c,p,a] tensor, channels-last in and out with no layout copy: one NT GEMM
    [cols, cin] x [cout, cin]^T on the zero-copy 2-D view (fp32 master weight, cast per call for bf16 features).
    bf16 widths that are only multiples of 16 run the fp32 intra GEMM kernel (single, identity anchor neighbour);
    cin = 1 (the occupancy feature of the first block) is an outer product; odd shapes go to torch.
    col_stats=True (bias must be None): returns (y, part) -- part = the block partials of y's per-channel statistics from the
    GEMM's epilogue (sums_from_partials / norm_act_pair(part_b=...)), or None on the paths that do not produce them.
    x_amax: device scalar >= max|x| when the caller has one (two-piece fp16 GEMMs, gemm.absmax)."""
    cout, cin = weight.shape[0], weight.shape[1]
    part = None
    if x.is_cuda and ((x.dtype == torch.bfloat16 and cin % 32 == 0) or (x.dtype == torch.float32 and cin % 16 == 0)):
        xc = to_cl(x)                   # K = cin is a whole number of 64-byte half K steps: the MFMA GEMM kernels
        b, c, p, a = xc.shape
        w2 = weight.reshape(cout, cin)
        # output widths padded with zero rows (a 1- or 4-channel head), sliced off again: the data-gradient GEMM contracts
        # over cout and takes the MFMA kernels only for whole 64-byte half K steps (32 bf16 / 16 fp32 values) -- padded to 8
        # (what the weight-gradient kernels need) the two output convolutions of the rotation head ran their data
        # gradients on the any-shape kernel, 0.07 ms each
        pad = (-cout) % (32 if x.dtype == torch.bfloat16 else 16)
        if pad:
            w2 = torch.cat((w2, w2.new_zeros(pad, cin)), 0)
        if col_stats and not pad and bias is None:
            y2d, part = gemm.matmul_nt(xc.permute(0, 2, 3, 1).reshape(-1, c), w2, True, x_amax)
        else:
            y2d = gemm.matmul_nt(xc.permute(0, 2, 3, 1).reshape(-1, c), w2, False, x_amax)
        if pad:
            y2d = y2d[:, :cout]
        y = y2d.reshape(b, p, a, cout).permute(0, 3, 1, 2)
    elif x.is_cuda and cin % 16 == 0 and cout % 16 == 0:
        y = IntraSO3ConvFn.apply(cast_feats(x, torch.float32), weight.reshape(cout, cin),
                                 _identity_index(x.shape[3], x.device))
        y = cast_feats(y, x.dtype)
    elif x.is_cuda and cin == 1 and cout % 4 == 0 and 256 % (cout // 4) == 0 and cout <= 1024:
        # single input channel (the occupancy feature of the first block): an outer product, written channels-last
        b, _, p, a = x.shape
        y2d = Conv1x1C1Fn.apply(to_cl(x).float().reshape(-1), weight.reshape(cout).float())
        y = y2d.reshape(b, p, a, cout).permute(0, 3, 1, 2)
    elif x.is_cuda and cin == 1:
        y = (to_cl(x).permute(0, 2, 3, 1).float() * weight.reshape(cout)).permute(0, 3, 1, 2)
    else:
        y = torch.nn.functional.conv2d(x.float(), weight.reshape(cout, cin, 1, 1))
    y = y if bias is None else y + bias.view(1, -1, 1, 1)
    return (y, part) if col_stats else y


def linear(x, weight, bias=None):
    """nn.Linear / a 1x1 convolution on rows: x [rows, cin] @ weight[cout, cin]^T (+ bias) on the library's NT / TN GEMMs
    (fp32, cin % 16 == 0; output widths padded to whole 8-channel groups as in conv1x1).  The per-anchor tails of the
    heads (attention logits, class logits) go through this instead of MIOpen / BLAS."""
    cout, cin = weight.shape[0], weight.shape[1]
    w2 = weight.reshape(cout, cin)
    if x.is_cuda and x.dtype == torch.float32 and cin % 16 == 0:
        pad = (-cout) % 8
        if pad:
            w2 = torch.cat((w2, w2.new_zeros(pad, cin)), 0)
        y = gemm.matmul_nt(x.contiguous(), w2)
        if pad:
            y = y[:, :cout]
    else:
        y = torch.nn.functional.linear(x.float(), w2)
    return y if bias is None else y + bias


class PointnetSO3ConvGemmFn(torch.autograd.Function):
    """PointnetSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:219-235) composed with the library's GEMMs (csrc/pointnet.hip,
    "GEMM-composed form"): Z = F W[:, :c]^T on the matrix pipe of the features' dtype (fp32 result), then one streaming pass
    adds the three coordinate channels and the bias and takes the max / arg-max over points.  Backward: the gradient routed
    through the arg-max point as a dense dZ, dF = dZ W and dW[:, :c] = dZ^T F as GEMMs, coordinate columns and bias by a
    small fixed-order reduction.  Same semantics as PointnetSO3ConvFn (first maximum wins)."""

    @staticmethod
    def forward(ctx, feats, xyz, anchors, weight, bias):
        lib = _lib.get_lib()
        fc = to_cl(feats, "feats")
        b, c, p, a = fc.shape
        co = weight.shape[0]
        if weight.numel() != co * (c + 3):
            raise ValueError(f"embed weight must be [co, c+3, 1, 1] with c={c}, got {tuple(weight.shape)}")
        if tuple(xyz.shape) != (b, 3, p):
            raise ValueError(f"xyz must be [b,3,p]=({b},3,{p}), got {tuple(xyz.shape)}")
        w = weight.reshape(co, c + 3).contiguous().float()
        xyz = xyz.contiguous()
        anc = anchors.contiguous() if (anchors is not None and a > 1) else None
        F2 = fc.permute(0, 2, 3, 1).reshape(b * p * a, c)            # view of the channels-last buffer
        wc = gemm.cast(w[:, :c].contiguous(), fc.dtype)              # [co, c] in the features' dtype
        Z = _launch("pointnet_fwd", ("pointnet", b, p, a, c, co), 2.0 * b * p * a * co * (c + 3), fc.device,
                    lambda: gemm.gemm_nt(F2, wc, out_dtype=torch.float32))
        out = torch.empty((b, a, co), dtype=torch.float32, device=fc.device)
        arg = torch.empty((b, a, co), dtype=torch.int32, device=fc.device)
        ctr = torch.empty((b, 3), dtype=torch.float32, device=fc.device)
        bs = bias.contiguous().float() if bias is not None else None
        _lib.check(lib.epn_pointnet_max_f32(_lib.dev_ptr(Z, "Z"), _lib.dev_ptr(xyz, "xyz"), _lib.dev_ptr(anc, "anchors"),
                                            _lib.dev_ptr(w, "weight"), _lib.dev_ptr(bs, "bias"), _lib.dev_ptr(out, "out"),
                                            _lib.dev_ptr(arg, "argmax", torch.int32), _lib.dev_ptr(ctr, "centre"),
                                            b, p, a, c, co, _lib.stream_of(fc)), "pointnet_max")
        ctx.save_for_backward(fc, xyz, anc, w, wc, arg, ctr)
        ctx.cfg = (b, p, a, c, co, tuple(weight.shape), bias is not None, weight.dtype)
        return out.permute(0, 2, 1)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.get_lib()
        fc, xyz, anc, w, wc, arg, ctr = ctx.saved_tensors
        b, p, a, c, co, wshape, has_bias, wdtype = ctx.cfg
        g = grad_out.permute(0, 2, 1).contiguous().float()          # [b][a][co]
        st = _lib.stream_of(fc)
        need_f = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[3] or (has_bias and ctx.needs_input_grad[4])
        dF = dW = db = None
        dZ = torch.empty((b * p * a, co), dtype=fc.dtype, device=fc.device)
        dz = lib.epn_pointnet_dz_bf16 if fc.dtype == torch.bfloat16 else lib.epn_pointnet_dz_f32
        _lib.check(dz(_lib.dev_ptr(g, "grad_out"), _lib.dev_ptr(arg, "argmax", torch.int32),
                      ctypes.c_void_p(dZ.data_ptr()), b, p, a, co, st), "pointnet_dz")
        F2 = fc.permute(0, 2, 3, 1).reshape(b * p * a, c)
        if need_f:
            dF = empty_cl(b, c, p, a, fc.device, fc.dtype)
            dF2 = dF.permute(0, 2, 3, 1).reshape(b * p * a, c)
            wt = gemm.transpose_cast(w[:, :c].contiguous(), fc.dtype)   # [c, co]: dF = dZ W as an NT GEMM
            _launch("pointnet_bwd_data", ("pointnet", b, p, a, c, co), 2.0 * b * p * a * co * c, fc.device,
                    lambda: gemm.gemm_nt(dZ, wt, out=dF2))
        if need_w:
            dW = torch.empty((co, c + 3), dtype=torch.float32, device=fc.device)
            db = torch.empty(co, dtype=torch.float32, device=fc.device) if has_bias else None
            _launch("pointnet_bwd_weight", ("pointnet", b, p, a, c, co), 2.0 * b * p * a * co * c, fc.device,
                    lambda: gemm.gemm_tn(dZ, F2, out=dW[:, :c]))
            _lib.check(lib.epn_pointnet_bwd_coord_f32(_lib.dev_ptr(g, "grad_out"), _lib.dev_ptr(arg, "argmax", torch.int32),
                                                      _lib.dev_ptr(xyz, "xyz"), _lib.dev_ptr(anc, "anchors"),
                                                      _lib.dev_ptr(ctr, "centre"), _lib.dev_ptr(dW, "grad_W"),
                                                      _lib.dev_ptr(db, "grad_bias"), b, p, a, c, co, st),
                       "pointnet_bwd_coord")
            dW = dW.view(wshape).to(wdtype)
        return dF, None, None, dW, db


def pointnet_so3conv(feats, xyz, anchors, weight, bias):
    """The GEMM-composed form for feature widths the GEMM kernels take (c % 16 == 0, co % 8 == 0) and at least 16 k feature
    rows, in the features' own dtype (classification / rotation heads, 122 880 rows: rotation network 1794-1809 -> 1850
    clouds/s in an A/B on one box; the 3DMatch head has 4096 rows and its six extra launches cost more than they save);
    otherwise, and with EPN_POINTNET=fused, the fused fp32 kernels (bf16 features converted once).  EPN_POINTNET=gemm forces
    the composed form wherever the widths allow."""
    c, co = feats.shape[1], weight.shape[0]
    form = ab("EPN_POINTNET")
    rows = feats.shape[0] * feats.shape[2] * feats.shape[3]
    if (feats.is_cuda and c % 16 == 0 and co % 8 == 0 and feats.dtype in FEATURE_DTYPES and form != "fused"
            and (form == "gemm" or rows >= 16384)):
        return PointnetSO3ConvGemmFn.apply(feats, xyz, anchors, weight, bias)
    return PointnetSO3ConvFn.apply(cast_feats(feats, torch.float32), xyz, anchors, weight, bias)
