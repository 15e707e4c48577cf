"""`vgtk.cuda.zpconv` -- importable because vgtk.spconv.functional / vgtk.so3conv.functional import it
at module load (vgtk/vgtk/spconv/functional.py:14); the four functions are dead code in the reference
(their only callers are commented out) and are not part of the hot path."""


def _dead(name):
    def f(*args, **kwargs):
        raise NotImplementedError(f"vgtk.cuda.zpconv.{name}: legacy ZPConv kernel, unreachable in the reference's "
                                  "shipped models; the SO(3) path uses the fused kernels in epn_pointcloud_amd.ops")
    f.__name__ = name
    return f


inter_zpconv_forward = _dead("inter_zpconv_forward")
inter_zpconv_backward = _dead("inter_zpconv_backward")
intra_zpconv_forward = _dead("intra_zpconv_forward")
intra_zpconv_backward = _dead("intra_zpconv_backward")
