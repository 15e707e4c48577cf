"""epn_pointcloud_amd -- MI355X-native hot path of EPN's SE(3) separable point convolution.

Holds only what the path needs:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/epn_so3conv.h)
  _lib.py    ctypes binding on torch device tensors (no CPU / eager fallback)
  vgtk/      host-side mirror of the reference's `vgtk` operator / nn.Module API for this path
  ops.py     autograd Functions over the C ABI (channels-last feature tensors)

`install_vgtk_alias()` registers the mirror under the reference's import names (`vgtk`,
`vgtk.spconv`, `vgtk.so3conv`, `vgtk.cuda.grouping`, ...) so SPConvNets-style code imports unchanged.
"""
import sys

__version__ = "0.1.0"


def install_vgtk_alias():
    """Make `import vgtk`, `import vgtk.spconv as zptk`, `import vgtk.so3conv as sptk` resolve here."""
    from . import vgtk as _v
    prefix = _v.__name__
    for name, mod in list(sys.modules.items()):
        if name == prefix or name.startswith(prefix + "."):
            sys.modules["vgtk" + name[len(prefix):]] = mod
    return _v
