"""Extension namespace `vgtk.cuda` (reference: three CUDAExtensions built by vgtk/setup.py:30-34).
Here each module is a thin allocator + argument checker over the C ABI of libepn_so3conv.so."""
from . import gathering, grouping, zpconv  # noqa: F401
