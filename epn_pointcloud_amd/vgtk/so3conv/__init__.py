"""Mirror of vgtk/vgtk/so3conv (vgtk/vgtk/so3conv/__init__.py:1-3)."""
from ..spconv import SphericalPointCloud  # noqa: F401
from .functional import *  # noqa: F401,F403
from .modules import *  # noqa: F401,F403
from . import functional  # noqa: F401
