"""Mirror of vgtk/vgtk/spconv (vgtk/vgtk/spconv/__init__.py:1-3), hot-path subset: SphericalPointCloud and
the grouping helpers the SO(3) path shares.  The ZPConv modules (BasicZPConv/IntraZPConv/InterZPConv/
AnchorProp, spconv/modules.py:54-149) are used by no shipped model and are out of scope."""
from .base import SphericalPointCloud  # noqa: F401
from .functional import *  # noqa: F401,F403
from . import functional  # noqa: F401
