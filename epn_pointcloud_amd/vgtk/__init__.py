"""Mirror of the reference package `vgtk` (vgtk/vgtk/__init__.py:1-13), hot-path subset:
functional, point3d, pc, spconv, so3conv, utils and the extension namespace `cuda`.
Out of scope: app (trainer/logger), loss, transform, mesh, voxel."""
from . import cuda  # noqa: F401
from . import functional  # noqa: F401
from . import point3d  # noqa: F401
from . import pc  # noqa: F401
from . import spconv  # noqa: F401
from . import so3conv  # noqa: F401
from .utils import batch_gather, batch_zip, LearningRateScheduler  # noqa: F401
