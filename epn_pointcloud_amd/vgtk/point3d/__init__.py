"""Mirror of vgtk/vgtk/point3d: only PointSet (used by SphericalPointCloud, vgtk/vgtk/spconv/base.py:2,6)."""
import torch


class PointSet():
    """vgtk/vgtk/point3d/base.py:15-82 -- thin wrapper over p [(b,) 3|4, n]."""

    def __init__(self, p):
        self._p = p

    @property
    def is_hom(self):
        return self._p.shape[-2] == 4

    @property
    def n_batch(self):
        return self._p.shape[0]

    @property
    def n_point(self):
        return self._p.shape[-1]

    @property
    def device(self):
        return self._p.device

    @property
    def data(self):
        return self._p

    def to_hom(self):
        if self.is_hom:
            return PointSet(self._p)
        ones = torch.ones(self.n_batch, 1, self.n_point).to(self.device)
        return PointSet(torch.cat((self._p, ones), dim=-2))

    def from_hom(self):
        return PointSet(self._p if not self.is_hom else self._p[..., :3, :])
