"""Mirror of vgtk/vgtk/utils.py: batch_gather (the hot-path entry), batch_zip.  The trainer-side
LearningRateScheduler of that file is out of scope and not mirrored."""
from .cuda import gathering as cuda_gather


def batch_gather(x, idx, dim=1):
    """vgtk/vgtk/utils.py:25-27 -- x[b,c,n], idx[b,m] -> [b,c,m]; `dim` is ignored there too."""
    return cuda_gather.gather_points_forward(x, idx.int())


def batch_zip(x, y, idx):
    raise NotImplementedError('batch zip cuda not implemented')  # vgtk/vgtk/utils.py:29-30
