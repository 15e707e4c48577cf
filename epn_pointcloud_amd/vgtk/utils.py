"""Mirror of vgtk/vgtk/utils.py: batch_gather (the hot-path entry), batch_zip, and the trainer-side
LearningRateScheduler (vgtk/vgtk/utils.py:33-68; exported by the reference package, vgtk/vgtk/__init__.py:12, and used by
its trainer, vgtk/vgtk/app/trainer.py:167) -- pure Python, kept so that code written against the reference API imports."""
from .cuda import gathering as cuda_gather


def batch_gather(x, idx, dim=1):
    """vgtk/vgtk/utils.py:25-27 -- x[b,c,n], idx[b,m] -> [b,c,m]; `dim` is ignored there too."""
    return cuda_gather.gather_points_forward(x, idx.int())


def batch_zip(x, y, idx):
    raise NotImplementedError('batch zip cuda not implemented')  # vgtk/vgtk/utils.py:29-30


class LearningRateScheduler:
    """Step-count learning-rate schedule with the reference's interface: LearningRateScheduler(optimizer, init_lr, lr_type,
    decay_step, **kwargs); `step()` is called once per iteration and returns the current rate.  Every `decay_step` calls
    the rate becomes schedule(n) with n = calls // decay_step and is written into every param group.  lr_type:
    'constant' (kwargs: decay_rate, ignored) or 'exp_decay' (init_lr * decay_rate ** n)."""

    def __init__(self, optimizer, init_lr, lr_type, decay_step, **kwargs):
        self.optimizer, self.init_lr, self.lr = optimizer, init_lr, init_lr
        self.lr_type, self.decay_step, self.counter = lr_type, decay_step, 0
        maker = getattr(self, "_" + lr_type, None)
        if maker is None:
            raise AttributeError(f"unknown lr_type {lr_type!r} (constant | exp_decay)")
        self.schedule_func = maker(**kwargs)

    def step(self):
        self.counter += 1
        if self.counter % self.decay_step == 0:
            new = self.schedule_func(self.counter // self.decay_step)
            print("[Optimizer] Adjusting learning rate %f ---> %f" % (self.lr, new))
            for group in self.optimizer.param_groups:
                group["lr"] = new
            self.lr = new
        return self.lr

    def _constant(self, decay_rate=None):
        return lambda n: self.init_lr

    def _exp_decay(self, decay_rate):
        self.decay_rate = decay_rate
        return lambda n: self.init_lr * decay_rate ** n
