"""Mirror of vgtk/vgtk/utils.py: batch_gather (the hot-path entry), batch_zip, LearningRateScheduler."""
from .cuda import gathering as cuda_gather


def batch_gather(x, idx, dim=1):
    """vgtk/vgtk/utils.py:25-27 -- x[b,c,n], idx[b,m] -> [b,c,m]; `dim` is ignored there too."""
    return cuda_gather.gather_points_forward(x, idx.int())


def batch_zip(x, y, idx):
    raise NotImplementedError('batch zip cuda not implemented')  # vgtk/vgtk/utils.py:29-30


class LearningRateScheduler():
    """vgtk/vgtk/utils.py:33-68: every `decay_step` calls, lr = f(counter // decay_step)."""

    def __init__(self, optimizer, init_lr, lr_type, decay_step, **kwargs):
        self.counter = 0
        self.init_lr = init_lr
        self.lr = init_lr
        self.lr_type = lr_type
        self.optimizer = optimizer
        self.decay_step = decay_step
        self.schedule_func = getattr(self, f'_{lr_type}')(**kwargs)

    def step(self):
        self.counter += 1
        if self.counter % self.decay_step == 0:
            lr = self.schedule_func(self.counter // self.decay_step)
            print("[Optimizer] Adjusting learning rate %f ---> %f" % (self.lr, lr))
            for group in self.optimizer.param_groups:
                group['lr'] = lr
            self.lr = lr
        return self.lr

    def _constant(self, decay_rate):
        return lambda x: self.init_lr

    def _exp_decay(self, decay_rate):
        self.decay_rate = decay_rate
        return lambda x: self.init_lr * decay_rate ** x
