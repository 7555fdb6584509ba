"""Learning-rate factor of the reference's training loop (a boundary helper beside the alpha schedule).

``UpDownScheduler(pretrain_iter, resume_iter, scheduler_flag)`` is what train.py:95-96 hands to ``torch.optim.lr_scheduler.LambdaLR``
(/root/reference/model/utils/lr_scheduler.py:31-42): the multiplier is 10 while the iteration counted from the end of SR pretraining
lies strictly inside (70000, 95000) and the flag is set, otherwise 1.  ``LambdaLR`` calls it with ITS step counter, which restarts
at 0 on resume, hence the ``resume_iter`` offset.

``WarmupMultiStepLR(cfg, optimizer, milestones, gamma, warmup_factor, warmup_iters, last_epoch)`` (same file, :14-29; not used by
train.py, kept for callers that import it): despite its name and base class the reference's ``get_lr`` never consults the milestones --
the rate ramps linearly from ``warmup_factor * cfg.SOLVER.LR`` to ``cfg.SOLVER.LR`` over ``warmup_iters`` steps and stays there
(this quirk is reproduced).  It also returns ONE value whatever the number of parameter groups; torch >= 2.6 zips strictly and
raises on that with more than one group, so this class hands the same rate to every group -- identical for the single-group
optimisers the reference builds."""
from torch.optim.lr_scheduler import MultiStepLR

BOOST_WINDOW = (70000, 95000)      # open interval, in iterations after SR pretraining
BOOST_FACTOR = 10


class UpDownScheduler:
    def __init__(self, pretrain_iter, resume_iter, scheduler_flag):
        self.pretrain_iter, self.resume_iter, self.scheduler_flag = pretrain_iter, resume_iter, scheduler_flag

    def main_iter(self, step):
        """iterations since the joint phase began, for LambdaLR step ``step``"""
        return step + self.resume_iter - (self.pretrain_iter - 1)

    def __call__(self, step):
        boosted = bool(self.scheduler_flag) and BOOST_WINDOW[0] < self.main_iter(step) < BOOST_WINDOW[1]
        return BOOST_FACTOR if boosted else 1


class WarmupMultiStepLR(MultiStepLR):
    def __init__(self, cfg, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500, last_epoch=-1):
        self.base_rate = float(cfg.SOLVER.LR)
        self.warmup_factor, self.warmup_iters = warmup_factor, warmup_iters
        super().__init__(optimizer, milestones, gamma, last_epoch)

    def ramp(self, step):
        """multiplier of cfg.SOLVER.LR at scheduler step ``step``"""
        if step >= self.warmup_iters:
            return 1.0
        t = step / self.warmup_iters
        return self.warmup_factor + (1.0 - self.warmup_factor) * t

    def get_lr(self):
        return [self.base_rate * self.ramp(self.last_epoch)] * len(self.optimizer.param_groups)
