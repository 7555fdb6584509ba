"""Learning-rate factor of the reference's training loop (a boundary helper beside the alpha schedule).

``UpDownScheduler(pretrain_iter, resume_iter, scheduler_flag)`` is what train.py:95-96 hands to ``torch.optim.lr_scheduler.LambdaLR``
(/root/reference/model/utils/lr_scheduler.py:31-42): the multiplier is 10 while the iteration counted from the end of SR pretraining
lies strictly inside (70000, 95000) and the flag is set, otherwise 1.  ``LambdaLR`` calls it with ITS step counter, which restarts
at 0 on resume, hence the ``resume_iter`` offset."""

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
