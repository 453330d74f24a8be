"""LR schedule used by the reference's Run.py (CaSE/Run.py:28): linear warm-up, then cosine with hard restarts.

The reference takes it from ``transformers.optimization`` (pinned transformers==2.1.1, not importable offline), so
this restates the commonly documented definition.  Not verifiable against the PINNED version; round 6 pins it against the
same function of the transformers release this image carries (5.x): identical learning rates over warm-up, one and three cycles and past
the end (tests/test_host_logic.py::test_lr_schedule_equals_the_installed_transformers_implementation).
    step < warmup : step / max(1, warmup)
    else          : progress = (step - warmup) / max(1, total - warmup); 0 if progress >= 1 else
                    max(0, 0.5 * (1 + cos(pi * ((cycles * progress) mod 1))))"""
import math

from torch.optim.lr_scheduler import LambdaLR


def get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=1.0,
                                                       last_epoch=-1):
    def factor(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * progress) % 1.0))))

    return LambdaLR(optimizer, factor, last_epoch)
