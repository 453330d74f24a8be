"""Exponential moving average of the trainable parameters (reference: common/EMA.py:1-32).

shadow <- decay * shadow + (1 - decay) * param after every optimizer step; one multi-tensor lerp over all
parameters instead of a Python loop with a clone per tensor."""
import torch

from .. import ops


class EMA():
    def __init__(self, model, decay):
        self.model = model
        self.decay = decay
        self.shadow = {}
        self.backup = {}

    def _trainable(self):
        return [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    def register(self):
        self.shadow = {n: p.data.clone() for n, p in self._trainable()}

    def update(self):
        named = self._trainable()
        assert all(n in self.shadow for n, _ in named)
        with torch.no_grad():
            torch._foreach_lerp_([self.shadow[n] for n, _ in named], [p.data for _, p in named], 1.0 - self.decay)

    def apply_shadow(self):
        for n, p in self._trainable():
            assert n in self.shadow
            self.backup[n] = p.data
            p.data = self.shadow[n]
        ops.invalidate_param_cache()  # p.data swaps do not bump _version: cached bf16 operand copies would go stale

    def restore(self):
        for n, p in self._trainable():
            assert n in self.backup
            p.data = self.backup[n]
        self.backup = {}
        ops.invalidate_param_cache()
