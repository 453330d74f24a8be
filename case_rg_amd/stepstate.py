"""Device-resident step state (ABI 600, include/case_hip.h ``CaseStepState``) -- the scalars of a training step that change from step
to step (reference: common/CumulativeTrainer.py:52-78; CaSE/Run.py:27-28): where the dropout counter stream stands, the learning rate
the scheduler set, Adam's bias corrections.  As kernel ARGUMENTS they would be frozen into a captured hipGraph (a replayed step would
redraw the same masks and repeat one Adam step size); as 64 bytes of device memory that every dropout site and the optimizer kernel
read when they RUN, one captured step serves every step.

The host stays the owner of the values: before a step it writes the struct into the next slot of a small ring of pinned copies and
enqueues ONE 64-byte host -> device copy on the compute stream, ahead of the step's kernels (eager launches or a graph replay) --
exact for any LR scheduler, no kernel, and no race: the copy is stream-ordered and a slot is rewritten only after its copy has run.
``case_step_advance`` (one single-thread launch) does the same on the device for replay loops that never come back to the host.
"""
import ctypes as C
import math

import torch

from . import _abi as A
from . import config


class StepState(object):
    RING = 32

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("case_rg_amd.stepstate.StepState lives in device memory; there is no CPU path")
        self.dev = torch.zeros(C.sizeof(A.StepState), dtype=torch.uint8, device=self.device)
        self.ring = torch.zeros(self.RING, C.sizeof(A.StepState), dtype=torch.uint8).pin_memory()
        self._events = [None] * self.RING
        self._slot = 0
        self.host = A.StepState()  # what the device holds once the last upload has run
        self.staged_lr, self.staged_step = None, 0

    @property
    def address(self):
        return self.dev.data_ptr()

    def stage_adam(self, lr, beta1, beta2, step):
        """The optimizer's scalars for step number ``step`` (1-based), formed in double and rounded to f32 once -- the same expressions
        the per-tensor table entries carry (optim.py), so both routes give the same bits."""
        self.host.lr, self.host.step = float(lr), int(step)
        self.staged_lr, self.staged_step = float(lr), int(step)  # (host.lr is the f32 rounding: comparisons use these)
        self.host.step_size = float(lr) / (1.0 - beta1 ** step)
        self.host.bc2_sqrt = math.sqrt(1.0 - beta2 ** step)

    def upload(self, rng_base=None):
        """Enqueue the struct (with ``rng_base``, default: config's stream base) on the current stream.  Never inside a capture."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("StepState.upload() inside a stream capture: upload before the capture / replay, the graph only READS the state")
        base = config.rng_state()[1] if rng_base is None else int(rng_base)
        if base & 1:
            raise ValueError("the dropout counter base must be even (the kernels hash element pairs)")
        self.host.rng_base = base
        i = self._slot
        self._slot = (i + 1) % self.RING
        if self._events[i] is not None:
            self._events[i].synchronize()  # the copy out of this slot RING uploads ago (long done)
        C.memmove(self.ring[i].data_ptr(), C.addressof(self.host), C.sizeof(A.StepState))
        self.dev.copy_(self.ring[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[i] = ev

    def advance_on_device(self, rng_stride, beta1, beta2):
        """case_step_advance: step += 1, rng_base += rng_stride, Adam scalars from the state's lr -- capturable (one tiny kernel)."""
        from . import ops
        A.call("case_step_advance", self.address, int(rng_stride), float(beta1), float(beta2), ops._stream())

    def read(self):
        """The device's copy (synchronises): tests / debugging."""
        out = A.StepState()
        raw = self.dev.cpu().numpy().tobytes()
        C.memmove(C.addressof(out), raw, C.sizeof(A.StepState))
        return out
