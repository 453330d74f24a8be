"""Adam for the MI355X path: torch.optim.Adam's update rule (CaSE/Run.py:27: ``optim.Adam(lr=2.5e-4)``, default betas / eps, no
weight decay, no amsgrad) as ONE multi-tensor launch that also applies what the reference's loop does around it
(common/CumulativeTrainer.py:70-76): the global-norm clip (``clip_grad_norm_(params, 1)``, second launch for the norm), the EMA
update (common/EMA.py:13-18) and the refresh of the bf16 operand copies the kernels read.

``CumulativeTrainer`` recognises this class and hands it the clip threshold and its EMA object; with any other optimizer it
runs the reference's three separate calls.  State (exp_avg, exp_avg_sq, step) lives in ``self.state`` like torch's, so
``state_dict`` / ``load_state_dict`` and the resumable checkpoint work unchanged."""
import ctypes as C
import math

import torch

from . import _abi as A
from . import ops


class _Entry(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("shadow", C.c_void_p),
                ("p_bf16", C.c_void_p), ("numel", C.c_int64), ("step_size", C.c_float), ("bc2_sqrt", C.c_float)]


if C.sizeof(_Entry) != A.lib.case_sizeof_opt_tensor():
    raise ImportError("case_rg_amd.optim: CaseOptTensor is %d bytes in libcase_hip.so, %d here" % (A.lib.case_sizeof_opt_tensor(), C.sizeof(_Entry)))


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, low_precision=None):
        """``low_precision``: dtype (torch.bfloat16) of operand copies to refresh in the same pass, or None."""
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.low_precision = low_precision
        self._chunks = {}
        # the entry table carries the step-dependent Adam scalars, so its bytes change every step: it goes up through a ring of PINNED
        # staging buffers with a non-blocking copy into ONE persistent device table (no pageable upload, no new device tensor per step)
        self._table = None   # device uint8
        self._stage = []     # [(pinned uint8, event of its last upload)]
        self._stage_next = 0
        self._low = {}  # id(parameter) -> its operand copy, rewritten in place by every step
        self._norm_ws = None  # f32 [1 + chunks]: squared gradient norm + its per-chunk partials
        # hipGraph capture of the step (common/CumulativeTrainer.py, ``capture=True``): the table upload of a captured step is a memcpy node
        # out of its OWN pinned staging buffer (re-run by every replay, so eager steps in between may reuse ``_table``), and whatever a
        # captured kernel reads that was allocated outside the capture is kept alive here for the life of the optimizer
        self._graph_stage = None
        self._graph_keep = []
        self.generation = 0  # bumped when the state tensors are replaced wholesale (load_state_dict): captured steps compare it
        self.last_stepped = []

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.generation += 1

    def next_step_number(self):
        """Step count the NEXT ``step()`` will give the parameters that have taken every step so far (1-based)."""
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        return 1 + (max(steps) if steps else 0)

    def stage_step(self, state):
        """Write the next step's Adam scalars (lr of group 0, betas, step number) into ``state`` (a stepstate.StepState) -- the caller
        uploads it before the step's kernels.  One set of hyper-parameters only: the device struct holds one step size."""
        g0 = self.param_groups[0]
        if any(g["lr"] != g0["lr"] or g["betas"] != g0["betas"] for g in self.param_groups):
            raise RuntimeError("FusedAdam: the device-resident step state carries ONE learning rate / beta pair; the groups differ")
        state.stage_adam(g0["lr"], g0["betas"][0], g0["betas"][1], self.next_step_number())

    def prepare_capture(self):
        """Before ``torch.cuda.graph`` records a step: a pinned staging buffer for the captured table upload (pinned memory cannot be
        allocated while a stream is capturing) and a parameter-copy cache that holds nothing but the optimizer's own persistent operand
        copies (an eagerly allocated cast that a captured kernel reads could be freed -- and its address reused -- later)."""
        n = max(4096, 0 if self._table is None else self._table.numel())
        self._graph_stage = torch.empty(n, dtype=torch.uint8).pin_memory()
        self._graph_keep.append(self._graph_stage)
        ops.invalidate_param_cache()
        self.reseed_param_cache()

    def reseed_param_cache(self):
        for group in self.param_groups:
            for p in group["params"]:
                lp = self._low.get(id(p))
                if lp is not None:
                    A.call("case_cast", p.data_ptr(), lp.data_ptr(), p.numel(), A.F32, ops._DT[lp.dtype], ops._stream())
                    ops.seed_param_cache(p, lp)

    def advance_host_steps(self, params):
        """A captured step was replayed: move the host-side step counts of the parameters it updates (state_dict / checkpoints)."""
        for p in params:
            self.state[p]["step"] = int(self.state[p]["step"]) + 1

    def _chunk_list(self, numels, device):
        key = (tuple(numels), str(device))
        if key not in self._chunks:
            per = A.lib.case_optim_chunk_elems()
            pairs = [(i, c) for i, n in enumerate(numels) for c in range((n + per - 1) // per)]
            self._chunks = {key: torch.tensor(pairs, dtype=torch.int32).to(device)}
        return self._chunks[key]

    @torch.no_grad()
    def step(self, closure=None, clip_norm=None, ema=None, state=None):
        """One optimizer step.  ``clip_norm``: global L2 clip threshold applied on the fly (gradients are left untouched);
        ``ema``: a ``case_rg_amd.common.EMA.EMA`` whose shadow weights are updated in the same pass; ``state``: a
        ``stepstate.StepState`` staged by ``stage_step`` and uploaded by the caller -- the kernel then reads the step size and the
        bias correction from the device struct instead of the table (required under stream capture: the table of a captured step is static)."""
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and (state is None or self._graph_stage is None):
            raise RuntimeError("FusedAdam.step() under stream capture needs prepare_capture() and a device-resident step state")
        self.last_stepped = []
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        shadow_of = {}
        if ema is not None:
            names = {id(p): n for n, p in ema.model.named_parameters()}
            shadow_of = {pid: ema.shadow[n] for pid, n in names.items() if n in ema.shadow}
        for group in self.param_groups:
            # parameters with a gradient take the Adam update; those without one (unused this step: the reference builds DDP with
            # find_unused_parameters=True, and Masque alternates 'ps_train' / 'train') only have their EMA shadow moved, as
            # EMA.update() does for every trainable parameter (common/EMA.py:13-18)
            params = [p for p in group["params"] if p.grad is not None]
            idle = [p for p in group["params"] if p.grad is None and p.requires_grad and id(p) in shadow_of]
            if not params and not idle:
                continue  # (a group without any gradient still moves its EMA shadows: the idle entries below)
            dev = (params or idle)[0].device
            if not (params or idle)[0].is_cuda:
                raise RuntimeError("case_rg_amd.optim.FusedAdam runs on the GPU only; there is no CPU path")
            beta1, beta2 = group["betas"]
            lr = float(group["lr"])
            entries = (_Entry * (len(params) + len(idle)))()
            fresh = {}
            dev_state = state
            for i, p in enumerate(params):
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FusedAdam expects contiguous float32 parameters and gradients")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = st["step"] = int(st["step"]) + 1  # one step count per parameter, as torch.optim.Adam keeps
                if state is not None and (step != state.staged_step or lr != state.staged_lr):
                    # a parameter that skipped earlier steps (Masque alternates 'ps_train' / 'train') has its own bias corrections: the
                    # table entries carry them, the one-size device struct cannot
                    if capturing:
                        raise RuntimeError("FusedAdam: a captured step needs every updated parameter at the same step count (got %d, state %d)"
                                           % (step, state.staged_step))
                    dev_state = None
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                sh = shadow_of.get(id(p))
                lp = None
                if self.low_precision is not None and p.dim() > 1:
                    lp = self._low.get(id(p))
                    if lp is None or lp.shape != p.shape or lp.device != dev:
                        lp = self._low[id(p)] = torch.empty(p.shape, dtype=self.low_precision, device=dev)
                    fresh[id(p)] = lp
                entries[i] = _Entry(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                    0 if sh is None else sh.data_ptr(), 0 if lp is None else lp.data_ptr(), p.numel(),
                                    lr / (1.0 - beta1 ** step), math.sqrt(1.0 - beta2 ** step))
                st["_g"] = g  # keep a non-contiguous gradient's copy alive until the launch has consumed it
            for j, p in enumerate(idle):
                entries[len(params) + j] = _Entry(p.data_ptr(), 0, 0, 0, shadow_of[id(p)].data_ptr(), 0, p.numel(), 0.0, 1.0)
            raw = bytes(entries)
            if capturing:
                if self._table is None or self._table.numel() < len(raw) or self._table.device != dev or len(raw) > self._graph_stage.numel():
                    raise RuntimeError("FusedAdam: run the step eagerly once before capturing it (the entry table is sized by the first step)")
                # a memcpy node out of this capture's own pinned buffer: every replay re-uploads the captured table first
                self._graph_stage[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
                self._table[:len(raw)].copy_(self._graph_stage[:len(raw)], non_blocking=True)
            else:
                if self._table is None or self._table.numel() < len(raw) or self._table.device != dev:
                    self._table = torch.empty(max(len(raw), 4096), dtype=torch.uint8, device=dev)
                    self._stage = [(torch.empty(self._table.numel(), dtype=torch.uint8).pin_memory(), None) for _ in range(3)]
                stage, ev = self._stage[self._stage_next]
                if ev is not None:
                    ev.synchronize()  # its upload of three steps ago (long done: the host never runs three steps ahead of the device)
                stage[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
                self._table[:len(raw)].copy_(stage[:len(raw)], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._stage[self._stage_next] = (stage, ev)
                self._stage_next = (self._stage_next + 1) % len(self._stage)
            table = self._table
            chunks = self._chunk_list([p.numel() for p in params + idle], dev)
            stream = torch.cuda.current_stream().cuda_stream
            sumsq = None
            if clip_norm is not None:
                # [0] the squared global norm, [1:] one partial per chunk, summed in a fixed order by one workgroup: the clip
                # coefficient is bit-identical on every data-parallel rank (and run to run)
                if self._norm_ws is None or self._norm_ws.numel() != chunks.shape[0] + 1 or self._norm_ws.device != dev:
                    self._norm_ws = torch.empty(chunks.shape[0] + 1, dtype=torch.float32, device=dev)
                sumsq = self._norm_ws
                A.call("case_optim_sumsq", table.data_ptr(), chunks.data_ptr(), chunks.shape[0], sumsq.data_ptr() + 4, sumsq.data_ptr(),
                       stream)
            A.call("case_optim_adam_ema", table.data_ptr(), chunks.data_ptr(), chunks.shape[0], None if sumsq is None else sumsq.data_ptr(),
                   float(clip_norm or 0.0), beta1, beta2, group["eps"], 0.0 if ema is None else 1.0 - ema.decay,
                   None if dev_state is None else dev_state.address, stream)
            if capturing:  # (allocated outside the capture, read by its kernels on every replay)
                self._graph_keep += [table, chunks, sumsq]
            self.last_stepped += params
            for p in params:
                self.state[p].pop("_g", None)
            # the kernel wrote the parameters behind autograd's back (_version did not move): drop every cached operand copy and
            # seed the cache with the copies this pass produced
            ops.invalidate_param_cache()
            for p in params:
                if id(p) in fresh:
                    ops.seed_param_cache(p, fresh[id(p)])
        return loss
