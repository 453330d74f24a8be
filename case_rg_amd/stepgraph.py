"""hipGraph-captured training steps (round 6; reference loop: common/CumulativeTrainer.py:52-78, default geometry CaSE/Run.py:72-78).

At the reference's own default geometry (hidden 256, batch 16) a step is ~1250 kernels of 5-20 us: 14.4 ms of GPU work that the host
needs 20 ms to launch.  ``StepGraphs`` records ONE step -- forward, backward, the loss copy, clip + Adam + EMA + operand refresh,
zero_grad -- per (method, input shapes) into a hipGraph after a few eager steps and replays it for every later batch of that shape:

  * inputs are copied into the static tensors the capture read (stream-ordered, ahead of the replay);
  * what changes from step to step is DATA on the device, not a kernel argument (ABI 600, ``stepstate.StepState``): the base of the
    dropout counter stream, the learning rate, Adam's bias corrections -- one 64-byte upload ahead of the replay, so every replay draws
    new masks and takes the right Adam step; eager steps of a capturing trainer number their dropout sites the same way, so an eager
    step and a replayed step of the same position in the stream draw identical masks;
  * host-side bookkeeping the recording pass would have done per step (per-parameter step counts, the scheduler, the RNG position) is
    replayed on the host;
  * with a process group (world > 1) the step is captured in TWO segments around the gradient all-reduce, which stays eager:
    forward + backward | GradSync's bucketed collectives | clip + Adam + EMA.  The optimizer segment reads the gradients out of
    GradSync's persistent flat buckets, so both tables are static.

Anything the capture cannot serve -- another batch shape beyond ``MAX_GRAPHS``, gradient accumulation, an optimizer other than
FusedAdam, parameters / optimizer state / EMA shadows that were replaced since the capture -- runs the eager step.  The captured
memory (activations of one step) stays reserved in the graph's private pool.
"""
import time

import torch

from . import config, ops
from .optim import FusedAdam


def _signature(data, method):
    return (method,) + tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(data.items()) if torch.is_tensor(v))


class _Captured(object):
    """One recorded step: the graph(s), the static inputs they read and what the host has to redo per replay."""
    fwd_bwd = optim = static = grads = sync_params = stepped = None
    consumed = nloss = epoch = opt_generation = 0
    shadow = pointers = None


class StepGraphs(object):
    WARM = 2        # eager steps of a shape before it is captured (allocator, lazily built optimizer state, zero-arena estimate)
    MAX_GRAPHS = 4  # captured shapes kept at once (each holds one step's activations)

    def __init__(self, trainer, auto=False):
        self.trainer = trainer
        # ``auto``: keep a capture only if it pays -- the first replays of a shape are timed against its eager warm-up steps (wall clock;
        # train_batch ends in a host read, so both are whole-step times) and a capture that is not at least 3 % faster is dropped (its
        # private pool -- one step's activations -- goes back) and the shape stays eager.  GPU-bound geometries (BASELINE cfg 2) lose nothing
        # but the trial; launch-bound ones (the reference's default geometry) keep the graph.
        self.auto = auto
        self.eager_ms, self.replay_ms = {}, {}
        self.seen = {}
        self.graphs = {}
        self.disabled = set()
        self.replays = 0
        # The reference's collate pads the answers of a batch to its LONGEST answer (CaSE/CaSEDataset.py:135-136), so data['response'] changes
        # shape from batch to batch while every other tensor is fixed-size: padded here with PAD (0) up to the model's max_target_length, all
        # batches share one captured step.  PAD targets are ignored by the loss (ignore_index 0, CaSE/Model.py:306), are masked as keys and sit
        # behind every real position of the causal decoder: losses and gradients are unchanged (tested), the decoder just runs a few padded rows.
        self.pad_response = getattr(trainer.model, "max_target_length", None)

    def reset(self):
        """Forget every captured step (after a checkpoint load or any other wholesale replacement of tensors the captures read)."""
        self.graphs.clear()
        self.seen.clear()

    # ------------------------------------------------------------------------------------------
    def _pointers(self, optimizer):
        out = []
        for p in self.trainer.model.parameters():
            st = optimizer.state.get(p) or {}
            out.append((p.data_ptr(), st["exp_avg"].data_ptr() if "exp_avg" in st else 0, st["exp_avg_sq"].data_ptr() if "exp_avg_sq" in st else 0))
        return out

    def run(self, epoch, data, method, optimizer, scheduler):
        """One optimizer step from a captured graph; None = the caller runs the eager step."""
        tr = self.trainer
        if not isinstance(optimizer, FusedAdam) or tr.accumulation_steps != 1 or tr.step_state is None:
            return None
        if any(torch.is_tensor(v) and not v.is_cuda for v in data.values()):
            return None  # (a host tensor would be uploaded inside the step: not capturable)
        r = data.get("response")
        if self.pad_response and torch.is_tensor(r) and r.dim() == 2 and r.shape[1] < self.pad_response:
            data = dict(data)
            data["response"] = torch.nn.functional.pad(r, (0, self.pad_response - r.shape[1]))
        sig = _signature(data, method)
        if sig in self.disabled:
            return None
        g = self.graphs.get(sig)
        if g is None:
            n = self.seen[sig] = self.seen.get(sig, 0) + 1
            if n <= self.WARM or len(self.graphs) >= self.MAX_GRAPHS:
                self._sig_running_eagerly = sig  # (the trainer reports the step's wall time through note_eager_ms)
                return None
            before = config.rng_state()[1]
            optimizer.last_stepped = []
            try:
                g = self.graphs[sig] = self._capture(data, method, optimizer)
            except Exception as err:  # noqa: BLE001 -- an op the capture cannot record (a host read, an unsupported call): this shape stays eager
                import warnings
                warnings.warn("case_rg_amd.stepgraph: the training step could not be captured (%s: %s); running it eagerly" % (type(err).__name__, err))
                self.disabled.add(sig)
                for p in optimizer.last_stepped:  # host-side effects of a recording pass that executed nothing
                    optimizer.state[p]["step"] = int(optimizer.state[p]["step"]) - 1
                config.skip_rng(before - config.rng_state()[1])
                optimizer.zero_grad()
                if tr.sync is not None:
                    tr.sync.no_sync(False)
                return None
        elif g.opt_generation != optimizer.generation or g.shadow is not tr.ema.shadow:
            del self.graphs[sig]  # moments / EMA shadows were replaced (load_state_dict, load_checkpoint): warm up and record again
            self.seen[sig] = 0
            return None
        if g.epoch != ops.PARAM_EPOCH:
            # somebody rewrote parameters through .data since the last replay (EMA swap for evaluation, broadcast, eager steps of another
            # shape): the captured forward reads the optimizer's persistent bf16 copies -- refresh them; moved storages end the capture
            if g.pointers != self._pointers(optimizer):
                del self.graphs[sig]
                self.seen[sig] = 0
                return None
            optimizer.reseed_param_cache()
        for k, v in g.static.items():
            src = data[k]
            if src is not v:
                v.copy_(src, non_blocking=True)
        st = tr.step_state
        t0 = time.perf_counter() if self.auto and len(self.replay_ms.get(sig, ())) < 3 else None
        base = config.begin_step()
        optimizer.stage_step(st)
        st.upload(base)
        if g.optim is None:
            g.fwd_bwd.replay()
        else:  # world > 1: forward + backward | eager bucketed all-reduce | optimizer
            g.fwd_bwd.replay()
            for p, grad in zip(g.sync_params, g.grads):
                p.grad = grad  # (None for a parameter this method leaves without a gradient: its bucket slice is zero-filled)
            tr.sync.reduce_now()
            g.optim.replay()
            for p in g.sync_params:
                p.grad = None
        config.skip_rng(g.consumed)
        optimizer.advance_host_steps(g.stepped)
        g.epoch = ops.PARAM_EPOCH
        self.replays += 1
        if scheduler is not None:
            scheduler.step()
        done = torch.cuda.Event()
        done.record()
        done.synchronize()
        losses = tr._loss_host[:g.nloss].tolist()
        if t0 is not None:
            times = self.replay_ms.setdefault(sig, [])
            times.append((time.perf_counter() - t0) * 1e3)
            eager = self.eager_ms.get(sig)
            if len(times) == 3 and eager and sorted(times)[1] > 0.97 * min(eager):
                del self.graphs[sig]  # the replay does not pay at this geometry: back to eager steps, for good
                self.disabled.add(sig)
        return losses

    def note_eager_ms(self, ms):
        """Wall time of the eager step the trainer just ran for the signature ``run`` declined (auto mode's yardstick)."""
        sig = getattr(self, "_sig_running_eagerly", None)
        if sig is not None:
            self.eager_ms.setdefault(sig, []).append(ms)
            self._sig_running_eagerly = None

    # ------------------------------------------------------------------------------------------
    def _capture(self, data, method, optimizer):
        tr = self.trainer
        st = tr.step_state
        segmented = tr.sync is not None and tr.sync.active
        g = _Captured()
        g.static = {k: v.clone() for k, v in data.items() if torch.is_tensor(v) and v.is_cuda}
        feed = dict(data)
        feed.update(g.static)
        optimizer.zero_grad()
        optimizer.prepare_capture()
        if tr._loss_host is None:
            tr._loss_host = torch.empty(8, dtype=torch.float32).pin_memory()
        base = config.begin_step()
        optimizer.stage_step(st)  # (nothing is uploaded or executed here: the recording pass only needs the staged step number)
        torch.cuda.synchronize()
        if segmented:
            tr.sync.no_sync(True)  # the hooks stay silent inside the capture: the collectives run between the two segments
        g.fwd_bwd = torch.cuda.CUDAGraph()
        nloss = [0]

        def forward_backward():
            loss = tr.model(dict(feed), method=method)
            parts = torch.cat([l.mean().reshape(1) for l in loss]) if isinstance(loss, (tuple, list)) else loss.mean().reshape(1)
            parts.sum().backward()
            nloss[0] = parts.numel()
            tr._loss_host[:parts.numel()].copy_(parts.detach().float(), non_blocking=True)

        def optimize():
            optimizer.step(clip_norm=1.0, ema=tr.ema, state=st)
            optimizer.zero_grad()

        try:
            if not segmented:
                with torch.cuda.graph(g.fwd_bwd, capture_error_mode="thread_local"):
                    forward_backward()
                    optimize()
            else:
                with torch.cuda.graph(g.fwd_bwd, capture_error_mode="thread_local"):
                    forward_backward()
                g.sync_params = list(tr.sync.params)
                g.grads = [p.grad for p in g.sync_params]  # static: every replay writes these addresses
                tr.sync.adopt_bucket_views()  # p.grad := the view of its flat bucket, as finish() leaves it (no communication)
                g.optim = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g.optim, capture_error_mode="thread_local", pool=g.fwd_bwd.pool()):
                    optimize()
        finally:
            if segmented:
                tr.sync.no_sync(False)
        g.nloss = nloss[0]
        g.consumed = config.rng_state()[1] - base
        # the recording pass executed nothing: take back its host-side effects (the first replay is the step itself)
        g.stepped = list(optimizer.last_stepped)
        for p in g.stepped:
            optimizer.state[p]["step"] = int(optimizer.state[p]["step"]) - 1
        config.skip_rng(-g.consumed)
        g.epoch = ops.PARAM_EPOCH
        g.opt_generation = optimizer.generation
        g.shadow = tr.ema.shadow
        g.pointers = self._pointers(optimizer)
        return g
