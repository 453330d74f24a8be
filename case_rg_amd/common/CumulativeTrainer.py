"""Training / prediction loop (reference: common/CumulativeTrainer.py:13-156), same public API.

What is kept: constructor signature, ``train_batch`` / ``train_epoch`` / ``predict`` / ``serialize``, gradient
accumulation, clip-norm 1, optimizer -> EMA -> scheduler -> zero_grad order, DistributedSampler sharding, per-rank
prediction lists.  What differs (none of it changes the numbers):
  * data parallelism is ``case_rg_amd.parallel.GradSync`` (bucketed RCCL all-reduce overlapped with backward) instead of
    a DistributedDataParallel wrapper, so ``self.model`` stays the bare module and ``serialize`` also works on one GPU / CPU
    (the reference crashes on ``self.model.module`` there);
  * the per-loss ``.cpu().item()`` (3 device->host syncs per step, :57) is one stacked copy;
  * batches are uploaded one ahead on a side stream from pinned memory (``utils.pipeline.DevicePrefetcher``)
    instead of blocking ``.cuda()`` calls inside the step;
  * ``save_checkpoint`` / ``load_checkpoint`` write what a resume needs next to the reference's weights-only
    ``<epoch>.pkl`` (:80-86): optimizer moments + step, scheduler, EMA shadow, the accumulation counter and the RNG streams.
"""
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.utils.data.distributed import DistributedSampler

from .. import config, ops
from ..optim import FusedAdam
from ..parallel import GradSync
from ..utils.pipeline import DevicePrefetcher
from .EMA import EMA


def init_params(model, escape=None):
    """xavier-uniform on every parameter with dim > 1, embedding row 0 included (reference :13-24)."""
    for name, param in model.named_parameters():
        if escape is not None and escape in name:
            continue
        if param.data.dim() > 1:
            torch.nn.init.xavier_uniform_(param.data)
    if hasattr(model, 'reset_parameters'):
        model.reset_parameters()
    ops.invalidate_param_cache()  # writes through .data leave _version alone


class CumulativeTrainer(object):
    def __init__(self, model, tokenizer, detokenizer, local_rank, num_gpus, accumulation_steps=1, ema_rate=0.995, capture=None):
        """``capture`` (not in the reference; True / False / "auto"; default: environment CASE_STEP_GRAPH = 1 / 0 / auto): replay the training step from a hipGraph
        recorded after two eager steps per batch shape (``case_rg_amd.stepgraph``) -- for the geometries whose step is launch-bound
        (the reference's default: hidden 256, batch 16).  The per-step scalars then live in device memory (``self.step_state``)."""
        self.local_rank = local_rank
        self.num_gpus = num_gpus
        self.tokenizer = tokenizer
        self.detokenizer = detokenizer
        if local_rank is not None and torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        self.model = model.cuda() if torch.cuda.is_available() else model
        self.accumulation_steps = accumulation_steps
        self.accumulation_count = 0
        # CASE_FORCE_GRADSYNC: keep the bucket / hook / collective machinery live on a one-rank group (one-GPU rehearsal of the DP path)
        self.sync = (GradSync(self.model, force=bool(os.environ.get("CASE_FORCE_GRADSYNC")))
                     if dist.is_available() and dist.is_initialized() else None)
        self.ema = EMA(self.model, ema_rate)
        self.ema.register()
        self._loss_host = None
        if capture is None:
            env = os.environ.get("CASE_STEP_GRAPH", "0")
            capture = "auto" if env == "auto" else env == "1"
        self.step_state, self.graphs = None, None
        if capture and torch.cuda.is_available():
            from ..stepgraph import StepGraphs
            from ..stepstate import StepState
            self.step_state = StepState(next(self.model.parameters()).device)
            config.set_device_state(self.step_state.address)  # every dropout site from here on adds the device-resident base
            self.graphs = StepGraphs(self, auto=capture == "auto")  # "auto": a capture is kept only where its replay is faster than the eager step

    def close(self):
        """Detach the device-resident step state from the process-wide dropout configuration (a capturing trainer owns it)."""
        if self.step_state is not None and config.device_state() == self.step_state.address:
            config.set_device_state(None)
        self.step_state, self.graphs = None, None

    def train_batch(self, epoch, data, method, optimizer, scheduler=None):
        try:
            return self._train_batch(epoch, data, method, optimizer, scheduler)
        except Exception:
            # the step died between backward and finish(): release what GradSync holds (reserved CUs, bucket counters).  Only for ordinary
            # exceptions -- a KeyboardInterrupt / SystemExit must not wait on collectives whose peers may be gone -- and a failure inside
            # abort() is reported but never replaces the error that ended the step.  Recovery after a ONE-rank failure is not supported:
            # the other ranks are blocked in their collectives until the process group's timeout.
            if self.sync is not None:
                try:
                    self.sync.abort()
                except Exception as cleanup:  # noqa: BLE001
                    print("CumulativeTrainer: GradSync.abort() failed while handling the step's error: %r" % (cleanup,), file=sys.stderr)
            raise

    def _train_batch(self, epoch, data, method, optimizer, scheduler=None):
        if self.graphs is not None:
            losses = self.graphs.run(epoch, data, method, optimizer, scheduler)
            if losses is not None:
                self.accumulation_count += 1
                return losses
            t0 = time.perf_counter()
            losses = self._eager_batch(epoch, data, method, optimizer, scheduler)
            self.graphs.note_eager_ms((time.perf_counter() - t0) * 1e3)
            return losses
        return self._eager_batch(epoch, data, method, optimizer, scheduler)

    def _eager_batch(self, epoch, data, method, optimizer, scheduler=None):
        self.accumulation_count += 1
        boundary = self.accumulation_count % self.accumulation_steps == 0
        if self.sync is not None:
            self.sync.no_sync(not boundary)
        state = self.step_state if isinstance(optimizer, FusedAdam) else None
        if self.step_state is not None:
            # device-resident step scalars: the dropout sites of this step are numbered from 0 and add the base uploaded here -- the
            # same numbering a replayed capture of the step uses
            base = config.begin_step()
            if state is not None and boundary:
                optimizer.stage_step(state)
            self.step_state.upload(base)
        loss = self.model(data, method=method)
        if isinstance(loss, (tuple, list)):
            parts = torch.cat([l.mean().reshape(1) for l in loss])
        else:
            parts = loss.mean().reshape(1)
        (parts.sum() / self.accumulation_steps).backward()
        # the losses go to a pinned buffer without blocking; the host waits for them only AFTER the all-reduce wait and the
        # optimizer kernels are enqueued (round 3 drained the device here, in the middle of the step, before enqueueing them)
        host, done = None, None
        if parts.is_cuda:
            if self._loss_host is None or self._loss_host.numel() < parts.numel():
                self._loss_host = torch.empty(max(8, parts.numel()), dtype=torch.float32).pin_memory()
            host = self._loss_host[:parts.numel()]
            host.copy_(parts.detach().float(), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        if boundary:
            if self.sync is not None:
                self.sync.finish()
            if isinstance(optimizer, FusedAdam):  # clip + Adam + EMA + bf16 operand refresh in one multi-tensor pass (K15)
                optimizer.step(clip_norm=1.0, ema=self.ema, state=state)
            else:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1)
                optimizer.step()
                ops.invalidate_param_cache()  # optimizers that write through p.data leave _version alone
                self.ema.update()
            if scheduler is not None:
                scheduler.step()
            optimizer.zero_grad()
        if host is None:
            return parts.detach().cpu().tolist()
        done.synchronize()
        return host.tolist()

    def serialize(self, epoch, output_path):
        if self.local_rank not in (0, None):
            return
        output_path = os.path.join(output_path, 'model/')
        os.makedirs(output_path, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(output_path, '.'.join([str(epoch), 'pkl'])))

    def save_checkpoint(self, epoch, output_path, optimizer, scheduler=None):
        """Resumable state of the loop after ``epoch``: ``<output_path>/model/<epoch>.ckpt`` (rank 0 only, like serialize)."""
        if self.local_rank not in (0, None):
            return None
        output_path = os.path.join(output_path, 'model/')
        os.makedirs(output_path, exist_ok=True)
        state = {
            'epoch': epoch,
            'model': self.model.state_dict(),
            'optimizer': optimizer.state_dict(),
            'scheduler': None if scheduler is None else scheduler.state_dict(),
            'ema_shadow': dict(self.ema.shadow),
            'ema_decay': self.ema.decay,
            'accumulation_count': self.accumulation_count,
            'rng': {'torch': torch.get_rng_state(), 'dropout_counter': config.rng_state(),
                    'cuda': torch.cuda.get_rng_state() if torch.cuda.is_available() else None},
        }
        path = os.path.join(output_path, '.'.join([str(epoch), 'ckpt']))
        torch.save(state, path)
        return path

    def load_checkpoint(self, path, optimizer, scheduler=None):
        """Restore everything ``save_checkpoint`` wrote (every rank loads the same file); returns the epoch it was taken after."""
        device = next(self.model.parameters()).device
        state = torch.load(path, map_location=device, weights_only=False)
        self.model.load_state_dict(state['model'], strict=True)
        optimizer.load_state_dict(state['optimizer'])
        if scheduler is not None and state['scheduler'] is not None:
            scheduler.load_state_dict(state['scheduler'])
        named = dict(self.model.named_parameters())
        if set(state['ema_shadow']) != set(self.ema.shadow):
            raise RuntimeError('checkpoint EMA shadow does not match the model\'s trainable parameters')
        self.ema.shadow = {n: t.to(named[n].device, copy=True) for n, t in state['ema_shadow'].items()}
        self.ema.decay = state['ema_decay']
        self.accumulation_count = state['accumulation_count']
        torch.set_rng_state(state['rng']['torch'].cpu())
        if state['rng']['cuda'] is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(state['rng']['cuda'].cpu())
        config.set_rng_state(state['rng']['dropout_counter'])
        if self.graphs is not None:
            self.graphs.reset()  # the captures read tensors this load replaced (EMA shadows, optimizer moments)
        ops.invalidate_param_cache()  # load_state_dict copies in place under no_grad: cached bf16 operand copies are stale
        return state['epoch']

    def _loader(self, dataset, collate_fn, batch_size, shuffle, epoch=None):
        if dist.is_available() and dist.is_initialized():
            sampler = DistributedSampler(dataset, shuffle=shuffle)
            if epoch is not None:
                sampler.set_epoch(epoch)
            return torch.utils.data.DataLoader(dataset, collate_fn=collate_fn, batch_size=batch_size, sampler=sampler, pin_memory=True)
        return torch.utils.data.DataLoader(dataset, collate_fn=collate_fn, batch_size=batch_size, shuffle=shuffle,
                                           pin_memory=torch.cuda.is_available())

    def train_epoch(self, method, train_dataset, train_collate_fn, batch_size, epoch, optimizer, scheduler=None):
        self.model.train()
        loader = self._loader(train_dataset, train_collate_fn, batch_size, True, epoch)
        start, count, bloss = time.time(), 0, 0

        def report():
            msg = ['Method', method, 'Epoch', epoch, 'Batch ', count, 'Loss ', bloss, 'Time ', time.time() - start]
            if scheduler is not None:
                msg += ['Learning rate ', scheduler.get_last_lr()]
            print(*msg)
            sys.stdout.flush()

        for j, data in enumerate(DevicePrefetcher(loader), 0):  # batch j+1 is uploaded on a side stream while j computes (f3)
            count += 1
            bloss = self.train_batch(epoch, data, method=method, optimizer=optimizer, scheduler=scheduler)
            if j > 0 and j % 100 == 0:
                report()
        if self.accumulation_count % self.accumulation_steps != 0:  # flush a partial group (no clip / EMA, reference :122-126)
            if self.sync is not None:
                self.sync.no_sync(False)
                self.sync.finish()
            optimizer.step()
            ops.invalidate_param_cache()
            if scheduler is not None:
                scheduler.step()
            optimizer.zero_grad()
        report()

    def predict(self, method, dataset, collate_fn, batch_size):
        self.model.eval()
        rs = []
        with torch.no_grad():
            for data in DevicePrefetcher(self._loader(dataset, collate_fn, batch_size, False)):
                rs.append([data, self.model(data, method=method)])
        return rs
