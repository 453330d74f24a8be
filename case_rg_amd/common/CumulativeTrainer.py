"""Training / prediction loop (reference: common/CumulativeTrainer.py:13-156), same public API.

What is kept: constructor signature, ``train_batch`` / ``train_epoch`` / ``predict`` / ``serialize``, gradient
accumulation, clip-norm 1, optimizer -> EMA -> scheduler -> zero_grad order, DistributedSampler sharding, per-rank
prediction lists.  What differs (none of it changes the numbers):
  * data parallelism is ``case_rg_amd.parallel.GradSync`` (bucketed RCCL all-reduce overlapped with backward) instead of
    a DistributedDataParallel wrapper, so ``self.model`` stays the bare module and ``serialize`` also works on one GPU / CPU
    (the reference crashes on ``self.model.module`` there);
  * the per-loss ``.cpu().item()`` (3 device->host syncs per step, :57) is one stacked copy.
"""
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.utils.data.distributed import DistributedSampler

from .. import ops
from ..parallel import GradSync
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


def _to_device(data):
    if not torch.cuda.is_available():
        return data
    return {k: (v.cuda(non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}


class CumulativeTrainer(object):
    def __init__(self, model, tokenizer, detokenizer, local_rank, num_gpus, accumulation_steps=1, ema_rate=0.995):
        self.local_rank = local_rank
        self.num_gpus = num_gpus
        self.tokenizer = tokenizer
        self.detokenizer = detokenizer
        if local_rank is not None and torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        self.model = model.cuda() if torch.cuda.is_available() else model
        self.accumulation_steps = accumulation_steps
        self.accumulation_count = 0
        self.sync = GradSync(self.model) if dist.is_available() and dist.is_initialized() else None
        self.ema = EMA(self.model, ema_rate)
        self.ema.register()

    def train_batch(self, epoch, data, method, optimizer, scheduler=None):
        self.accumulation_count += 1
        boundary = self.accumulation_count % self.accumulation_steps == 0
        if self.sync is not None:
            self.sync.no_sync(not boundary)
        loss = self.model(data, method=method)
        if isinstance(loss, (tuple, list)):
            parts = torch.cat([l.mean().reshape(1) for l in loss])
        else:
            parts = loss.mean().reshape(1)
        (parts.sum() / self.accumulation_steps).backward()
        closs = parts.detach().cpu().tolist()
        if boundary:
            if self.sync is not None:
                self.sync.finish()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1)
            optimizer.step()
            ops.invalidate_param_cache()  # optimizers that write through p.data leave _version alone
            self.ema.update()
            if scheduler is not None:
                scheduler.step()
            optimizer.zero_grad()
        return closs

    def serialize(self, epoch, output_path):
        if self.local_rank not in (0, None):
            return
        output_path = os.path.join(output_path, 'model/')
        os.makedirs(output_path, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(output_path, '.'.join([str(epoch), 'pkl'])))

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

        for j, data in enumerate(loader, 0):
            count += 1
            bloss = self.train_batch(epoch, _to_device(data), method=method, optimizer=optimizer, scheduler=scheduler)
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
            for data in self._loader(dataset, collate_fn, batch_size, False):
                data = _to_device(data)
                rs.append([data, self.model(data, method=method)])
        return rs
