"""Input pipeline of the training / prediction loop (reference: common/CumulativeTrainer.py:93,101-108 builds a
DataLoader and moves each batch with blocking ``.cuda()`` calls inside the step).

``DevicePrefetcher`` keeps the upload off the step's critical path: batch k+1 is collated (by the DataLoader, into pinned
memory), copied host -> device with non-blocking copies on a side stream and handed over through an event while batch k is
being computed.  Without a GPU it is a pass-through, so the CPU trainer tests run the same loop."""
import torch


class DevicePrefetcher(object):
    def __init__(self, loader, device=None):
        self.loader = loader
        self.device = device
        self.enabled = torch.cuda.is_available()
        if self.enabled and device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch, stream):
        with torch.cuda.stream(stream):
            out = {}
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    if not v.is_cuda and not v.is_pinned():
                        v = v.pin_memory()  # a DataLoader built with pin_memory=True has done this already
                    v = v.to(self.device, non_blocking=True)
                out[k] = v
            done = torch.cuda.Event()
            done.record(stream)
        return out, done

    def __iter__(self):
        if not self.enabled:
            for batch in self.loader:
                yield batch
            return
        side = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        ahead = None
        try:
            ahead = self._upload(next(it), side)
        except StopIteration:
            return
        while ahead is not None:
            batch, done = ahead
            try:
                ahead = self._upload(next(it), side)  # enqueue the next upload before this batch is consumed
            except StopIteration:
                ahead = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            for v in batch.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)  # allocated on the side stream, used on the compute stream
            yield batch
