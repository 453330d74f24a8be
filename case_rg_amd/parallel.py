"""Single-node data parallelism: one process per GPU, gradients summed with bucketed all-reduce
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The reference wraps the model in DistributedDataParallel (common/CumulativeTrainer.py:45-47).  Here the model
stays a bare module: per-parameter post-accumulate hooks fill flat fp32 buckets in reverse registration order
(~ the order backward produces them: decoder first, encoder last) and launch one asynchronous all-reduce per
bucket as soon as it is complete, so communication of the decoder / block gradients overlaps the rest of
backward (which is > 90 % of its time).  ``finish()`` waits, averages and scatters the buckets back.
Batch items never interact in forward, so there is no other collective on the path.
"""
import os

import torch
import torch.distributed as dist


class GradSync:
    ABORT_WAIT = __import__("datetime").timedelta(seconds=30)  # abort(): longest wait per collective already in flight

    def __init__(self, model, bucket_mb=64, process_group=None, force=False, reserve_cus=None, comm_dtype=None):
        """``force`` keeps the bucket / hook / collective machinery active on a one-rank group (used by the GPU test that
        drives the RCCL path on a single device).
        ``reserve_cus``: compute units the persistent kernels (256 x 256 GEMM, K16 .. K19) leave free while the group is active, so
        that RCCL's kernels can be resident beside them and the all-reduce really overlaps backward (default: environment
        CASE_DP_RESERVE_CUS, else 8 = one per XCD; 0 switches it off).  ``comm_dtype``: torch.bfloat16 halves the bytes on the wire
        (gradients are rounded once before the sum; default f32, environment CASE_DP_BF16=1 selects bf16)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []  # [flat buffer, [(param, offset, numel)], pending count, work handle]
        self._slot = {}
        self._armed = False
        self._next = 0  # index of the next bucket to launch (in-order collectives)
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.exposed = []  # (start, end) event pairs around finish()'s waits on the compute stream: exposed_ms()
        self.comm_dtype = comm_dtype if comm_dtype is not None else (torch.bfloat16 if os.environ.get("CASE_DP_BF16") == "1" else None)
        self._reserved_now = False
        self.reserved_cus = 0
        if not self.active:
            return
        if reserve_cus is None:
            reserve_cus = int(os.environ.get("CASE_DP_RESERVE_CUS", "8"))
        self.reserved_cus = reserve_cus if self.params and self.params[0].is_cuda else 0
        self._reserved_now = False  # the reservation is held only while collectives are in flight: first bucket launch .. finish()
        # RCCL averages inside the collective (ReduceOp.AVG); gloo (the CPU tests, the shared-GPU rehearsal) sums and finish() divides
        self._avg_in_collective = dist.get_backend(process_group) == "nccl"
        cap = int(bucket_mb * (1 << 20) // 4)
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self._close_bucket(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += (p.numel() + 3) // 4 * 4  # every view starts on a 16-byte boundary (the optimizer kernel's vector path; the padding stays zero)
        if cur:
            self._close_bucket(cur, cur_n)
        self.broadcast_parameters(model)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._armed = True

    def _close_bucket(self, items, numel):
        p0 = items[0][0]
        flat = torch.zeros(numel, dtype=torch.float32, device=p0.device)
        idx = len(self.buckets)
        views = [flat[off:off + n].view_as(p) for p, off, n in items]
        self.buckets.append({"flat": flat, "items": items, "views": views, "pending": len(items), "work": None})
        for p, _, _ in items:
            self._slot[id(p)] = idx

    def broadcast_parameters(self, model):
        """Rank 0's parameters and buffers become everyone's (what DDP does at wrap time)."""
        from . import ops
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)
        ops.invalidate_param_cache()  # the broadcast writes through .data: cached bf16 operand copies are stale

    def no_sync(self, flag=True):
        """Gradient accumulation micro-steps: skip the all-reduce (the reference does not)."""
        if self.active and self._reserved_now and self._next == 0:
            self._reserve(False)  # a step that never reached finish() (backward raised) must not leave the chip short of CUs
        self._armed = not flag

    def abort(self):
        """A step was abandoned between backward and finish() (an exception, an early exit): wait for the collectives already in
        flight -- every rank launched the same ones --, forget the partial state and hand the reserved CUs back."""
        if not self.active:
            return
        try:
            for b in self.buckets:
                if b["work"] is not None:
                    # bounded: if a peer died, its half of the collective never arrives (the RCCL watchdog ends the job at the group's
                    # own timeout); an abandoned step must not hang here for that long
                    try:
                        b["work"].wait(timeout=self.ABORT_WAIT)
                    except TypeError:  # a backend whose Work.wait takes no timeout
                        b["work"].wait()
        finally:
            for b in self.buckets:
                b["work"], b["pending"] = None, len(b["items"])
                b.pop("events", None)
            self._next = 0
            self._reserve(False)

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self.buckets[self._slot[id(p)]]
        b["pending"] -= 1
        if p.is_cuda:
            # the hook runs on the stream that produced this gradient (the package puts the query-side block stacks on a second stream:
            # common/heads.run_block_pair); whoever gathers the bucket -- maybe a hook on the OTHER stream -- waits for exactly this point,
            # not for everything queued on that stream
            from . import ops
            if ops.AUX_STREAMS:
                ev = torch.cuda.Event()
                ev.record()
                b.setdefault("events", []).append(ev)
        self._launch_ready()

    def _launch_ready(self, flush=False):
        """Launch buckets strictly in index order -- bucket k only after 0..k-1 -- so every rank issues the same sequence of
        collectives whatever order its hooks fired in (a parameter that is unused on one rank delays its bucket, and the ones
        behind it, to ``finish()`` on that rank only; the sequence stays the same)."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b["pending"] > 0 and not flush:
                return
            self._launch(b)
            self._next += 1

    def _reserve(self, on):
        """The persistent kernels leave ``reserved_cus`` compute units to RCCL only between the first all-reduce of a step and the end of
        finish(): the forward pass (a third of the step, no collective in flight) runs on the whole chip."""
        if self.reserved_cus and on != self._reserved_now:
            from . import _abi
            _abi.call("case_set_reserved_cus", int(self.reserved_cus) if on else 0)
            self._reserved_now = on

    def _launch(self, b):
        """Gather the bucket's gradients into its flat buffer (one multi-tensor copy; gradients that already ARE the bucket
        views, e.g. after an accumulation micro-step, need none) and start the asynchronous all-reduce."""
        self._reserve(True)
        if b["flat"].is_cuda:  # the bucket's gradients may have been produced on another stream than the one this launch runs on
            cur = torch.cuda.current_stream()
            for ev in b.pop("events", ()):
                cur.wait_event(ev)
        src, dst = [], []
        for (p, off, n), view in zip(b["items"], b["views"]):
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                src.append(p.grad)
                dst.append(view)
        if src:
            torch._foreach_copy_(dst, src)
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        if self.comm_dtype is not None:
            if b.get("wire") is None:
                b["wire"] = torch.empty(b["flat"].numel(), dtype=self.comm_dtype, device=b["flat"].device)
            b["wire"].copy_(b["flat"])
            b["work"] = dist.all_reduce(b["wire"], op=op, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)

    def finish(self):
        """Flush the buckets whose hooks did not all fire (zeros stand in for the missing gradients), wait, average, and make
        every ``param.grad`` a VIEW of its bucket: the optimizer reads the reduced gradients in place, nothing is copied back."""
        if not self.active:
            return
        if not self._armed:
            self._reserve(False)
            return
        self._launch_ready(flush=True)
        timed = self.buckets and self.buckets[0]["flat"].is_cuda
        if timed:  # how long the compute stream stands still for the collectives = the communication backward did not hide
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in self.buckets:
            b["work"].wait()
        if timed:
            e1.record()
            self.exposed.append((e0, e1))
            del self.exposed[:-64]
        for b in self.buckets:
            if self.comm_dtype is not None:
                b["flat"].copy_(b["wire"])
            if not self._avg_in_collective:
                b["flat"].div_(self.world)
            for (p, _, _), view in zip(b["items"], b["views"]):
                p.grad = view
            b["work"], b["pending"] = None, len(b["items"])
        self._next = 0
        self._reserve(False)

    def reduce_now(self):
        """All-reduce the gradients as they stand in ``param.grad`` WITHOUT having seen backward (a replayed hipGraph fires no hooks):
        every bucket is gathered and launched in order, then ``finish()``'s wait / average / view assignment.  ``None`` gradients count
        as zeros, as in ``finish()``."""
        if not self.active:
            return
        self._armed = True
        self.finish()

    def adopt_bucket_views(self):
        """``param.grad`` := the parameter's view of its flat bucket, for every parameter, without communication -- the state
        ``finish()`` leaves behind.  The optimizer segment of a captured step is recorded against these persistent addresses."""
        for b in self.buckets:
            for (p, _, _), view in zip(b["items"], b["views"]):
                p.grad = view

    def exposed_ms(self, last=None):
        """Mean time per step the compute stream waited in finish() (synchronises on the recorded events)."""
        pairs = self.exposed[-last:] if last else self.exposed
        if not pairs:
            return 0.0
        pairs[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
