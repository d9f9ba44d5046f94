"""Data-parallel gradient reduction for Caduceus pre-training: one process per GPU, `torch.distributed` backend "nccl"
(= RCCL over xGMI on ROCm), gradients of all parameters living in a few flat fp32 buckets that are all-reduced
asynchronously as soon as their last gradient has been accumulated -- i.e. overlapped with the rest of backward.

Replaces Lightning's `DDPStrategy(find_unused_parameters=False, gradient_as_bucket_view=True)` used by the reference
(/root/reference/train.py:629-639).  Sized for this model: 7.7 M parameters = 30.9 MB of fp32 gradients (SURVEY.md
section 5), so the default is 4 buckets of ~8 MB in reverse parameter order: the first all-reduce starts after the last
layers' backward and hides behind the remaining ones; on xGMI's point-to-point links the whole exchange is < 0.5 ms.
Sequences are independent, so there is no other collective on the data path.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, bucket_bytes: int = 8 << 20,
                 average: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        # testing aid: run the collectives even in a 1-rank group (exercises the RCCL path on a single GPU)
        self._force = os.environ.get("CADUCEUS_DP_FORCE_COLLECTIVE") == "1" and dist.is_initialized()
        seen, plist = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:  # tied parameters appear once
                seen.add(id(p))
                plist.append(p)
        plist.reverse()  # gradients become ready roughly in reverse registration order
        self.params: List[torch.nn.Parameter] = plist
        self.buckets: List[torch.Tensor] = []
        self._bucket_of = {}
        self._pending: List[int] = []
        self._handles: List[Optional[object]] = []
        self.sync_enabled = True
        cur, cur_bytes = [], 0
        groups = []
        for p in plist:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            groups.append(cur)
        self._groups = groups
        self._views = {}
        for bi, grp in enumerate(groups):
            n = sum(p.numel() for p in grp)
            flat = torch.zeros(n, dtype=torch.float32, device=grp[0].device)
            off = 0
            for p in grp:
                if p.dtype != torch.float32:
                    raise TypeError("master parameters are expected in fp32")
                self._views[id(p)] = flat[off:off + p.numel()].view_as(p)  # the gradient's home inside the bucket
                p.grad = self._views[id(p)]
                off += p.numel()
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)
            self.buckets.append(flat)
        self._sizes = [len(g) for g in groups]
        self._reset_counters()
        # measurement aid (bench.py): time every finish() -- on a GPU as a pair of events on the compute stream, i.e. how long that
        # stream had to wait for all-reduces that backward did NOT hide (the exposed part); on CPU as host time
        self.time_exposed = False
        self.allreduces_launched = 0  # collectives issued so far: one per bucket and OPTIMIZER step, whatever the number of micro-steps
        self._exposed = []

    def _reset_counters(self):
        self._pending = list(self._sizes)
        self._handles = [None] * len(self.buckets)

    def _gather(self, bi: int):
        """Bring the gradients autograd left OUTSIDE the bucket (p.grad was None: autograd assigned its own tensor) into
        their views with one multi-tensor launch per bucket, instead of one accumulate kernel per parameter."""
        srcs, dsts, stale = [], [], []
        for p in self._groups[bi]:
            view = self._views[id(p)]
            g = p.grad
            if g is None:
                stale.append(view)          # parameter without gradient this step
            elif g.data_ptr() != view.data_ptr():
                srcs.append(g)
                dsts.append(view)
        # a gradient found outside the bucket means p.grad was None before this backward (zero_grad() below, or the caller's
        # zero_grad(set_to_none=True)): it REPLACES the view's content; accumulation micro-steps never get here, autograd
        # adds into the view in place
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        if stale:
            torch._foreach_zero_(stale)
        for p in self._groups[bi]:
            p.grad = self._views[id(p)]

    def _launch(self, bi: int):
        flat = self.buckets[bi]
        if self.average:
            flat.div_(self.world)
        self._handles[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.allreduces_launched += 1

    def _hook(self, p: torch.nn.Parameter):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._gather(bi)
            if self.sync_enabled and (self.world > 1 or self._force):
                self._launch(bi)

    def finish(self):
        """Call after backward: gathers the buckets whose last gradient never arrived (unused parameters), waits for the
        outstanding all-reduces (and launches any that never triggered).
        Parameters that received no gradient keep a ZERO gradient (their bucket view is zeroed for the all-reduce and stays
        attached), where DDP / Lightning leave `grad = None`: an optimizer with weight decay therefore still decays them.  All
        parameters of the Caduceus models receive gradients, so the two coincide on this path."""
        if not self.sync_enabled:
            raise RuntimeError("BucketedGradReducer.finish() inside no_sync(): call it after the last micro-step, outside the "
                               "context (the all-reduce would be skipped silently)")
        mark = self._mark() if self.time_exposed else None
        self._finish()
        if mark is not None:
            self._exposed.append((mark, self._mark()))

    def _mark(self):
        dev = self.buckets[0].device if self.buckets else torch.device("cpu")
        if dev.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(dev))
            return e
        import time
        return time.perf_counter()

    def exposed_ms(self, reset: bool = True) -> List[float]:
        """Per finish() since the last reset: milliseconds between entering and leaving it on the compute stream (GPU: the wait for
        all-reduces still in flight after backward + the launch of buckets that never triggered; synchronises)."""
        out = []
        for a, b in self._exposed:
            if isinstance(a, float):
                out.append((b - a) * 1e3)
            else:
                b.synchronize()
                out.append(a.elapsed_time(b))
        if reset:
            self._exposed = []
        return out

    def _finish(self):
        for bi in range(len(self.buckets)):
            if self._pending[bi] != 0:
                self._gather(bi)
        if self.sync_enabled and (self.world > 1 or self._force):
            for bi in range(len(self.buckets)):
                if self._handles[bi] is None:
                    self._launch(bi)
            for h in self._handles:
                h.wait()
        self._reset_counters()

    def zero_grad(self):
        """Start a new accumulation: gradients are detached from the buckets (p.grad = None), so autograd hands over its own
        tensors and the next gather OVERWRITES the bucket -- no zero-fill, no per-parameter accumulate kernels."""
        for grp in self._groups:
            for p in grp:
                p.grad = None

    class _NoSync:
        def __init__(self, r):
            self.r = r

        def __enter__(self):
            self.r.sync_enabled = False

        def __exit__(self, *a):
            # the accumulation micro-steps decremented the per-bucket counters without launching anything: start the
            # final (synchronising) micro-step from full counters, so that its hooks fire the all-reduces DURING its
            # backward (overlap) instead of leaving every bucket to finish()
            self.r.sync_enabled = True
            self.r._pending = list(self.r._sizes)

    def no_sync(self):
        """Gradient accumulation micro-steps (accumulate_grad_batches, configs/experiment/hg38/hg38.yaml:17):
        skip the collective; call `finish()` only after the last micro-step's backward (outside this context)."""
        return BucketedGradReducer._NoSync(self)
