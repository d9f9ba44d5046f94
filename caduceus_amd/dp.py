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
        for bi, grp in enumerate(groups):
            n = sum(p.numel() for p in grp)
            flat = torch.zeros(n, dtype=torch.float32, device=grp[0].device)
            off = 0
            for p in grp:
                if p.dtype != torch.float32:
                    raise TypeError("master parameters are expected in fp32")
                p.grad = flat[off:off + p.numel()].view_as(p)  # gradient lives inside the bucket
                off += p.numel()
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)
            self.buckets.append(flat)
        self._sizes = [len(g) for g in groups]
        # (bucket, element offset) of every parameter, for re-attaching a gradient autograd replaced
        self._slot = {}
        for bi, grp in enumerate(groups):
            off = 0
            for p in grp:
                self._slot[id(p)] = (bi, off)
                off += p.numel()
        self._reset_counters()

    def _reset_counters(self):
        self._pending = list(self._sizes)
        self._handles = [None] * len(self.buckets)

    def _hook(self, p: torch.nn.Parameter):
        bi, off = self._slot[id(p)]
        flat = self.buckets[bi]
        if p.grad.data_ptr() < flat.data_ptr() or p.grad.data_ptr() >= flat.data_ptr() + flat.numel() * 4:
            # autograd replaced .grad (e.g. after zero_grad(set_to_none=True)): copy back into the bucket view
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.grad)
            p.grad = view
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and self.sync_enabled and (self.world > 1 or self._force):
            if self.average:
                flat.div_(self.world)
            self._handles[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Call after backward: waits for the outstanding all-reduces (and launches any that never triggered)."""
        if self.sync_enabled and (self.world > 1 or self._force):
            for bi, flat in enumerate(self.buckets):
                if self._handles[bi] is None:
                    if self.average:
                        flat.div_(self.world)
                    self._handles[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for h in self._handles:
                h.wait()
        self._reset_counters()

    def zero_grad(self):
        for flat in self.buckets:
            flat.zero_()

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
