"""Data-parallel gradient reduction for one-process-per-GPU training: replaces the reference's single-process
nn.DataParallel (reference: utils/functions.py:154-160; scatter / replicate / gather / reduce_add_coalesced every step,
SURVEY.md 2c) with bucketed all-reduce(SUM) over RCCL, launched from inside backward as soon as a bucket's last
gradient has been enqueued (asr_hip.params.grad_ready), i.e. overlapped with the rest of backward.

The backend is whatever torch.distributed was initialised with: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests.  Gradients are SUMMED, not averaged: the loss of each rank is local_sum / GLOBAL token count (CEFn), which is
exactly the reference's loss over the gathered global batch (loss-normalisation note).
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat, bucket_bytes=32 << 20, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # ASR_FORCE_DDP=1: issue the collectives even with a single rank (exercises the RCCL + hipGraph path on one GPU)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("ASR_FORCE_DDP") == "1")
        # buckets are contiguous slices of the flat gradient buffer, filled from the END of the buffer backwards
        # because backward produces gradients roughly in reverse registration order (decoder first, conv stack last)
        n = len(flat.params)
        self.bucket_of = [0] * n
        self.buckets = []          # dicts: lo, hi, members(set of param indices)
        cap = max(1, int(bucket_bytes) // 4)
        hi = flat.total
        members = set()
        size = 0
        for i in range(n - 1, -1, -1):
            lo = flat.offsets[i]
            members.add(i)
            size = hi - lo
            if size >= cap or i == 0:
                self.buckets.append({"lo": lo, "hi": hi, "members": members})
                hi = lo
                members = set()
        for b, bk in enumerate(self.buckets):
            for i in bk["members"]:
                self.bucket_of[i] = b
        self._reset()

    def _reset(self):
        self.pending = [set(b["members"]) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.works = []

    def broadcast_parameters(self, src=0):
        if self.active:
            dist.broadcast(self.flat.data, src, group=self.group)

    def mark_ready(self, p):
        i = self.flat.index.get(id(p))
        if i is None:
            return
        b = self.bucket_of[i]
        self.pending[b].discard(i)
        if not self.pending[b] and not self.launched[b]:
            self._launch(b)

    def _launch(self, b):
        self.launched[b] = True
        if self.active:
            bk = self.buckets[b]
            self.works.append(dist.all_reduce(self.flat.grad[bk["lo"]:bk["hi"]], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=True))

    def finish(self):
        """Reduce whatever has not been reduced yet and make the current stream wait for every bucket."""
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for w in self.works:
            w.wait()
        self._reset()

    def all_reduce_scalar_(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class HipDataParallel(torch.nn.Module):
    """Drop-in for the reference's `nn.DataParallel(model, device_ids)` wrapper: same `.module` attribute and the same
    'module.'-prefixed state_dict keys (reference: utils/functions.py:154-160, train.py:92-99), but every rank runs the
    whole model on ITS slice of the batch; gradient exchange is done by GradReducer."""

    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = device_ids

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


def rank_shard(bins, rank, world):
    """Deterministic disjoint split of the BucketingSampler's bins over ranks; every rank gets the same number of bins
    (the tail that does not divide evenly is dropped so that collectives never deadlock)."""
    n = (len(bins) // world) * world
    return [bins[i] for i in range(rank, n, world)]
