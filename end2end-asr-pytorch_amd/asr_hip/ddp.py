"""Data-parallel gradient reduction for one-process-per-GPU training: replaces the reference's single-process
nn.DataParallel (reference: utils/functions.py:154-160; scatter / replicate / gather / reduce_add_coalesced every step,
SURVEY.md 2c) with all-reduce(SUM) over RCCL of slices of the ONE flat fp32 gradient buffer.

Two launch modes share this class:
  * eager (trainer, variable shapes): buckets of ~32 MiB are all-reduced from inside backward as soon as the last
    gradient of a bucket has been enqueued (asr_hip.params.grad_ready), i.e. overlapped with the rest of backward;
  * graph replay (asr_hip/graph.py, fixed shapes): the collectives stay OUTSIDE the captured hipGraphs -- `hold` is set,
    mark_ready does nothing and GraphedTrainStep issues `all_reduce_range` between graph segments.

The backend is whatever torch.distributed was initialised with: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the tests.

Loss normalisation: the reference's loss is the mean over the non-PAD tokens of the GATHERED
batch.  Every rank back-propagates its un-normalised local SUM; the three CE statistics [loss_sum, token count,
num_correct] live in a 64-float `stats` slot at the end of the flat gradient buffer, so they are summed over ranks by
the same all-reduce as the gradients, and the optimiser kernel multiplies every gradient by 1 / global count
(asr_grad_coef).  No collective of its own, no host synchronisation, exact equivalence with one process on the
concatenated batch.
"""
import os

import torch
import torch.distributed as dist


class _WireWork:
    """Handle of an all-reduce that travelled in bf16: wait() = wait for the collective, then widen the staging slice back into the
    fp32 gradient slice on the current stream (the optimiser and the clipping norm read fp32)."""

    def __init__(self, work, staging, grad):
        self.work, self.staging, self.grad = work, staging, grad

    def wait(self):
        if self.work is not None:
            self.work.wait()
        from . import ops
        ops.widen_flat(self.staging, self.grad)


class _Both:
    def __init__(self, a, b):
        self.a, self.b = a, b

    def wait(self):
        self.a.wait()
        self.b.wait()


class GradReducer:
    WIRE_MIN = 1 << 16          # elements: smaller slices (the conv front end's 0.4 M parameters are above it too) always travel in fp32

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, wire="fp32"):
        """wire: "fp32" (default) or "bf16" -- the gradient slices are cast into a bf16 staging buffer (one launch), all-reduced
        there (half the bytes per xGMI link: 74 MB instead of 147 MB for configs[1]) and widened back; the sum itself is then taken in
        bf16 by the collective (relative error ~2^-9 per addition).  The stats slot (loss sum, token count, correct count: exact
        integers / the loss) always travels in fp32.  SURVEY.md 2c costed this; the default stays fp32 until it is measured on xGMI."""
        if wire not in ("fp32", "bf16"):
            raise ValueError("--grad-wire must be fp32 or bf16")
        self.wire = wire
        self._staging = None
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # ASR_FORCE_DDP=1: issue the collectives even with a single rank (exercises the whole path on one GPU)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("ASR_FORCE_DDP") == "1")
        self.hold = False          # graph-replay mode: GraphedTrainStep owns the collectives
        # Buckets are contiguous slices of the flat gradient buffer.  Backward produces gradients roughly in reverse
        # registration order, so slices are grown from the END of the buffer; the 4-D convolution weights (and their
        # biases) form buckets of their own because their gradients arrive LAST (the conv stack is the first layer) even
        # though nn.Module registers them after the encoder / decoder.  The last bucket carries the stats slot.
        n = len(flat.params)
        is_conv = [False] * n
        for i, p in enumerate(flat.params):      # conv weights and the 1-D parameters that follow them (biases, BatchNorm affine)
            is_conv[i] = p.dim() == 4 or (p.dim() == 1 and i > 0 and is_conv[i - 1])
        self.bucket_of = [0] * n
        self.buckets = []          # dicts: lo, hi, members(set of param indices)
        cap = max(1, int(bucket_bytes) // 4)
        hi = flat.total_all
        members = set()
        for i in range(n - 1, -1, -1):
            lo = flat.offsets[i]
            members.add(i)
            boundary = i == 0 or is_conv[i - 1] != is_conv[i]
            if hi - lo >= cap or boundary:
                self.buckets.append({"lo": lo, "hi": hi, "members": members})
                hi = lo
                members = set()
        for b, bk in enumerate(self.buckets):
            for i in bk["members"]:
                self.bucket_of[i] = b
        self.done = False
        self._reset()

    def _reset(self):
        self.pending = [set(b["members"]) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.done = False

    def begin_step(self):
        """Called by zero_grad(): a new backward is about to fill the gradient buffer."""
        self._reset()

    def broadcast_parameters(self, src=0):
        if self.active:
            dist.broadcast(self.flat.data, src, group=self.group)

    def mark_ready(self, p):
        if self.hold:
            return
        if self.done:              # a backward without zero_grad() in between: start a new round
            self._reset()
        i = self.flat.index.get(id(p))
        if i is None:
            return
        b = self.bucket_of[i]
        self.pending[b].discard(i)
        if not self.pending[b] and not self.launched[b]:
            self._launch(b)

    def _launch(self, b):
        self.launched[b] = True
        bk = self.buckets[b]
        w = self.all_reduce_range(bk["lo"], bk["hi"])
        if w is not None:
            self.works.append(w)

    def all_reduce_range(self, lo, hi, async_op=True):
        """SUM-all-reduce flat.grad_all[lo:hi] (gradients and, when the range reaches the end, the stats slot)."""
        if not self.active or hi <= lo:
            return None
        total = self.flat.total
        ghi = min(hi, total)                      # [lo, ghi) = gradients, [ghi, hi) = the stats slot (fp32 always)
        # (both ends of a bf16 slice on 16-byte boundaries of both buffers: parameter slots are 64-element aligned)
        if self.wire == "bf16" and ghi - lo >= self.WIRE_MIN and lo % 8 == 0:
            from . import ops
            if self._staging is None:
                self._staging = torch.empty(total, device=self.flat.grad_all.device, dtype=torch.bfloat16)
            st, g = self._staging[lo:ghi], self.flat.grad_all[lo:ghi]
            ops.cast_flat(g, st)
            w = dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            tail = None
            if hi > ghi:
                tail = dist.all_reduce(self.flat.grad_all[ghi:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            ww = _WireWork(w if async_op else None, st, g)
            if not async_op:
                ww.wait()
                return None
            return _Both(ww, tail) if tail is not None else ww
        return dist.all_reduce(self.flat.grad_all[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def finish(self):
        """Reduce whatever has not been reduced yet and make the current stream wait for every bucket.  Idempotent within a
        step: clip_grad_norm_() and step() both call it, the gradients are reduced once."""
        if self.done or self.hold:
            return
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for w in self.works:
            w.wait()
        self.works = []
        self.done = True


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
    """Split every BucketingSampler bin over the ranks: rank r takes utterances r, r+world, ... of each bin, so the GLOBAL
    batch stays --batch-size utterances and an epoch has as many optimiser steps as in the reference, whose nn.DataParallel
    scatters one batch over the GPUs (reference: utils/functions.py:154-160).  Every rank gets the same number of utterances
    per bin (the remainder of a bin that does not divide evenly is dropped; bins smaller than the world size are skipped), so
    collectives never deadlock."""
    out = []
    for b in bins:
        n = (len(b) // world) * world
        if n == 0:
            continue
        out.append([b[i] for i in range(rank, n, world)])
    return out
