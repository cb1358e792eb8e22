"""Whole-step hipGraph: step_advance, zero_grad, forward, label-smoothed CE, backward, (clip), Noam/Adam captured ONCE and
replayed per batch.  A training step of the 4-layer model is ~450 small launches; eagerly they cost ~13 ms of host time
per step on top of ~30 ms of GPU time, replayed they cost none (MI355X guide: "capture launch-bound inner loops in
hipGraphs").  Everything that changes between steps lives in device memory: the batch (static buffers), the dropout
seed counter and the optimiser step (ops.step_state()).

Shapes are static per graph: one GraphedTrainStep per (B, T_src, L_tgt) bucket.
"""
import torch

from . import ops
from . import params as P


class GraphedTrainStep:
    def __init__(self, model, opt, smoothing, src, src_len, tgt, clip_max_norm=None, warmup_steps=2):
        from utils.metrics import calculate_metrics
        self._metrics = calculate_metrics
        self.model, self.opt, self.smoothing, self.clip = model, opt, float(smoothing), clip_max_norm
        dev = src.device
        self.src = src.clone()
        self.tgt = tgt.clone()
        self.src_len = torch.as_tensor(src_len).to(device=dev, dtype=torch.int32).clone()
        self.lr_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        adam = opt.optimizer
        adam._ensure_flat()
        # the device-side step counter continues from the optimiser's host-side count
        st = ops.step_state(dev)
        st[1] = int(opt._step)
        self.factor_ms = float(opt.factor) * float(opt.model_size) ** -0.5
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup_steps)):       # eager warm-up: allocates workspaces, shadows, big-LDS attributes
                self._body()
                self._host_after()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.sums = self._body()
        self._host_after()                               # capture does not execute; replay below does
        self.graph.replay()

    def _body(self):
        ops.step_advance()
        self.opt.zero_grad()
        pred, gold, _, _ = self.model(self.src, self.src_len, self.tgt)
        loss, sums = self._metrics(pred, gold, smoothing=self.smoothing, loss_type="ce", sync=False)
        loss.backward()
        adam = self.opt.optimizer
        if self.clip is not None:
            adam.clip_grad_norm_(self.clip)
        adam.step_device(self.factor_ms, float(self.opt.warmup), float(self.opt.min_lr), self.lr_dev)
        return loss.detach(), sums

    def _host_after(self):
        self.opt._step += 1
        self.opt._rate = self.opt.rate()
        self.opt.optimizer.after_replay(1)

    def __call__(self, src=None, src_len=None, tgt=None):
        """Copy the batch into the static buffers (skip arguments that are already there) and replay.
        Returns (loss, sums) device tensors: sums = [loss_sum, non-PAD count, num_correct]."""
        if src is not None and src.data_ptr() != self.src.data_ptr():
            self.src.copy_(src, non_blocking=True)
        if tgt is not None and tgt.data_ptr() != self.tgt.data_ptr():
            self.tgt.copy_(tgt, non_blocking=True)
        if src_len is not None:
            sl = torch.as_tensor(src_len)
            if not (sl.is_cuda and sl.data_ptr() == self.src_len.data_ptr()):
                self.src_len.copy_(sl.to(torch.int32), non_blocking=True)
        self.graph.replay()
        self._host_after()
        return self.loss, self.sums
