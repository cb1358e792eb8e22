"""Noam learning-rate schedule around a fused Adam (reference: utils/optimizer.py:3-32, utils/functions.py:107).

NoamOpt keeps the reference's attributes (`optimizer`, `_step`, `_rate`, `warmup`, `factor`, `model_size`, `min_lr`) and
methods (`step`, `zero_grad`, `rate`).  FusedAdam is a torch.optim.Adam whose step() is ONE asr_adam_step launch over the
flat fp32 parameter / gradient / moment buffers, so `optimizer.state_dict()` keeps torch's Adam schema (checkpoints
interchange with the reference).
"""
import torch

from asr_hip import ops
from asr_hip import params as P


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, ddp_bucket_bytes=None, ddp_wire="fp32"):
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self.flat = None            # asr_hip.params.FlatParams, created once the parameters live on a GPU
        self.reducer = None         # asr_hip.ddp.GradReducer (only under --parallel with world_size > 1)
        self.ddp_bucket_bytes = ddp_bucket_bytes
        self.ddp_wire = ddp_wire
        self.grad_scale = None      # device scalar set by clip_grad_norm_()
        self._t = 0
        self._ensure_flat()

    def _ensure_flat(self):
        """Flatten lazily: the reference builds the optimiser BEFORE model.cuda() (train.py:101-110)."""
        if self.flat is not None:
            self.flat.rebind()
            return
        plist = [p for g in self.param_groups for p in g['params']]
        if not plist or not plist[0].is_cuda:
            return
        old = {p: dict(self.state[p]) for p in plist if p in self.state and 'exp_avg' in self.state[p]}
        self.flat = P.FlatParams(plist, order=self._slot_order(plist))
        self._m = torch.zeros_like(self.flat.data)
        self._v = torch.zeros_like(self.flat.data)
        for p, o in zip(self.flat.params, self.flat.offsets):
            if p in old:
                n = p.numel()
                self._m[o:o + n].copy_(old[p]['exp_avg'].reshape(-1))
                self._v[o:o + n].copy_(old[p]['exp_avg_sq'].reshape(-1))
                self._t = max(self._t, int(float(old[p]['step'])))
        self._bind_state()
        if self.ddp_bucket_bytes is not None:
            import torch.distributed as dist
            import os
            if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("ASR_FORCE_DDP") == "1"):
                from asr_hip.ddp import GradReducer
                self.reducer = GradReducer(self.flat, bucket_bytes=self.ddp_bucket_bytes, wire=self.ddp_wire)
                self.reducer.broadcast_parameters(0)
        P.set_reducer(self.reducer)

    @staticmethod
    def _slot_order(plist):
        """Registration order, except that inside every attention block the three projection weights (and then their
        three biases) are made adjacent, so Q/K/V (or K/V) run as ONE GEMM over a contiguous slice of the flat buffers.
        nn.Module registration order inside MultiHeadAttention is q.w q.b k.w k.b v.w v.b ... (models/common_layers.py)."""
        out, i = [], 0
        while i < len(plist):
            grp = plist[i:i + 6]
            if (len(grp) == 6 and all(g.dim() == 2 for g in grp[0::2]) and all(g.dim() == 1 for g in grp[1::2]) and
                    grp[0].shape == grp[2].shape == grp[4].shape and grp[1].shape == grp[3].shape == grp[5].shape and
                    grp[0].shape[0] == grp[1].shape[0] and getattr(grp[0], "_asr_qkv", False)):
                out += [grp[0], grp[2], grp[4], grp[1], grp[3], grp[5]]
                i += 6
            else:
                out.append(plist[i])
                i += 1
        # the decoder's cross-attention K / V projections of ALL layers read the same encoder output: their weights (k0 v0 k1 v1 ...) and
        # then their biases are made adjacent across the layers, so the K | V of every layer is ONE GEMM (round 6: models/asr/transformer.py
        # Decoder tags them with `_asr_cross_kv = (kind, index)`; the per-layer K | V pair stays adjacent inside the group)
        cross = [q for q in out if getattr(q, "_asr_cross_kv", None) is not None]
        if cross:
            ids = [id(q) for q in out]
            first = min(ids.index(id(q)) for q in cross)
            rest = [q for q in out if getattr(q, "_asr_cross_kv", None) is None]
            grp = sorted([q for q in cross if q._asr_cross_kv[0] == "w"], key=lambda q: q._asr_cross_kv[1]) + \
                  sorted([q for q in cross if q._asr_cross_kv[0] == "b"], key=lambda q: q._asr_cross_kv[1])
            first = min(first, len(rest))
            out = rest[:first] + grp + rest[first:]
        return out

    def _bind_state(self):
        for p, o in zip(self.flat.params, self.flat.offsets):
            n = p.numel()
            st = self.state[p]
            st['step'] = torch.tensor(float(self._t))
            st['exp_avg'] = self._m[o:o + n].view(p.shape)
            st['exp_avg_sq'] = self._v[o:o + n].view(p.shape)

    def zero_grad(self, set_to_none=False):
        """Gradients are accumulation targets of the kernels: zero them, never drop them."""
        self._ensure_flat()
        # (a host without a device: is_current_stream_capturing() raises hipErrorNoDevice there, and nothing can be capturing)
        if self.flat is None or not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
            # nothing of a previous (failed) backward pass may trail into the gradients zeroed here; a completed pass left the queues empty.
            # (Inside a capture the queues belong to the graph body being recorded: graph.py flushes them itself.)
            ops.reset_pending()
        if self.reducer is not None:
            self.reducer.begin_step()
        if self.flat is not None:
            self.flat.zero_grad()
            return
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    p.grad.zero_()

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ (reference: trainer.py:108-109) without a host sync: the coefficient stays on
        the device and is folded into the next step()."""
        ops.join_deferred()                     # weight gradients still running on the second stream
        if self.reducer is not None:
            self.reducer.finish()
        dev = self.param_groups[0]['params'][0].device
        acc = torch.zeros(1, device=dev, dtype=torch.float32)
        if self.flat is not None:
            ops.sumsq_acc(self.flat.grad, acc)
        else:
            for g in self.param_groups:
                for p in g['params']:
                    if p.grad is not None:
                        ops.sumsq_acc(p.grad.reshape(-1), acc)
        self.grad_scale = torch.empty(1, device=dev, dtype=torch.float32)
        # data parallel: the buffer holds gradients of the un-normalised loss SUM; 1 / global token count is folded in here
        ops.grad_coef(acc, max_norm, self._denominator(), self.grad_scale)
        return acc

    def _denominator(self):
        """Device scalar the gradients still have to be divided by (the all-reduced non-PAD token count under data
        parallelism, asr_hip/ddp.py), or None."""
        if self.reducer is not None and self.reducer.active and self.flat is not None:
            return self.flat.stats[1:2]
        return None

    def _pre_step(self):
        """Finish the gradient exchange; make sure grad_scale carries 1 / global count when it is owed."""
        ops.join_deferred()                     # weight gradients still running on the second stream
        if self.reducer is not None:
            self.reducer.finish()
        den = self._denominator()
        if den is not None and self.grad_scale is None:
            self.grad_scale = torch.empty(1, device=den.device, dtype=torch.float32)
            ops.grad_coef(None, 0.0, den, self.grad_scale)

    def global_loss(self):
        """Data parallel: the reference's loss over the gathered batch = all-reduced loss sum / all-reduced token count
        (valid after clip_grad_norm_() / step() of the current step, i.e. once the stats slot has been reduced).  Host float."""
        st = self.flat.stats[:2].tolist()
        return st[0] / max(st[1], 1.0)

    @torch.no_grad()
    def step(self, closure=None):
        self._ensure_flat()
        self._pre_step()
        self._t += 1
        for group in self.param_groups:
            b1, b2 = group['betas']
            if self.flat is not None:
                ops.adam_step(self.flat.data, self.flat.grad, self._m, self._v, group['lr'], b1, b2, group['eps'], self._t,
                              self.grad_scale)
                for p in self.flat.params:
                    self.state[p]['step'] += 1
                break
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p.data)
                    st['exp_avg_sq'] = torch.zeros_like(p.data)
                st['step'] += 1
                ops.adam_step(p.data.view(-1), p.grad.view(-1), st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1),
                              group['lr'], b1, b2, group['eps'], int(st['step'].item()), self.grad_scale)
        self.grad_scale = None
        P.bump_generation()          # masters were rewritten through the C ABI: weight shadows are stale

    @torch.no_grad()
    def step_device(self, factor_ms, warmup, min_lr, lr_out=None, guard=None):
        """Graph-replayable step: step count, Noam lr and bias corrections are computed ON THE DEVICE from
        ops.step_state()[1] (advanced by ops.step_advance() at the top of the step).  Needs flat storage."""
        self._ensure_flat()
        if self.flat is None:
            raise RuntimeError("step_device needs the parameters on a GPU")
        self._pre_step()
        b1, b2 = self.param_groups[0]['betas']
        # bf16 compute: the step itself writes the flat bf16 shadow the next step's GEMMs read (no cast pass over the masters)
        shadow = self.flat.shadow_for_step(torch.bfloat16) if ops.compute_dtype() == torch.bfloat16 else None
        ops.adam_noam_step(self.flat.data, self.flat.grad, self._m, self._v, b1, b2, self.param_groups[0]['eps'], factor_ms,
                           warmup, min_lr, self.grad_scale, lr_out, guard, shadow)
        self._shadow_by_step = shadow is not None
        self.grad_scale = None

    def after_replay(self, n=1):
        """Host-side bookkeeping for `n` device-side steps (state_dict 'step' fields, shadow invalidation)."""
        self._t += n
        for p in self.flat.params:
            self.state[p]['step'] += n
        P.bump_generation()
        if getattr(self, "_shadow_by_step", False):
            self.flat.mark_shadow_fresh(torch.bfloat16)       # written by the step that this call accounts for

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self.flat is None:
            self._ensure_flat()
            return
        t = 0
        for p, o in zip(self.flat.params, self.flat.offsets):
            st = self.state.get(p, {})
            if 'exp_avg' in st:
                n = p.numel()
                self._m[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self._v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                t = max(t, int(float(st['step'])))
        self._t = t
        self._bind_state()


class NoamOpt:
    "Optim wrapper that implements rate (reference: utils/optimizer.py:3-32)."

    def __init__(self, model_size, factor, warmup, optimizer, min_lr=1e-5):
        self.optimizer = optimizer
        self._step = 0
        self.warmup = warmup
        self.factor = factor
        self.model_size = model_size
        self._rate = 0
        self.min_lr = min_lr

    def step(self):
        self._step += 1
        rate = self.rate()
        for g in self.optimizer.param_groups:
            g['lr'] = rate
        self._rate = rate
        self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()

    def rate(self, step=None):
        s = self._step if step is None else step
        return max(self.min_lr, self.factor * (self.model_size ** (-0.5) * min(s ** (-0.5), s * self.warmup ** (-1.5))))


class AnnealingOpt:
    "lr / lr_anneal per step() call (reference: utils/optimizer.py:34-45; never selected by train.py:103)."

    def __init__(self, lr, lr_anneal, optimizer):
        self.optimizer, self.lr, self.lr_anneal = optimizer, lr, lr_anneal

    def step(self):
        for g in self.optimizer.param_groups:
            g['lr'] = g['lr'] / self.lr_anneal
