"""Parameter-side state of the HIP path: compute-dtype weight shadows, fp32 gradient buffers, flat storage and the
hooks a data-parallel gradient reducer attaches to.

Masters stay fp32 `nn.Parameter`s with the reference's names and shapes (checkpoint compatibility).
Kernels never read them directly in bf16 mode: they read shadows in kernel-friendly layouts
  linear  W (N,K)          -> W  (N, pad8(K)) and W^T (K, pad8(N))       (forward / dgrad are both NT contractions)
  conv3x3 W (Co,Ci,3,3)    -> wk (Co,9,Ci)   and wd  (Ci,9 flipped,Co)   (forward / dgrad implicit GEMM)
refreshed lazily whenever the master changed (torch version counter, or `bump_generation()` from the fused
optimiser which rewrites masters through the C ABI).
"""
import torch

from . import ops

_state = {"generation": 0, "reducer": None, "seed_ctr": 0, "base_seed": None}


def bump_generation():
    _state["generation"] += 1


def next_seed():
    """Distinct 64-bit seed per dropout site and per step, derived from torch's global seed."""
    if _state["base_seed"] is None:
        _state["base_seed"] = int(torch.initial_seed()) & 0xFFFFFFFF
    _state["seed_ctr"] += 1
    return ((_state["base_seed"] << 32) ^ (_state["seed_ctr"] * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def _key(p, dtype):
    return (p._version, _state["generation"], dtype, p.data_ptr())


def linear_weight(p, dtype=None):
    """Compute-dtype view (N,K) of a linear weight for the NT forward / NN data-gradient kernels.  Parameters that live
    in a FlatParams buffer share ONE flat shadow refreshed by a single cast launch per optimiser step; fp32 mode reads
    the master itself.  Falls back to the per-weight padded shadow when K is not a whole number of 16-byte chunks."""
    dtype = dtype or ops.compute_dtype()
    N = p.shape[0]
    K = p.numel() // N
    if K % 8 == 0:
        if dtype == torch.float32:
            return p.data.view(N, K)
        home = p.__dict__.get("_asr_flat")
        if home is not None:
            flat, off = home
            return flat.shadow_view(p, off, dtype).view(N, K)
    return linear_shadow(p, dtype)[0]


def linear_shadow(p, dtype=None):
    """p: (N,K) or (N,K,1) fp32 master -> (W (N,Kp) , Wt (K,Np)) in compute dtype, zero-padded to multiples of 8."""
    dtype = dtype or ops.compute_dtype()
    sh = p.__dict__.get("_asr_shadow")
    key = _key(p, dtype)
    if sh is not None and sh["key"] == key:
        return sh["w"], sh["wt"]
    w2 = p.data.view(p.shape[0], -1)
    N, K = w2.shape
    if sh is None or sh["w"].dtype != dtype or sh["w"].device != p.device:
        sh = {"w": torch.zeros((N, ops._pad8(K)), device=p.device, dtype=dtype),
              "wt": torch.zeros((K, ops._pad8(N)), device=p.device, dtype=dtype)}
        p.__dict__["_asr_shadow"] = sh
    ops.cast_into(w2, sh["w"], sh["wt"])
    sh["key"] = key
    return sh["w"], sh["wt"]


def conv_shadow(p, dtype=None):
    """p: (Cout,Cin,3,3) fp32 master -> (wk (Cout,9,Cin), wd (Cin,9,Cout)) in compute dtype."""
    dtype = dtype or ops.compute_dtype()
    sh = p.__dict__.get("_asr_shadow")
    key = _key(p, dtype)
    if sh is not None and sh["key"] == key:
        return sh["wk"], sh["wd"]
    Cout, Cin = p.shape[0], p.shape[1]
    if sh is None or sh["wk"].dtype != dtype or sh["wk"].device != p.device:
        sh = {"wk": torch.empty((Cout, 9, Cin), device=p.device, dtype=dtype),
              "wd": torch.empty((Cin, 9, Cout), device=p.device, dtype=dtype)}
        p.__dict__["_asr_shadow"] = sh
    ops.conv_pack_weight(p.data, sh["wk"], sh["wd"])
    sh["key"] = key
    return sh["wk"], sh["wd"]


def tcf_perm_shadow(p, C, H2, dtype=None):
    """p: the encoder input projection's (N, C * H2) master (columns in the model's feature order c * H2 + h2, reference
    transformer.py:74-76) -> its compute-dtype shadow with the columns in CHANNEL-LAST order h2 * C + c: the B operand of
    ops.gemm_nn_poolbwd.  Refreshed like the conv packs: one launch whenever the master changed (every optimiser step)."""
    dtype = dtype or ops.compute_dtype()
    sh = p.__dict__.get("_asr_tcf_perm")
    key = _key(p, dtype) + (C, H2)
    if sh is not None and sh["key"] == key:
        return sh["w"]
    W = linear_weight(p, dtype)
    if sh is None or sh["w"].dtype != dtype or sh["w"].device != p.device or sh["w"].shape != W.shape:
        sh = {"w": torch.empty_like(W)}
        p.__dict__["_asr_tcf_perm"] = sh
    ops.permute_cols_tcf(W, sh["w"], C, H2)
    sh["key"] = key
    return sh["w"]


def conv_shadows(params, dtype=None):
    """conv_shadow() for several weights at once: the stale ones are packed by ONE launch (asr_conv_pack_weight_multi)."""
    dtype = dtype or ops.compute_dtype()
    stale = []
    for p in params:
        sh = p.__dict__.get("_asr_shadow")
        key = _key(p, dtype)
        if sh is not None and sh["key"] == key:
            continue
        Cout, Cin = p.shape[0], p.shape[1]
        if sh is None or sh["wk"].dtype != dtype or sh["wk"].device != p.device:
            sh = {"wk": torch.empty((Cout, 9, Cin), device=p.device, dtype=dtype),
                  "wd": torch.empty((Cin, 9, Cout), device=p.device, dtype=dtype)}
            p.__dict__["_asr_shadow"] = sh
        stale.append((p, sh, key))
    if stale:
        ops.conv_pack_weight_multi([(p.data, sh["wk"], sh["wd"]) for p, sh, _ in stale])
        for p, sh, key in stale:
            sh["key"] = key


def fused(params):
    """If the given parameters occupy CONSECUTIVE slots of one FlatParams buffer (e.g. the Q, K, V weights of an attention
    block, which FlatParams lays out back to back), return (master_flat_view, grad_flat_view, flat, offset) over all of
    them, else None.  Lets several projections run as one GEMM."""
    homes = [p.__dict__.get("_asr_flat") for p in params]
    if any(h is None for h in homes) or any(h[0] is not homes[0][0] for h in homes):
        return None
    flat, off = homes[0]
    cur = off
    for p, (_, o) in zip(params, homes):
        if o != cur:
            return None
        cur = o + p.numel()            # slots are contiguous only if every size is a multiple of the slot alignment
        if p.numel() % FlatParams.ALIGN != 0 and p is not params[-1]:
            return None
    n = cur - off
    return flat.data[off:off + n], flat.grad[off:off + n], flat, off


def grad_of(p):
    """fp32 gradient buffer of a parameter (kernels ACCUMULATE into it; `zero_grad` must zero, not drop, it)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p.data)
    return p.grad


def set_reducer(r):
    _state["reducer"] = r


def grad_ready(*params):
    """Called by a backward as soon as the LAST contribution to each parameter's gradient has been enqueued."""
    r = _state["reducer"]
    if r is not None:
        for p in params:
            r.mark_ready(p)


# ------------------------------------------------------------------------------------------------ flat storage
class FlatParams:
    """Re-homes every parameter of a module into ONE fp32 buffer (and its gradient into another), each parameter at a
    64-element aligned offset, so that the optimiser is a single kernel launch and DDP buckets are plain slices.
    `p.data` / `p.grad` become views; names, shapes and values are unchanged."""

    ALIGN = 64
    STATS = 64

    def __init__(self, module_or_params, order=None):
        params = []
        seen = set()
        it = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
        for p in it:
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if order is not None:           # a permutation of `params` (same objects): controls which weights are adjacent
            assert len(order) == len(params) and {id(p) for p in order} == seen
            params = list(order)
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        off = 0
        self.offsets = []
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.params = params
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        # gradient buffer + a 64-float statistics slot behind it: [CE loss sum, non-PAD token count, num_correct, ...].
        # Under data parallelism the slot is summed over ranks by the same all-reduce as the gradients (asr_hip/ddp.py).
        self.total_all = off + self.STATS
        self.grad_all = torch.zeros(self.total_all, device=dev, dtype=torch.float32)
        self.grad = self.grad_all[:off]
        self.stats = self.grad_all[off:]
        for p, o in zip(params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            if p.grad is not None:
                self.grad[o:o + n].copy_(p.grad.reshape(-1))
            p.grad = self.grad[o:o + n].view(p.shape)
            p.__dict__["_asr_flat"] = (self, o)
        self.index = {id(p): i for i, p in enumerate(params)}
        self._shadow = {}               # dtype -> (tensor, generation, {id(p): version})

    def range_of(self, p):
        i = self.index[id(p)]
        return self.offsets[i], self.offsets[i] + p.numel()

    def rebind(self):
        """Re-attach views after something replaced p.data / p.grad (e.g. .to(), load_state_dict on another device)."""
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            if p.data.data_ptr() != self.data[o:o + n].data_ptr():
                self.data[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.data[o:o + n].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != self.grad[o:o + n].data_ptr():
                p.grad = self.grad[o:o + n].view(p.shape)

    def shadow_view(self, p, off, dtype):
        """Slice of the flat compute-dtype shadow holding parameter p; the whole shadow is re-cast (one launch) whenever
        the optimiser stepped (generation) or p was modified through torch (version counter)."""
        ent = self._shadow.get(dtype)
        if ent is None:
            ent = [torch.empty(self.total, device=self.data.device, dtype=dtype), -1, {}]
            self._shadow[dtype] = ent
        if ent[1] != _state["generation"] or ent[2].get(id(p)) != p._version:
            ops.cast_flat(self.data, ent[0])
            ent[1] = _state["generation"]
            ent[2] = {id(q): q._version for q in self.params}
        return ent[0][off:off + p.numel()]

    def shadow_for_step(self, dtype):
        """The flat compute-dtype shadow as a destination for the optimiser kernel (asr_adam_noam_step writes the rounded new
        parameters into it); mark_shadow_fresh() afterwards."""
        ent = self._shadow.get(dtype)
        if ent is None:
            ent = [torch.empty(self.total, device=self.data.device, dtype=dtype), -1, {}]
            self._shadow[dtype] = ent
        return ent[0]

    def mark_shadow_fresh(self, dtype):
        ent = self._shadow.get(dtype)
        if ent is not None:
            ent[1] = _state["generation"]
            ent[2] = {id(q): q._version for q in self.params}

    def shadow_is_stale(self, dtype):
        """True when a master was modified since the shadow was written -- through torch (version counters) or through the C ABI
        (an eager FusedAdam.step() between two replays: generation counter) -- which a replayed graph cannot notice."""
        ent = self._shadow.get(dtype)
        return ent is not None and (ent[1] != _state["generation"] or any(ent[2].get(id(q)) != q._version for q in self.params))

    def zero_grad(self):
        self.grad_all.zero_()
