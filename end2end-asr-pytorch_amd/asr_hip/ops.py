"""Tensor-level wrappers over the C ABI (one function per entry point, no arithmetic in Python).

Everything here launches HIP kernels from libasr_hip.so on torch's current stream.  torch is used for device
memory (torch.empty / zeros) and nothing else.
"""
import math
import os

import torch

from . import lib as L

_cfg = {"dtype": torch.bfloat16, "state": None}


def step_state(device=None):
    """Device uint64[4] {dropout seed counter, optimiser step, "the last optimiser launch skipped its update", unused}: mixed into
    every dropout seed and read by the graph-replayable optimiser kernel; advanced once per training step by step_advance()."""
    st = _cfg["state"]
    if st is None or (device is not None and st.device != torch.device(device)):
        st = torch.zeros(4, dtype=torch.int64, device=device or "cuda")
        _cfg["state"] = st
    return st


def step_advance():
    L.call("asr_step_advance", L.ptr(step_state()), L.stream())


def _seed_dev(t):
    st = _cfg["state"]
    return L.ptr(st) if st is not None and st.device == t.device else None


def set_compute_dtype(dtype):
    """torch.float32 = parity mode (fp32 storage, fp32 MFMA); torch.bfloat16 = perf mode (bf16 in / fp32 accumulate)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    _cfg["dtype"] = dtype


def compute_dtype():
    return _cfg["dtype"]


def _pad8(n):
    """Leading-dimension padding of contraction axes: 8 elements (one 16-byte bf16 chunk) for small axes, a whole 128-byte
    K step (64 bf16) for large ones so that the LDS-DMA GEMM path applies (e.g. V = 4364 -> 4416)."""
    if n >= 256:
        return (n + 63) // 64 * 64
    return (n + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------ second stream
_side = {"stream": None, "enabled": os.environ.get("ASR_OVERLAP", "1") != "0"}


class fork:
    """`with ops.fork():` runs the enclosed launches on a second HIP stream, ordered after everything already enqueued on the
    current stream; `.join()` (or the next `fork.join_all()`) makes the current stream wait for them.  Used to run a layer's
    weight-gradient kernel next to its data-gradient kernel: each alone is latency bound and leaves most of the chip idle.
    Works inside a hipGraph capture (fork / join become graph edges).  ASR_OVERLAP=0 runs everything on one stream."""

    def __init__(self):
        # only while a hipGraph is being captured: replayed, the fork / join are free graph edges; issued eagerly, the extra
        # event traffic costs more host time per step than the overlap returns (measured 9.7 -> 10.0 ms)
        self.on = _side["enabled"] and (torch.cuda.is_current_stream_capturing() or os.environ.get("ASR_OVERLAP") == "2")
        if self.on:
            if _side["stream"] is None:
                _side["stream"] = torch.cuda.Stream()
            self.side = _side["stream"]
            self.main = torch.cuda.current_stream()

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)



# ---- deferred, grouped weight gradients.  Only a linear layer's DATA gradient feeds the rest of backward; its weight gradient is a
# latency-bound chain of 16 - 64 blocks when launched alone.  While a hipGraph is being captured (bf16, no data-parallel reducer
# waiting for per-layer gradients) the layers' (dy, x) pairs are queued and contracted by ONE launch per 16 layers at the end of the
# transformer's backward (asr_gemm_tn_grouped; per-layer launches everywhere else).  (Round 3: also under the multi-graph
# data-parallel step, whose reducer only exchanges between graphs.)
_wgrad_q = []
WGRAD_GROUP = 48      # layers per grouped launch (<= 48: asr_gemm_tn_grouped).  Round 5: whole-contraction blocks dispatched longest first want the LARGEST group (the headline's 46 layers are one launch of 503 blocks); round 3's equal pieces: 16 / 24 / 32 measured, profiles/r03_grouped_wgrad_group_size_ab.txt
# a group is also closed once it holds this many 64-row stages of 256 x 256 blocks (about 150 per workgroup of the scheduled kernel):
# with 12 720 rows per layer (configs[3]) groups of 16 layers measure 0.3 - 0.5 ms per step faster than groups of 32, with 6 400 rows
# groups of 32 are the faster ones -- both are ~38 000 stages.  profiles/r03_grouped_wgrad_group_size_ab.txt
# (round 5: off -- it was tuned for the equal-piece kernel; a module attribute, like WGRAD_GROUP, for tests/test_host.py -- no environment switch since round 6)
WGRAD_STAGES = 0
_wgrad_stages = [0]


# The deferred forms (grouped weight gradients, LayerNorm / one-launch-backward folds in one launch per step) need a point where
# "backward is over" is known: under a hipGraph capture every graph body ends with join_deferred(); in an EAGER backward pass the
# autograd engine calls us back when the pass has finished (queue_callback), so loss.backward() still returns complete gradients --
# the default train.py loop then issues ~60 launches per step less (round 4).  Outside both (an op called directly) nothing is deferred.
_backward_flush = {"armed": False}


def _backward_flush_cb():
    _backward_flush["armed"] = False
    join_deferred()


def deferral_ok():
    """True while a hipGraph is being captured, or inside an eager autograd backward pass whose end will flush the queues."""
    if torch.cuda.is_current_stream_capturing():
        return True
    from . import params as P_
    r = P_._state["reducer"]
    if r is not None and getattr(r, "active", False):
        return False                # an eager gradient reducer exchanges a bucket the moment its parameters report ready: nothing may trail
    if _backward_flush["armed"]:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_backward_flush_cb)
    except RuntimeError:            # not inside a backward pass
        return False
    _backward_flush["armed"] = True
    return True


def defer_wgrad_now(dtype=None):
    if (dtype or compute_dtype()) != torch.bfloat16:
        return False
    from . import params as P_
    r = P_._state["reducer"]
    # data parallel: an eager reducer wants every layer's gradient the moment its backward ran (bucket by bucket); in graph-replay
    # mode the collectives are issued BETWEEN the graphs (hold), every graph body ends with join_deferred() -> flush_wgrads(), so the
    # gradients of a graph's layers are complete before its slice is exchanged
    if not (r is None or not getattr(r, "active", False) or getattr(r, "hold", False)):
        return False
    return deferral_ok()


def queue_wgrad(dy, x, dw, db, N, K):
    """dw (N,K) fp32 += dy[:, :N]^T x[:, :K] and db (N) += column sums of dy, later (flush_wgrads).  dy and x stay referenced by the
    queue: the caller must not write to them afterwards."""
    assert gemm_tn_supported(dy, x) and dw.dtype == torch.float32 and dw.stride(1) == 1
    _wgrad_q.append((dy, x, dw, db, int(N), int(K)))
    _wgrad_stages[0] += -(-int(N) // 256) * -(-int(K) // 256) * -(-dy.shape[0] // 64)
    if len(_wgrad_q) >= WGRAD_GROUP or (WGRAD_STAGES and _wgrad_stages[0] >= WGRAD_STAGES):
        flush_wgrads()


def flush_wgrads(final=False):
    """Contract everything queued, on the launching stream (the same launches on a second stream, next to the data-gradient chain
    that follows, measured slower: profiles/r03_grouped_wgrad_ab.txt)."""
    _wgrad_stages[0] = 0
    while _wgrad_q:
        grp = _wgrad_q[:WGRAD_GROUP]
        del _wgrad_q[:WGRAD_GROUP]
        gemm_tn_grouped(grp)


def _tn_group_ok(e):
    """The layout test of asr_gemm_tn_grouped (csrc/gemm.hip) for one problem."""
    dy, x, dw, db, N, K = e
    if dy.shape[0] == 0 or N == 0 or K == 0:
        return True
    return (dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and
            dy.stride(0) >= N and dy.stride(0) < (1 << 22) and x.stride(0) < (1 << 22))


def gemm_tn_grouped(grp):
    """grp: up to 48 tuples (dy (M,>=N) bf16, x (M,>=K) bf16, dw (N,K) fp32, db (N) fp32 or None, N, K): dw += dy[:, :N]^T x[:, :K] and
    db += column sums of dy for all of them in one launch (asr_gemm_tn_grouped); problems whose layout the grouped kernel does not
    take (a row stride that is not a whole number of 16-byte chunks) go through the per-layer kernel, the others stay grouped."""
    import ctypes
    odd = [e for e in grp if not _tn_group_ok(e)]
    if odd:
        for dy, x, dw, db, N, K in odd:
            gemm_tn(dy, x, dw, colsum_acc=db, N=N, K=K)
        grp = [e for e in grp if _tn_group_ok(e)]
        if not grp:
            return
    n = len(grp)
    P_, L_, I_ = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    rc = L.load().asr_gemm_tn_grouped(
        n, P_(*[e[0].data_ptr() for e in grp]), L_(*[e[0].stride(0) for e in grp]), P_(*[e[1].data_ptr() for e in grp]),
        L_(*[e[1].stride(0) for e in grp]), P_(*[e[2].data_ptr() for e in grp]), L_(*[e[2].stride(0) for e in grp]),
        P_(*[(e[3].data_ptr() if e[3] is not None else None) for e in grp]), I_(*[e[0].shape[0] for e in grp]),
        I_(*[e[4] for e in grp]), I_(*[e[5] for e in grp]), L.dt(grp[0][0]), L.stream())
    if rc == L.EUNSUPPORTED:
        for dy, x, dw, db, N, K in grp:
            gemm_tn(dy, x, dw, colsum_acc=db, N=N, K=K)
    else:
        L.check(rc, "asr_gemm_tn_grouped")


def join_deferred():
    """Everything deferred is launched: the queued weight gradients and the partial-sum folds (called before anything reads the
    weight gradients and at the end of every captured graph body)."""
    _backward_flush["armed"] = False      # (a backward pass that raised never ran its callback: do not trust a stale flag)
    flush_wgrads()
    flush_ln_reduces()
    flush_tn_reduces()


# ------------------------------------------------------------------------------------------------ dense
def gemm_nt(A, B, out=None, bias=None, relu=False, accumulate=False, alpha=1.0, splits=1, out_dtype=None, K=None,
            relu_mask=None, N=None):
    """C[M,N] (+)= alpha * A[M,K] . B[N,K]^T (+bias) ; A, B 2-D with unit inner stride (row stride arbitrary)."""
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1 and A.dtype == B.dtype
    M = A.shape[0]
    N = B.shape[0] if N is None else N
    K = A.shape[1] if K is None else K
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=out_dtype or A.dtype)
        assert not accumulate
    assert out.stride(1) == 1 and out.shape == (M, N)
    if relu_mask is not None:
        assert relu_mask.dtype == A.dtype and relu_mask.stride(0) == out.stride(0) and relu_mask.stride(1) == 1
    flags = (L.GEMM_RELU if relu else 0) | (L.GEMM_ACCUMULATE if accumulate else 0)
    L.call("asr_gemm_nt", L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), L.ptr(out), out.stride(0), L.ptr(bias),
           L.ptr(relu_mask), M, N, K, float(alpha), flags, int(splits), L.dt(A), L.dt(out), L.stream())
    return out


def gemm_tn_supported(dy, x):
    """True when the transpose-free weight-gradient kernel applies to dy (M,N) / x (M,K)."""
    epc = 4 if dy.dtype == torch.float32 else 8
    return (dy.shape[0] > 0 and dy.stride(1) == 1 and x.stride(1) == 1 and
            dy.stride(0) % epc == 0 and x.stride(0) % epc == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def gemm_tn(dy, x, dw, colsum_acc=None, N=None, K=None, splits=0, use_ws=True):
    """dw (N,K) fp32 += dy[:, :N]^T @ x[:, :K]; colsum_acc (N) += column sums of dy.  dy, x row-major (M, ld)."""
    M = dy.shape[0]
    N = dy.shape[1] if N is None else N
    K = x.shape[1] if K is None else K
    assert dw.dtype == torch.float32 and dw.stride(1) == 1 and dy.dtype == x.dtype
    n_ws = L.load().asr_gemm_tn_workspace(M, N, K, int(splits), L.dt(dy)) if use_ws else 0
    ws = torch.empty(n_ws, device=dy.device, dtype=torch.float32) if n_ws else None
    L.call("asr_gemm_tn", L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), L.ptr(dw), dw.stride(0), L.ptr(colsum_acc), L.ptr(ws),
           n_ws, M, N, K, int(splits), L.dt(dy), L.stream())


def gemm_nn_supported(dy, w):
    """True when dx = dy (M,N) @ w (N,K) can run on the transpose-free data-gradient kernel.  N not a multiple of the kernel's
    reduction stage: dy must carry zero columns up to the next multiple (functions._as_compute / _pad_cols(x, 64) make them)."""
    bkr = 32 if dy.dtype == torch.float32 else 64
    epc = 4 if dy.dtype == torch.float32 else 8
    return (dy.dtype == w.dtype and dy.stride(0) >= (w.shape[0] + bkr - 1) // bkr * bkr and dy.shape[1] >= w.shape[0] and
            dy.stride(1) == 1 and
            w.stride(1) == 1 and dy.stride(0) % epc == 0 and w.stride(0) % epc == 0 and dy.data_ptr() % 16 == 0 and
            w.data_ptr() % 16 == 0)


def gemm_nn(dy, w, out=None, accumulate=False, relu_mask=None, alpha=1.0):
    """out (M,K) (+)= dy[:, :N] @ w (N,K), w in the weight's natural layout."""
    M, (N, K) = dy.shape[0], w.shape
    if out is None:
        out = torch.empty((M, K), device=dy.device, dtype=dy.dtype)
        assert not accumulate
    assert out.shape == (M, K) and out.stride(1) == 1
    if relu_mask is not None:
        assert relu_mask.dtype == dy.dtype and relu_mask.stride(0) == out.stride(0)
    L.call("asr_gemm_nn", L.ptr(dy), dy.stride(0), L.ptr(w), w.stride(0), L.ptr(out), out.stride(0), L.ptr(relu_mask), M, K, N,
           float(alpha), L.GEMM_ACCUMULATE if accumulate else 0, L.dt(dy), L.dt(out), L.stream())
    return out


def gemm_nn_rowdot(dy, w, o, o32, T):
    """(dx, rowdot): dx (M,K) = dy @ w (N,K) as gemm_nn, and rowdot (M / T, K / 64, T) fp32 = the sums of dx * o over each run of 64
    columns (o32, the un-rounded fp32 copy of o, is used when given) -- the attention backward's delta from the epilogue of the GEMM
    that produces dO.  None where the library has no such form (callers use gemm_nn and let attn_bwd compute delta)."""
    M, (N, K) = dy.shape[0], w.shape
    if dy.dtype != torch.bfloat16 or K % 64 != 0 or M % T != 0 or not o.is_contiguous() or o.numel() != M * K:
        return None
    dx = torch.empty((M, K), device=dy.device, dtype=dy.dtype)
    rowdot = torch.empty((M // T, K // 64, T), device=dy.device, dtype=torch.float32)
    rc = L.load().asr_gemm_nn_rowdot(L.ptr(dy), dy.stride(0), L.ptr(w), w.stride(0), L.ptr(dx), L.ptr(o), L.ptr(o32), L.ptr(rowdot),
                                     M, K, N, T, L.dt(dy), L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_gemm_nn_rowdot")
    return dx, rowdot


_nn_tn = True
# larger weights keep the two-stream pair: their 128 x 128-tile weight-gradient kernel moves half the operand bytes per flop, which
# is worth more than the fork / join it costs (512 x 5120 over 6400 rows: 112 us as a pair, 141 us as one launch)
_nn_tn_max = 1 << 21
_tn_fold_next = True
_tn_pending = []


def gemm_nn_tn_supported(dy, w, x):
    """True when a linear layer's dX and dW can be ONE launch (asr_gemm_nn_tn): bf16, inside a graph capture (the weight
    gradient is then complete only after flush_tn_reduces(), which join_deferred() issues at the end of backward)."""
    return (_nn_tn and dy.dtype == torch.bfloat16 and gemm_nn_supported(dy, w) and deferral_ok() and
            gemm_tn_supported(dy, x) and x.dtype == dy.dtype and w.shape[0] * w.shape[1] <= _nn_tn_max)


def gemm_nn_tn(dy, w, x, dw, db=None, out=None, accumulate=False, relu_mask=None):
    """out (M,K) (+)= dy[:, :N] @ w (N,K) and, in the same launch, the partial sums of dw (N,K) fp32 += dy^T @ x[:, :K] (folded
    into dw by flush_tn_reduces()) and db (N) += column sums of dy."""
    M, (N, K) = dy.shape[0], w.shape
    if out is None:
        out = torch.empty((M, K), device=dy.device, dtype=dy.dtype)
        assert not accumulate
    assert out.shape == (M, K) and out.stride(1) == 1 and out.dtype == dy.dtype
    assert dw.dtype == torch.float32 and dw.stride(1) == 1 and dw.shape == (N, K) and x.shape[0] == M and x.shape[1] >= K
    if relu_mask is not None:
        assert relu_mask.dtype == dy.dtype and relu_mask.stride(0) == out.stride(0)
    lib = L.load()
    n_ws = lib.asr_gemm_nn_tn_workspace(M, N, K, 0)
    ws = torch.empty(n_ws, device=dy.device, dtype=torch.float32)       # stays referenced until it has been folded
    # the previous layer's partial tiles are folded by extra workgroups of THIS launch (same stream: they are complete)
    prev = _tn_pending.pop() if (_tn_fold_next and _tn_pending) else None
    fold = (L.ptr(prev[0]), L.ptr(prev[1]), prev[1].stride(0), prev[2], prev[3], prev[4]) if prev else (None, None, 0, 0, 0, 0)
    L.call("asr_gemm_nn_tn", L.ptr(dy), dy.stride(0), L.ptr(w), w.stride(0), L.ptr(x), x.stride(0), L.ptr(out), out.stride(0),
           L.ptr(relu_mask), L.ptr(db), L.ptr(ws), n_ws, M, N, K, L.GEMM_ACCUMULATE if accumulate else 0, 0, L.dt(dy), *fold,
           L.stream())
    _tn_pending.append((ws, dw, N, K, lib.asr_gemm_nn_tn_splits(M, 0)))
    return out


def reset_pending():
    """Forget everything deferred that was never issued -- a capture or a backward pass that raised half way: queued weight
    gradients (their tensors belong to the failed batch), second stages (their workspaces are gone) and the armed flush flag (the
    autograd engine drops its final callbacks on error).  Called by the graph capture's error path and by FusedAdam.zero_grad():
    a new step never inherits a failed one's queue."""
    del _wgrad_q[:]
    _wgrad_stages[0] = 0
    del _tn_pending[:]
    del _ln_pending[:]
    _backward_flush["armed"] = False


def flush_tn_reduces():
    """Second stage of every gemm_nn_tn() issued since the last flush: one launch (per 48 layers) folds all partial tiles."""
    import ctypes
    if not _tn_pending:
        return
    n = len(_tn_pending)
    P_, I_, L_ = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_int64 * n
    L.call("asr_tn_reduce_multi", P_(*[e[0].data_ptr() for e in _tn_pending]), P_(*[e[1].data_ptr() for e in _tn_pending]),
           L_(*[e[1].stride(0) for e in _tn_pending]), I_(*[e[2] for e in _tn_pending]), I_(*[e[3] for e in _tn_pending]),
           I_(*[e[4] for e in _tn_pending]), n, L.stream())
    del _tn_pending[:]


def cast_flat(src, dst):
    L.call("asr_cast_flat", L.ptr(src), L.ptr(dst), src.numel(), L.dt(dst), L.stream())


def widen_flat(src_bf16, dst):
    """dst (fp32) = src (bf16), elementwise: the way back from the bf16 gradient wire format (asr_hip/ddp.py)."""
    assert src_bf16.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src_bf16.numel() == dst.numel()
    L.call("asr_widen_flat", L.ptr(src_bf16), L.ptr(dst), dst.numel(), L.stream())


def transpose_padded(x, colsum_acc=None):
    """(rows, cols) -> (cols, pad8(rows)) with zero pad columns (so a contraction may run over the padded axis).
    colsum_acc (fp32, cols): optionally accumulate the column sums of x (bias gradient) in the same pass."""
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    ld = _pad8(rows)
    out = torch.zeros((cols, ld), device=x.device, dtype=x.dtype) if ld != rows else \
        torch.empty((cols, ld), device=x.device, dtype=x.dtype)
    L.call("asr_transpose", L.ptr(x), x.stride(0), L.ptr(out), ld, rows, cols, L.ptr(colsum_acc), L.dt(x), L.stream())
    return out


def transpose(x):
    return transpose_padded(x)[:, :x.shape[0]]


def cast_and_transpose(src, dtype, want_same=True, want_t=True, pad=8):
    """fp32 (rows, cols) -> (copy in dtype with ld padded to `pad`, transpose in dtype with ld padded to 8).
    Pad columns are zero so a contraction may run over the padded K."""
    assert src.dim() == 2 and src.stride(1) == 1 and src.dtype == torch.float32
    rows, cols = src.shape
    same = t = None
    if want_same:
        same = torch.zeros((rows, (cols + pad - 1) // pad * pad), device=src.device, dtype=dtype)
    if want_t:
        t = torch.zeros((cols, _pad8(rows)), device=src.device, dtype=dtype)
    L.call("asr_cast_weight", L.ptr(src), src.stride(0), L.ptr(same), same.stride(0) if same is not None else 0,
           L.ptr(t), t.stride(0) if t is not None else 0, rows, cols, L.dt_of(dtype), L.stream())
    return same, t


def cast_into(src, same, t):
    """Refresh persistent shadow buffers (same: (rows, ld), t: (cols, ld_t)) from the fp32 master `src`."""
    rows, cols = src.shape
    dtype = (same if same is not None else t).dtype
    L.call("asr_cast_weight", L.ptr(src), src.stride(0), L.ptr(same), same.stride(0) if same is not None else 0,
           L.ptr(t), t.stride(0) if t is not None else 0, rows, cols, L.dt_of(dtype), L.stream())


def colsum_acc(x, out):
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    L.call("asr_colsum_acc", L.ptr(x), x.stride(0), x.shape[0], x.shape[1], L.ptr(out), L.dt(x), L.stream())


# ------------------------------------------------------------------------------------------------ layer norm
def add_ln_fwd(y, residual, gamma, beta, post_add=None, row_keep=None, eps=1e-5, p=0.0, seed=0):
    """y (M,D) contiguous is overwritten with z = dropout(y)+residual.  Returns (out, mean, rstd)."""
    M, D = y.shape
    assert y.is_contiguous() and (residual is None or residual.is_contiguous())
    out = torch.empty_like(y)
    mean = torch.empty(M, device=y.device, dtype=torch.float32)
    rstd = torch.empty(M, device=y.device, dtype=torch.float32)
    period = post_add.shape[0] if post_add is not None else 0
    L.call("asr_add_ln_fwd", L.ptr(y), L.ptr(residual), L.ptr(gamma), L.ptr(beta), L.ptr(post_add), period,
           L.ptr(row_keep), L.ptr(out), L.ptr(mean), L.ptr(rstd), M, D, float(eps), float(p), int(seed), _seed_dev(y),
           L.dt(y), L.stream())
    return out, mean, rstd


def add_ln_bwd(dout, z, mean, rstd, gamma, row_keep, dgamma, dbeta, p=0.0, seed=0):
    """Returns (d_res, d_y); d_y is d_res itself when p == 0 (unless weight gradients are being deferred)."""
    M, D = z.shape
    assert dout.is_contiguous() and z.is_contiguous()
    d_res = torch.empty_like(z)
    # (deferred weight gradients keep d_y until the end of backward while d_res goes on accumulating: never the same buffer then)
    d_y = torch.empty_like(z) if (p > 0 or defer_wgrad_now(z.dtype)) else d_res
    n_ws = L.load().asr_add_ln_bwd_workspace(M, D)
    ws = torch.empty(n_ws, device=z.device, dtype=torch.float32)       # caching allocator: no cost after the first step
    if D % 2 == 0 and deferral_ok():
        # graph capture: the dgamma / dbeta sums of all layers in one launch at the end of backward (flush_ln_reduces)
        L.call("asr_add_ln_bwd_partials", L.ptr(dout), L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(row_keep),
               L.ptr(d_res), L.ptr(d_y), L.ptr(ws), n_ws, M, D, float(p), int(seed), _seed_dev(z), L.dt(z), L.stream())
        _ln_pending.append((ws, M, D, dgamma, dbeta))
        return d_res, d_y
    L.call("asr_add_ln_bwd", L.ptr(dout), L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(row_keep),
           L.ptr(d_res), L.ptr(d_y), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), n_ws, M, D, float(p), int(seed), _seed_dev(z),
           L.dt(z), L.stream())
    return d_res, d_y


_ln_pending = []


def flush_ln_reduces():
    """Second stage of every LayerNorm backward issued since the last flush (same hidden size: one launch)."""
    import ctypes
    while _ln_pending:
        D = _ln_pending[0][2]
        grp = [e for e in _ln_pending if e[2] == D]
        _ln_pending[:] = [e for e in _ln_pending if e[2] != D]
        n = len(grp)
        P_, I_ = ctypes.c_void_p * n, ctypes.c_int * n
        L.call("asr_ln_reduce_multi", P_(*[e[0].data_ptr() for e in grp]), I_(*[e[1] for e in grp]),
               P_(*[e[3].data_ptr() for e in grp]), P_(*[e[4].data_ptr() for e in grp]), n, D, L.stream())


# ------------------------------------------------------------------------------------------------ attention
def _bt_strides(x, H, d):
    assert x.dim() == 3 and x.stride(2) == 1 and x.shape[2] == H * d
    return x.stride(0), x.stride(1)


def _mask_strides(mask, B, Tq, Tk):
    """uint8 mask: (B,Tk) per-key or (B,Tq,Tk) full -> (batch stride, query stride)."""
    if mask is None:
        return 0, 0
    assert mask.dtype == torch.uint8 and mask.is_contiguous()
    if mask.dim() == 2:
        assert mask.shape == (B, Tk)
        return Tk, 0
    assert mask.shape == (B, Tq, Tk)
    return Tq * Tk, Tk


def attn_fwd(q, k, v, H, d, key_len=None, key_pad=None, causal=False, scale=1.0, p=0.0, seed=0, want_attn=False, o32=None):
    """q (B,Tq,H*d), k/v (B,Tk,H*d) -> (o (B,Tq,H*d), lse (B,H,Tq), attn (H*B,Tq,Tk) or None).
    o32: optional fp32 (B,Tq,H*d) contiguous destination for an un-rounded copy of o (training with bf16 storage)."""
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    msb, msq = _mask_strides(key_pad, B, Tq, Tk)
    o = torch.empty((B, Tq, H * d), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    attn = torch.empty((H * B, Tq, Tk), device=q.device, dtype=torch.float32) if want_attn else None
    qs, ks, vs, os_ = _bt_strides(q, H, d), _bt_strides(k, H, d), _bt_strides(v, H, d), _bt_strides(o, H, d)
    assert o32 is None or (o32.dtype == torch.float32 and o32.shape == o.shape and o32.is_contiguous())
    L.call("asr_attn_fwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(o32), L.ptr(lse), L.ptr(attn), B, H, Tq, Tk, d, qs[0], qs[1],
           ks[0], ks[1], vs[0], vs[1], os_[0], os_[1], L.ptr(key_len), L.ptr(key_pad), msb, msq, int(causal),
           float(scale), float(p), int(seed), _seed_dev(q), L.dt(q), L.stream())
    return o, lse, attn


def attn_bwd(q, k, v, o, do, lse, H, d, key_len=None, key_pad=None, causal=False, scale=1.0, p=0.0, seed=0, out=None, o32=None,
             delta=None):
    """out: optional (dq, dk, dv) destination tensors; they must have the strides of q, k, v (the ABI reuses them).
    delta: rowsum(dO * O) (B, H, Tq) fp32 when the caller already has it (gemm_nn_rowdot) -- the delta launch is then skipped."""
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    assert do.is_contiguous() and o.is_contiguous()
    assert o32 is None or (o32.dtype == torch.float32 and o32.shape == o.shape and o32.is_contiguous())
    if out is None:
        dq = torch.empty_strided(q.shape, q.stride(), device=q.device, dtype=q.dtype)
        dk = torch.empty_strided(k.shape, k.stride(), device=k.device, dtype=k.dtype)
        dv = torch.empty_strided(v.shape, v.stride(), device=v.device, dtype=v.dtype)
    else:
        dq, dk, dv = out
    assert dq.stride() == q.stride() and dk.stride() == k.stride() and dv.stride() == v.stride()
    have_delta = delta is not None
    if have_delta:
        assert delta.shape == (B, H, Tq) and delta.dtype == torch.float32 and delta.is_contiguous()
    else:
        delta = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    msb, msq = _mask_strides(key_pad, B, Tq, Tk)
    qs, ks, vs, os_ = _bt_strides(q, H, d), _bt_strides(k, H, d), _bt_strides(v, H, d), _bt_strides(o, H, d)
    sd = _seed_dev(q)

    def launch(parts):
        L.call("asr_attn_bwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(o32), L.ptr(do), L.ptr(lse), L.ptr(delta), L.ptr(dq),
               L.ptr(dk), L.ptr(dv), B, H, Tq, Tk, d, qs[0], qs[1], ks[0], ks[1], vs[0], vs[1], os_[0], os_[1],
               L.ptr(key_len), L.ptr(key_pad), msb, msq, int(causal), float(scale), float(p), int(seed), sd, parts,
               L.dt(q), L.stream())

    launch(L.ATTN_DQ | L.ATTN_DKV if have_delta else L.ATTN_ALL)       # one launch for dQ and dK / dV (two launches on two streams cost a fork and a join: profiles/r02_ab_attn_both.txt)
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------ decoder input
def decoder_preprocess(tgt, Td):
    B, Lw = tgt.shape
    tgt = tgt.contiguous()
    dev = tgt.device
    seq_in = torch.empty((B, Td), device=dev, dtype=torch.int64)
    seq_out = torch.empty((B, Td), device=dev, dtype=torch.int64)
    key_pad = torch.empty((B, Td), device=dev, dtype=torch.uint8)
    row_keep = torch.empty((B, Td), device=dev, dtype=torch.uint8)
    # the flag is only ever read when a target CAN be too long for Td (models/asr/transformer.py): otherwise a scratch word, no fill launch
    overflow = torch.zeros(1, device=dev, dtype=torch.int32) if Lw + 1 > Td else _scratch_word(dev)
    L.call("asr_decoder_preprocess", L.ptr(tgt), B, Lw, Td, L.ptr(seq_in), L.ptr(seq_out), L.ptr(key_pad),
           L.ptr(row_keep), L.ptr(overflow), L.stream())
    return seq_in, seq_out, key_pad, row_keep, overflow


def embed_fwd(tok, table, pe, scale, p, seed, dtype):
    B, T = tok.shape
    D = table.shape[1]
    out = torch.empty((B, T, D), device=tok.device, dtype=dtype)
    L.call("asr_embed_fwd", L.ptr(tok), L.ptr(table), L.ptr(pe), L.ptr(out), B, T, D, float(scale), float(p), int(seed),
           _seed_dev(tok), L.dt_of(dtype), L.stream())
    return out


def embed_bwd(tok, dout, dtable, scale, p, seed, pad_id):
    B, T = tok.shape
    D = dtable.shape[1]
    assert dout.is_contiguous()
    L.call("asr_embed_bwd", L.ptr(tok), L.ptr(dout), L.ptr(dtable), B, T, D, float(scale), float(p), int(seed),
           _seed_dev(tok), int(pad_id), L.dt(dout), L.stream())


# ------------------------------------------------------------------------------------------------ loss
def ce_fwd(logits, gold, smoothing, pad_id, sums=None):
    """logits (M,V) fp32 -> (row_lse (M), argmax (M) int64, sums (3) fp32 = [loss_sum, count, num_correct]).
    sums: optional ZEROED fp32 destination (>= 3 elements) the kernel adds into, e.g. FlatParams.stats."""
    M, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and gold.is_contiguous()
    lse = torch.empty(M, device=logits.device, dtype=torch.float32)
    am = torch.empty(M, device=logits.device, dtype=torch.int64)
    if sums is None:
        sums = torch.zeros(3, device=logits.device, dtype=torch.float32)
    assert sums.dtype == torch.float32 and sums.numel() >= 3 and sums.is_contiguous()
    L.call("asr_ce_fwd", L.ptr(logits), logits.stride(0), L.ptr(gold), M, V, float(smoothing), int(pad_id), L.ptr(lse),
           L.ptr(am), L.ptr(sums), L.stream())
    return lse, am, sums


def ce_fwd_det(logits, gold, smoothing, pad_id, den=None):
    """ce_fwd with reproducible statistics: per-block partial sums added in a fixed order, no zeroed destination, the loss
    (= sums[0] / den, den = the non-PAD count unless a device scalar is given) from the same finish launch.
    -> (row_lse, argmax, sums (3), loss (1))."""
    M, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and gold.is_contiguous()
    dev = logits.device
    lse = torch.empty(M, device=dev, dtype=torch.float32)
    am = torch.empty(M, device=dev, dtype=torch.int64)
    nb = int(L.load().asr_ce_partial_blocks(M))
    part = torch.empty(max(nb, 1) * 3, device=dev, dtype=torch.float32)
    sums = torch.empty(3, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    L.call("asr_ce_fwd_partials", L.ptr(logits), logits.stride(0), L.ptr(gold), M, V, float(smoothing), int(pad_id), L.ptr(lse),
           L.ptr(am), L.ptr(part), L.stream())
    L.call("asr_ce_finish", L.ptr(part), nb, L.ptr(den), L.ptr(sums), L.ptr(loss), L.stream())
    return lse, am, sums, loss


# ------------------------------------------------------------------------------------------------ fp8 projections
_fp8 = {"on": False}


def set_fp8(on):
    """--precision fp8: the low-rank projections run their forward GEMMs on the fp8 MFMA (storage stays bf16)."""
    _fp8["on"] = bool(on)


def fp8_enabled():
    return _fp8["on"]


def quant_fp8(x):
    """x (M,K) bf16/fp32 -> (q (M, Kp) uint8 e4m3 bytes, scale (M,) fp32: row amax / 448)."""
    M, K = x.shape
    assert x.stride(1) == 1
    Kp = (K + 15) // 16 * 16
    q = torch.empty((M, Kp), device=x.device, dtype=torch.uint8)
    scale = torch.empty(M, device=x.device, dtype=torch.float32)
    L.call("asr_quant_fp8", L.ptr(x), x.stride(0), M, K, L.dt(x), L.ptr(q), Kp, L.ptr(scale), L.stream())
    return q, scale


def gemm_nt_fp8(qa, sa, qb, sb, bias=None, relu=False, out_dtype=torch.bfloat16, K=None):
    """C[m,n] = sa[m] sb[n] (qa . qb^T)[m,n] (+ bias)(ReLU): qa (M,Kp), qb (N,Kp) e4m3 bytes and row scales from quant_fp8."""
    M, N = qa.shape[0], qb.shape[0]
    K = qa.shape[1] if K is None else K
    out = torch.empty((M, N), device=qa.device, dtype=out_dtype)
    L.call("asr_gemm_nt_fp8", L.ptr(qa), qa.stride(0), L.ptr(sa), L.ptr(qb), qb.stride(0), L.ptr(sb), L.ptr(out), out.stride(0),
           L.ptr(bias), M, N, K, int(relu), L.dt_of(out_dtype), L.stream())
    return out


def ctc_fwd(logits, targets, input_lengths, target_lengths, blank=0):
    """logits (B,T,V) fp32, targets (B,Lmax) int64, lengths (B) int32 on the device -> (loss (1,), workspace)."""
    B, T, V = logits.shape
    Lmax = max(1, targets.shape[1])
    assert logits.dtype == torch.float32 and logits.is_contiguous() and targets.dtype == torch.int64 and targets.is_contiguous()
    assert input_lengths.dtype == torch.int32 and target_lengths.dtype == torch.int32 and targets.shape[1] >= 1
    n = L.load().asr_ctc_workspace(B, T, Lmax)
    ws = torch.empty(n, device=logits.device, dtype=torch.float32)
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    L.call("asr_ctc_fwd", L.ptr(logits), V, L.ptr(targets), L.ptr(input_lengths), L.ptr(target_lengths), B, T, V, Lmax, int(blank),
           L.ptr(ws), n, L.ptr(loss), L.stream())
    return loss, ws


def ctc_bwd(logits, targets, input_lengths, target_lengths, ws, grad_out, blank=0):
    B, T, V = logits.shape
    Lmax = max(1, targets.shape[1])
    dl = torch.empty((B, T, V), device=logits.device, dtype=torch.float32)
    L.call("asr_ctc_bwd", L.ptr(logits), V, L.ptr(targets), L.ptr(input_lengths), L.ptr(target_lengths), B, T, V, Lmax, int(blank),
           L.ptr(ws), L.ptr(grad_out), L.ptr(dl), V, L.stream())
    return dl


def decode_prepare(pe, pe_cur, key_len, state):
    """pe (T,D) fp32, state (>=1) int64 on the device: pe_cur = pe[state[0]], key_len[:] = state[0] + 1."""
    assert pe.dtype == torch.float32 and pe.is_contiguous() and pe_cur.dtype == torch.float32 and key_len.dtype == torch.int32
    L.call("asr_decode_prepare", L.ptr(pe), pe.shape[1], L.ptr(pe_cur), L.ptr(key_len), key_len.numel(), L.ptr(state), 0, L.stream())


def decode_advance(state):
    L.call("asr_decode_prepare", None, 0, None, None, 0, L.ptr(state), 1, L.stream())


def kv_append(k_src, v_src, k_cache, v_cache, state):
    """k_src / v_src (B, ncols) (row stride free) -> k_cache / v_cache (B, max_len, ncols) at row state[0]."""
    B, ncols = k_src.shape
    assert k_src.stride(1) == 1 and v_src.stride(1) == 1 and k_src.stride(0) == v_src.stride(0)
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == (B, k_cache.shape[1], ncols)
    L.call("asr_kv_append", L.ptr(k_src), L.ptr(v_src), k_src.stride(0), L.ptr(k_cache), L.ptr(v_cache), B, ncols,
           k_cache.shape[1], L.ptr(state), L.dt(k_src), L.stream())


def frag_pack(w):
    """(rows, K) -> the fragment-major order of asr_dec_gemm (include/asr_hip.h): rows padded with zeros to a multiple of 32,
    blocks of 32 rows x 16 columns, lane 32 h + i of a block holds [i][8 h .. 8 h + 7].  Returns a flat tensor."""
    rows, K = w.shape
    assert K % 16 == 0
    rp = (rows + 31) // 32 * 32
    if rp != rows:
        w = torch.cat([w, w.new_zeros(rp - rows, K)], 0)
    return w.view(rp // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def frag_unpack(f, rows, K):
    """Inverse of frag_pack (tests)."""
    rp = (rows + 31) // 32 * 32
    return f.view(rp // 32, K // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).contiguous().view(rp, K)[:rows]


def dec_gemm(W, bias, out, x=None, ln=None, embed=None, x_out=None, relu=False, w_frag=None, x_frag=False, out_frag=False, B=None):
    """out (B, N) = act(x W^T + bias) for B <= 32 decode rows (asr_dec_gemm).  W (N, K) bf16, or w_frag = (N, K) with W the
    flat frag_pack()ed weight.  Exactly one input: x (B, K) bf16 (x_frag: flat fragment-major, 32 rows);
    ln = (Y, R, gamma, beta, eps): x = LayerNorm(Y + R);  embed = (tok, table, pe, scale, state):
    x = table[tok] * scale + pe[state[0]].  x_out (B, K): the prologue's x is also stored there.  out_frag: out is a flat
    fragment-major (32, N) bf16 buffer and B is taken from x / Y / tok."""
    if w_frag is not None:
        N, K = w_frag
        ldw = K
        assert W.dtype == torch.bfloat16 and W.is_contiguous() and W.numel() == (N + 31) // 32 * 32 * K
    else:
        N, K = W.shape
        ldw = W.stride(0)
        assert W.dtype == torch.bfloat16 and W.stride(1) == 1
    layout = (1 if w_frag is not None else 0) | (2 if x_frag else 0) | (4 if out_frag else 0)
    if out_frag:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == 32 * N
        if B is None:
            B = (x if x is not None and not x_frag else (ln[0] if ln is not None else embed[0])).shape[0]
        ldo = N
    else:
        B = out.shape[0]
        assert out.stride(1) == 1 and out.shape[1] == N
        ldo = out.stride(0)
    pro = 0 if x is not None else (1 if ln is not None else 2)
    X = Y = R = g = bt = tok = table = pe = state = None
    eps, scale, ldx = 0.0, 1.0, 0
    if pro == 0 and x_frag:
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() == 32 * K
        X, ldx = x, K
    elif pro == 0:
        assert x.dtype == torch.bfloat16 and x.shape == (B, K) and x.stride(1) == 1
        X, ldx = x, x.stride(0)
    elif pro == 1:
        Y, R, g, bt, eps = ln
        assert Y.dtype == R.dtype == torch.bfloat16 and Y.shape == R.shape == (B, K) and Y.is_contiguous() and R.is_contiguous()
        assert g.dtype == bt.dtype == torch.float32
    else:
        tok, table, pe, scale, state = embed
        assert tok.dtype == torch.int64 and table.dtype == pe.dtype == torch.float32 and table.shape[1] == K == pe.shape[1]
        assert table.is_contiguous() and pe.is_contiguous()
    if x_out is not None:
        assert x_out.dtype == torch.bfloat16 and x_out.shape == (B, K) and x_out.is_contiguous()
    L.call("asr_dec_gemm", L.ptr(W), ldw, L.ptr(bias), L.ptr(out), ldo, B, N, K, int(relu), L.dt(out), layout, pro,
           L.ptr(X), ldx, L.ptr(Y), L.ptr(R), L.ptr(g), L.ptr(bt), float(eps), L.ptr(x_out), L.ptr(tok), L.ptr(table), L.ptr(pe),
           float(scale), L.ptr(state), L.stream())
    return out


def dec_attn(q, k_cache, v_cache, out, H, dk, scale, k_new=None, v_new=None, state=None, out_frag=False):
    """One query row per sequence and head (asr_dec_attn).  q (B, H*dk) (row stride free); k_cache / v_cache (B, rows, H*dk)
    (batch stride may be 0).  state given: self attention at t = state[0] with k_new / v_new (B, H*dk) appended at row t."""
    B = q.shape[0]
    assert q.dtype == torch.bfloat16 and q.stride(1) == 1 and k_cache.stride(2) == 1 and k_cache.stride() == v_cache.stride()
    if out_frag:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == 32 * H * dk
        ldo = H * dk
    else:
        assert out.stride(1) == 1 and out.shape == (B, H * dk)
        ldo = out.stride(0)
    ldn = 0
    if k_new is not None:
        assert k_new.stride(1) == 1 and v_new.stride(1) == 1 and k_new.stride(0) == v_new.stride(0)
        ldn = k_new.stride(0)
    L.call("asr_dec_attn", L.ptr(q), q.stride(0), L.ptr(k_new), L.ptr(v_new), ldn, L.ptr(k_cache), L.ptr(v_cache), k_cache.stride(0),
           k_cache.stride(1), k_cache.shape[1], L.ptr(out), ldo, B, H, dk, float(scale), int(out_frag), L.ptr(state), L.stream())
    return out


def dec_attn_fused(W, bias, k_cache, v_cache, out, H, dk, scale, ln=None, embed=None, x_out=None, state=None, self_attention=False,
                   out_frag=False):
    """LayerNorm / embedding prologue + this head's projections + single-query attention in one launch (asr_dec_attn_fused).
    W ((3 if self_attention else 1) * H*dk, D) bf16 row-major; ln = (Y, R, gamma, beta, eps) or embed = (tok, table, pe, scale)."""
    NW, D = W.shape
    assert W.dtype == torch.bfloat16 and W.is_contiguous() and NW == (3 if self_attention else 1) * H * dk
    assert k_cache.stride(2) == 1 and k_cache.stride() == v_cache.stride()
    Y = R = g = bt = tok = table = pe = None
    eps, es = 0.0, 1.0
    if ln is not None:
        Y, R, g, bt, eps = ln
        B = Y.shape[0]
        assert Y.dtype == R.dtype == torch.bfloat16 and Y.shape == R.shape == (B, D) and Y.is_contiguous() and R.is_contiguous()
    else:
        tok, table, pe, es = embed
        B = tok.shape[0]
        assert tok.dtype == torch.int64 and table.dtype == pe.dtype == torch.float32 and table.shape[1] == D == pe.shape[1]
    if out_frag:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == 32 * H * dk
        ldo = H * dk
    else:
        assert out.stride(1) == 1 and out.shape == (B, H * dk)
        ldo = out.stride(0)
    if x_out is not None:
        assert x_out.dtype == torch.bfloat16 and x_out.shape == (B, D) and x_out.is_contiguous()
    L.call("asr_dec_attn_fused", L.ptr(W), L.ptr(bias), D, int(self_attention), L.ptr(Y), L.ptr(R), L.ptr(g), L.ptr(bt), float(eps),
           L.ptr(tok), L.ptr(table), L.ptr(pe), float(es), L.ptr(x_out), L.ptr(k_cache), L.ptr(v_cache), k_cache.stride(0),
           k_cache.stride(1), k_cache.shape[1], L.ptr(out), ldo, B, H, dk, float(scale), int(out_frag), L.ptr(state), L.stream())
    return out


def dec_finish(logits, tok, done, out, eos, state, ticket):
    """tok = argmax(logits) per row, done |= tok == eos, out[state[0]] = tok, state[0] += 1 (asr_dec_finish)."""
    B, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and tok.dtype == torch.int64 and done.dtype == torch.bool
    assert out.dtype == torch.int64 and out.is_contiguous() and out.shape[1] == B and ticket.dtype == torch.int32
    L.call("asr_dec_finish", L.ptr(logits), logits.stride(0), V, L.ptr(tok), L.ptr(done), L.ptr(out), B, out.shape[0], int(eos),
           L.ptr(state), L.ptr(ticket), L.stream())


def argmax_rows(logits, out=None):
    M, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    if out is None:
        out = torch.empty(M, device=logits.device, dtype=torch.int64)
    assert out.dtype == torch.int64 and out.numel() == M and out.is_contiguous()
    L.call("asr_argmax_rows", L.ptr(logits), logits.stride(0), M, V, L.ptr(out), L.stream())
    return out


def logsoftmax_topk(logits, k):
    """(values (M,k) fp32, indices (M,k) int64) of the k largest log-softmax entries per row, best first (beam-search scoring)."""
    M, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    vals = torch.empty((M, k), device=logits.device, dtype=torch.float32)
    idx = torch.empty((M, k), device=logits.device, dtype=torch.int64)
    L.call("asr_logsoftmax_topk", L.ptr(logits), logits.stride(0), M, V, k, L.ptr(vals), L.ptr(idx), L.stream())
    return vals, idx


def ce_bwd(logits, gold, lse, smoothing, pad_id, grad_out, count, out_dtype=torch.float32, pad=8):
    """Returns dlogits as an (M,V) view of an (M, V rounded up to `pad`) buffer whose pad columns are zero; with a 64-column pad
    the WHOLE padded buffer is returned (the data-gradient GEMM contracts the padded width)."""
    M, V = logits.shape
    ld = (V + pad - 1) // pad * pad
    dl = torch.empty((M, ld), device=logits.device, dtype=out_dtype)
    L.call("asr_ce_bwd", L.ptr(logits), logits.stride(0), L.ptr(gold), L.ptr(lse), M, V, float(smoothing), int(pad_id),
           L.ptr(grad_out), L.ptr(count), L.ptr(dl), ld, L.dt(dl), L.stream())
    return dl if pad == 64 else dl[:, :V]


# ------------------------------------------------------------------------------------------------ optimiser
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=None):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    L.call("asr_adam_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
           float(eps), float(bc1), float(bc2), L.ptr(grad_scale), L.stream())


def adam_noam_step(p, g, m, v, beta1, beta2, eps, factor_ms, warmup, min_lr, grad_scale=None, lr_out=None, guard=None, shadow=None):
    """Adam update whose step count / Noam lr / bias corrections come from step_state()[1] on the device.  `guard`: optional
    device scalar (the step's loss sum); a non-finite guard or gradient scale leaves parameters and moments untouched.  `shadow`:
    optional bf16 tensor of p's size that receives the rounded new parameters (the flat compute-dtype shadow)."""
    assert shadow is None or (shadow.dtype == torch.bfloat16 and shadow.numel() == p.numel())
    L.call("asr_adam_noam_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(step_state(p.device)),
           float(beta1), float(beta2), float(eps), float(factor_ms), float(warmup), float(min_lr), L.ptr(grad_scale),
           L.ptr(lr_out), L.ptr(guard), L.ptr(shadow), L.stream())


def sumsq_acc(g, acc):
    L.call("asr_sumsq_acc", L.ptr(g), g.numel(), L.ptr(acc), L.stream())


def clip_coef(sumsq, max_norm, coef):
    L.call("asr_clip_coef", L.ptr(sumsq), float(max_norm), L.ptr(coef), L.stream())


def length_mask(lengths, T):
    """lengths (B) int32 on the device -> row_keep (B*T) uint8, 1 where t < length."""
    B = lengths.shape[0]
    out = torch.empty(B * T, device=lengths.device, dtype=torch.uint8)
    L.call("asr_length_mask", L.ptr(lengths), B, int(T), L.ptr(out), L.stream())
    return out


def ratio(num, den):
    """(1,) fp32 num / den on the device."""
    out = torch.empty(1, device=num.device, dtype=torch.float32)
    L.call("asr_ratio", L.ptr(num), L.ptr(den), L.ptr(out), L.stream())
    return out


_ones = {}
_consts = {}


def zero_scalar(device, dtype=torch.float32):
    """A shared 0-dim zero (read-only by convention: formal gradients that nobody consumes)."""
    t = _consts.get(("zero", str(device), dtype))
    if t is None:
        t = torch.zeros((), device=device, dtype=dtype)
        _consts[("zero", str(device), dtype)] = t
    return t


def _scratch_word(device):
    t = _consts.get(("scratch", str(device)))
    if t is None:
        t = torch.zeros(1, device=device, dtype=torch.int32)
        _consts[("scratch", str(device))] = t
    return t


def backward_from(loss):
    """loss.backward() without the ones_like() fill launch autograd would seed it with: the seed is a cached device scalar."""
    loss.backward(ones_scalar(loss.device).view(loss.shape))


def ones_scalar(device):
    t = _ones.get(str(device))
    if t is None:
        t = torch.ones(1, device=device, dtype=torch.float32)
        _ones[str(device)] = t
    return t


def grad_coef(sumsq, max_norm, denom, coef):
    """coef = (1/denom) * clip coefficient of the (1/denom)-scaled gradients; sumsq None = no clipping."""
    L.call("asr_grad_coef", L.ptr(sumsq), float(max_norm), L.ptr(denom), L.ptr(coef), L.stream())


# ------------------------------------------------------------------------------------------------ conv front end
def conv1_fwd(x, w, bias, dtype):
    B, _, H, W = x.shape
    C0 = w.shape[0]
    assert x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty((B, H, W, C0), device=x.device, dtype=dtype)
    L.call("asr_conv1_fwd", L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, W, C0, L.dt_of(dtype), L.stream())
    return y


def conv1_wgrad(x, dy, dw, db):
    B, H, W, C0 = dy.shape
    L.call("asr_conv1_wgrad", L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), B, H, W, C0, L.dt(dy), L.stream())


def conv_pack_weight(w, wk, wd):
    Cout, Cin = w.shape[0], w.shape[1]
    t = wk if wk is not None else wd
    L.call("asr_conv_pack_weight", L.ptr(w), L.ptr(wk), L.ptr(wd), Cout, Cin, L.dt(t), L.stream())


def conv_pack_weight_multi(items):
    """items: up to 8 (w (Cout,Cin,3,3) fp32, wk, wd) triples of one dtype -> all packed by one launch."""
    import ctypes
    n = len(items)
    if n == 0:
        return
    P_, I_ = ctypes.c_void_p * n, ctypes.c_int * n
    L.call("asr_conv_pack_weight_multi", n, P_(*[w.data_ptr() for w, _, _ in items]), P_(*[wk.data_ptr() for _, wk, _ in items]),
           P_(*[wd.data_ptr() for _, _, wd in items]), I_(*[w.shape[0] for w, _, _ in items]), I_(*[w.shape[1] for w, _, _ in items]),
           L.dt(items[0][1]), L.stream())


def conv3x3(x, wk, bias, Cout, relu, mask_src=None):
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=x.dtype)
    L.call("asr_conv3x3_igemm", L.ptr(x), L.ptr(wk), L.ptr(bias), L.ptr(mask_src), L.ptr(y), B, H, W, Cin, Cout,
           int(relu), L.dt(x), L.stream())
    return y


def conv3x3_relu_bits(x, wk, bias, Cout):
    """(y, bits): y = ReLU(conv3x3(x) + bias) and its ReLU mask at one bit per element (include/asr_hip.h: asr_relu_bits_bytes), for
    conv3x3_masked_by_bits on the way back.  None where the library has no such form (callers keep y as the mask)."""
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    nbytes = L.load().asr_relu_bits_bytes(B, H, W, Cout)
    if nbytes < 0:
        return None
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=x.dtype)
    bits = torch.empty((nbytes,), device=x.device, dtype=torch.uint8)
    rc = L.load().asr_conv3x3_igemm_bits(L.ptr(x), L.ptr(wk), L.ptr(bias), None, L.ptr(y), L.ptr(bits), B, H, W, Cin, Cout, 1, L.dt(x),
                                         L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_conv3x3_igemm_bits")
    return y, bits


def conv3x3_masked_by_bits(x, wk, bias, Cout, bits, relu=False):
    """conv3x3(x) (+ bias) zeroed where `bits` (conv3x3_relu_bits) is 0; None where the library has no such form."""
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=x.dtype)
    rc = L.load().asr_conv3x3_igemm_bits(L.ptr(x), L.ptr(wk), L.ptr(bias), L.ptr(bits), L.ptr(y), None, B, H, W, Cin, Cout, int(relu),
                                         L.dt(x), L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_conv3x3_igemm_bits")
    return y


def conv3x3_relu_pool(x, wk, bias, Cout):
    """(y, pool): y = ReLU(conv3x3(x) + bias) and its 2x2/2 max-pool.  One kernel where the library has the fused epilogue
    (bf16, 64 -> 64 channels), otherwise the convolution followed by the pooling kernel -- both are HIP paths."""
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=x.dtype)
    pool = torch.empty((B, H // 2, W // 2, Cout), device=x.device, dtype=x.dtype)
    rc = L.load().asr_conv3x3_relu_pool(L.ptr(x), L.ptr(wk), L.ptr(bias), L.ptr(y), L.ptr(pool), B, H, W, Cin, Cout, L.dt(x),
                                        L.stream())
    if rc == L.EUNSUPPORTED:
        y = conv3x3(x, wk, bias, Cout, relu=True)
        return y, maxpool_fwd(y)
    L.check(rc, "asr_conv3x3_relu_pool")
    return y, pool


def conv3x3_relu_pool_code(x, wk, bias, Cout, keep_y=False):
    """(y or None, pool, code): pool = 2x2/2 max-pool of ReLU(conv3x3(x) + bias), code = one selection byte per pooled element
    (maxpool_bwd_code routes the gradient with it); the un-pooled y is stored only on request.  None when the library has no fused
    form for this shape (callers use conv3x3_relu_pool)."""
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=x.dtype) if keep_y else None
    pool = torch.empty((B, H // 2, W // 2, Cout), device=x.device, dtype=x.dtype)
    code = torch.empty((B, H // 2, W // 2, Cout), device=x.device, dtype=torch.uint8)
    rc = L.load().asr_conv3x3_relu_pool_code(L.ptr(x), L.ptr(wk), L.ptr(bias), L.ptr(y), L.ptr(pool), L.ptr(code), B, H, W, Cin, Cout,
                                             L.dt(x), L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_conv3x3_relu_pool_code")
    return y, pool, code


def vgg_level0_fwd(src, w0, b0, wk2, b2):
    """(pool (B, H/2, W/2, 64) bf16, code uint8) = MaxPool(ReLU(conv.2(ReLU(conv.0(src))))) from ONE launch that never stores a
    full-resolution 64-channel tensor (csrc/conv_level0.hip); src (B, 1, H, W) or (B, H, W) fp32.  None when the library does not
    take the shape."""
    src3 = src.reshape(src.shape[0], src.shape[-2], src.shape[-1])
    assert src3.is_contiguous() and src3.dtype == torch.float32 and w0.shape[0] == 64
    B, H, W = src3.shape
    pool = torch.empty((B, H // 2, W // 2, 64), device=src.device, dtype=torch.bfloat16)
    code = torch.empty((B, H // 2, W // 2, 64), device=src.device, dtype=torch.uint8)
    rc = L.load().asr_vgg_level0_fwd(L.ptr(src3), L.ptr(w0), L.ptr(b0), L.ptr(wk2), L.ptr(b2), L.ptr(pool), L.ptr(code), B, H, W, L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_vgg_level0_fwd")
    return pool, code


def _level0_ws(B, H, W, device):
    n = L.load().asr_vgg_level0_bwd_workspace(B, H, W)
    return workspace("wgrad_ws", (max(int(n), 1),), torch.float32, device, zero=False), int(n)      # the conv weight gradients' shared scratch


def vgg_level0_dgrad(dpool, code, src, w0, b0, wd2, dw0, db0):
    """dw0 += , db0 += : conv.2's data gradient of the pooled gradient (expanded through the codes in the kernel), conv.0's ReLU mask
    recomputed, contracted against the frames -- no full-resolution gradient tensor exists."""
    src3 = src.reshape(src.shape[0], src.shape[-2], src.shape[-1])
    B, H, W = src3.shape
    assert dpool.is_contiguous() and code.is_contiguous() and tuple(dpool.shape) == (B, H // 2, W // 2, 64)
    ws, n = _level0_ws(B, H, W, src.device)
    L.call("asr_vgg_level0_dgrad", L.ptr(dpool), L.ptr(code), L.ptr(src3), L.ptr(w0), L.ptr(b0), L.ptr(wd2), L.ptr(dw0), L.ptr(db0),
           L.ptr(ws), n, B, H, W, L.stream())


def vgg_level0_wgrad(src, w0, b0, dpool, code, dw2, db2):
    """dw2 += , db2 += for conv.2 from ReLU(conv.0(src)) recomputed on halo patches and the expanded pooled gradient."""
    src3 = src.reshape(src.shape[0], src.shape[-2], src.shape[-1])
    B, H, W = src3.shape
    assert dpool.is_contiguous() and code.is_contiguous()
    ws, n = _level0_ws(B, H, W, src.device)
    L.call("asr_vgg_level0_wgrad", L.ptr(src3), L.ptr(w0), L.ptr(b0), L.ptr(dpool), L.ptr(code), L.ptr(dw2), L.ptr(db2), L.ptr(ws), n,
           B, H, W, L.stream())


def conv3x3_relu_pool_tcf_code(x, wk, bias, Cout, code_cl=False):
    """(pool (B, W/2, Cout * H/2), code): the encoder-layout max-pool of ReLU(conv3x3(x) + bias) and its selection bytes from the
    convolution's own epilogue (the un-pooled output is never stored); None when the library has no fused form for this shape.
    code_cl: the selection bytes CHANNEL LAST, (B, W/2, H/2, Cout) -- what gemm_nn_poolbwd reads -- instead of pool's own layout
    (B, W/2, Cout, H/2); None when only the other layout is available."""
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    pool = torch.empty((B, W // 2, Cout * (H // 2)), device=x.device, dtype=x.dtype)
    code = torch.empty((B, W // 2, H // 2, Cout) if code_cl else (B, W // 2, Cout * (H // 2)), device=x.device, dtype=torch.uint8)
    fn = L.load().asr_conv3x3_relu_pool_tcf_codecl if code_cl else L.load().asr_conv3x3_relu_pool_tcf_code
    rc = fn(L.ptr(x), L.ptr(wk), L.ptr(bias), L.ptr(pool), L.ptr(code), B, H, W, Cin, Cout, L.dt(x), L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_conv3x3_relu_pool_tcf_code")
    return pool, code


def permute_cols_tcf(src, dst, C, H2):
    """dst[r][h2 * C + c] = src[r][c * H2 + h2] (2-D, same shape): model feature order -> channel last (asr_permute_cols_tcf)."""
    assert src.dim() == 2 and dst.shape == src.shape and src.shape[1] >= C * H2 and src.dtype == dst.dtype
    L.call("asr_permute_cols_tcf", L.ptr(src), src.stride(0), L.ptr(dst), dst.stride(0), src.shape[0], C, H2, L.dt(src), L.stream())
    return dst


def gemm_nn_poolbwd(dy2d, w_perm, code_cl, x_shape):
    """The encoder input projection's data gradient with the second max-pool's backward in its epilogue: dy2d (B * W2, N) . w_perm
    (N, H2 * C; columns channel last) routed through the selection bytes code_cl (B, W2, H2, C) into the gradient of the un-pooled conv
    output, (B, 2 H2, 2 W2, C) NHWC = x_shape.  None when the library has no such form for the shape (callers: gemm_nn + maxpool_bwd_code)."""
    B, H, W, C = x_shape
    H2, W2 = H // 2, W // 2
    M, K = dy2d.shape[0], w_perm.shape[0]
    if (H % 2 or W % 2 or M != B * W2 or w_perm.shape[1] != H2 * C or dy2d.dtype != torch.bfloat16 or dy2d.stride(1) != 1 or
            w_perm.stride(1) != 1 or tuple(code_cl.shape) != (B, W2, H2, C) or not code_cl.is_contiguous()):
        return None
    dx = torch.empty(x_shape, device=dy2d.device, dtype=dy2d.dtype)
    rc = L.load().asr_gemm_nn_poolbwd(L.ptr(dy2d), dy2d.stride(0), L.ptr(w_perm), w_perm.stride(0), L.ptr(code_cl), L.ptr(dx), M, K, H2, W2,
                                      C, L.dt(dy2d), L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_gemm_nn_poolbwd")
    return dx


def maxpool_fwd_code(x, tcf=False):
    """(y, code) or None when the layout has no 16-byte form."""
    B, H, W, C = x.shape
    shape = (B, W // 2, C * (H // 2)) if tcf else (B, H // 2, W // 2, C)
    y = torch.empty(shape, device=x.device, dtype=x.dtype)
    code = torch.empty(shape, device=x.device, dtype=torch.uint8)
    rc = L.load().asr_maxpool_fwd_code(L.ptr(x), L.ptr(y), L.ptr(code), B, H, W, C, int(tcf), L.dt(x), L.stream())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, "asr_maxpool_fwd_code")
    return y, code


def maxpool_bwd_code(code, dy, x_shape, tcf=False):
    """dx (x_shape = (B, H, W, C)) from the pooled gradient and the forward's selection codes."""
    B, H, W, C = x_shape
    assert dy.is_contiguous() and code.is_contiguous() and code.dtype == torch.uint8
    dx = torch.empty(x_shape, device=dy.device, dtype=dy.dtype)
    L.call("asr_maxpool_bwd_code", L.ptr(code), L.ptr(dy), L.ptr(dx), B, H, W, C, int(tcf), L.dt(dy), L.stream())
    return dx


def maxpool_fwd(x, tcf=False):
    B, H, W, C = x.shape
    if tcf:
        y = torch.empty((B, W // 2, C * (H // 2)), device=x.device, dtype=x.dtype)
    else:
        y = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
    L.call("asr_maxpool_fwd", L.ptr(x), L.ptr(y), B, H, W, C, int(tcf), L.dt(x), L.stream())
    return y


def maxpool_bwd(x, dy, tcf=False):
    B, H, W, C = x.shape
    assert dy.is_contiguous()
    dx = torch.empty_like(x)
    L.call("asr_maxpool_bwd", L.ptr(x), L.ptr(dy), L.ptr(dx), B, H, W, C, int(tcf), L.dt(x), L.stream())
    return dx


def conv3x3_wgrad_nhwc(x, dy, dw, db=None):
    """dW (Cout,Cin,3,3) += , db (Cout) += from NHWC activations x (B,H,W,Cin) and gradients dy (B,H,W,Cout)."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape[:3] == x.shape[:3]
    n_ws = L.load().asr_conv3x3_wgrad_workspace(B, H, W, Cin, Cout)
    ws = workspace("wgrad_ws", (n_ws,), torch.float32, x.device, zero=False)      # 75 MB, shared by the three conv layers
    L.call("asr_conv3x3_wgrad_nhwc", L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(ws), n_ws, B, H, W, Cin, Cout, L.dt(x),
           L.stream())



# ------------------------------------------------------------------------------------------------ emb_cnn front end
_ws = {}


def workspace(tag, shape, dtype, device, zero=True, geom=None):
    """Persistent zero-initialised buffer: kernels rewrite the live region only, padding rows/columns stay zero.
    ONE grow-only allocation per tag (variable-length batches do not leak a buffer per shape): a request that fits is a view
    of it, re-zeroed only when the shape -- or `geom`, whatever else decides WHERE the live region lies inside an equal shape --
    differs from the previous request (the padding moves).  zero=False: scratch that its user overwrites completely (never
    re-zeroed: three layers sharing one tag cost three 75 MB fills per step otherwise)."""
    key = (tag, dtype, str(device))
    shape = tuple(int(x) for x in shape)
    n = 1
    for x in shape:
        n *= x
    ent = _ws.get(key)
    if ent is None or ent[0].numel() < n:
        ent = [torch.zeros(n, device=device, dtype=dtype), (shape, geom)]
        _ws[key] = ent
    elif ent[1] != (shape, geom):
        if zero:
            ent[0][:n].zero_()
        ent[1] = (shape, geom)
    return ent[0][:n].view(shape)


def conv_geom(B, H, W, C, KH, KW, SH, SW, PH, PW):
    """(B,H,W,C,KH,KW,SH,SW,PH,PW,OH,OW) of a strided 2-D convolution on an NHWC tensor."""
    return (B, H, W, C, KH, KW, SH, SW, PH, PW, (H + 2 * PH - KH) // SH + 1, (W + 2 * PW - KW) // SW + 1)


def im2col(x, g, col):
    """col (rows_alloc, ld) <- patches of NHWC x; row (b,oh,ow), column (ky,kx,c); padding written as zeros."""
    assert x.is_contiguous() and col.is_contiguous() and x.numel() == g[0] * g[1] * g[2] * g[3]
    L.call("asr_im2col", L.ptr(x), L.ptr(col), *g, col.stride(0), col.shape[0], L.dt(x), L.dt(col), L.stream())
    return col


def col2im(dcol, g, dtype=None):
    """dx (B,H,W,C) <- gather of dcol (B*OH*OW, ld): the data gradient of the convolution described by g."""
    B, H, W, C = g[:4]
    dx = torch.empty((B, H, W, C), device=dcol.device, dtype=dcol.dtype)
    L.call("asr_col2im", L.ptr(dcol), L.ptr(dx), *g, dcol.stride(0), L.dt(dcol), L.stream())
    return dx


def window_sum(Z, y, bias, groups, Wg, OW, KW, Cout):
    """y[g*OW + j, co] = bias[co] + sum_kx Z[g*Wg + j + kx, kx*Cout + co] for j < OW (asr_window_sum): the per-tap partial products
    of the unit-stride window convolution folded along time; columns >= Cout of y are zeroed."""
    assert Z.dtype == torch.float32 and y.dtype == torch.float32 and Z.stride(1) == 1 and y.stride(1) == 1
    assert Z.shape[0] >= groups * Wg + KW - 1 - (Wg - OW) and y.shape[0] >= groups * OW
    L.call("asr_window_sum", L.ptr(Z), Z.stride(0), L.ptr(y), y.stride(0), L.ptr(bias) if bias is not None else None, groups, Wg, OW,
           KW, Cout, L.stream())
    return y


def bn_batch_stats(y, M, C, ygrid=(0, 0)):
    """Training-mode BatchNorm statistics of the fp32 conv output y (rows, ld) over its first M rows / C columns:
    (mean, biased var), two passes (mean, then centred second moment).  ygrid = (Wg, OW): y is the un-compacted output of a
    window GEMM (groups of Wg rows, the first OW of each are the convolution's rows; include/asr_hip.h)."""
    nb = L.load().asr_bn_stats_blocks(M)
    part = torch.empty((nb, 2 * C), device=y.device, dtype=torch.float32)       # per-workgroup sums, added in a fixed order:
    s1 = torch.empty(2 * C, device=y.device, dtype=torch.float32)               # reproducible statistics (no atomics)
    L.call("asr_bn_stats_partial", L.ptr(y), y.stride(0), M, C, None, L.ptr(part), L.ptr(s1), ygrid[0], ygrid[1], L.stream())
    mean = s1[:C] / M
    s2 = torch.empty(2 * C, device=y.device, dtype=torch.float32)
    L.call("asr_bn_stats_partial", L.ptr(y), y.stride(0), M, C, L.ptr(mean), L.ptr(part), L.ptr(s2), ygrid[0], ygrid[1], L.stream())
    return mean, s2[C:] / M


def bn_train_stats(y, M, C, eps, momentum=-1.0, running_mean=None, running_var=None, num_batches=None, ygrid=(0, 0), valid=None):
    """-> (mean, rstd) of training-mode BatchNorm over y's first M rows / C columns, the module's running buffers updated in place
    (momentum < 0: left alone): asr_bn_batch_stats, four launches.  valid = (device int32[1], row_w): only rows with
    m % row_w < valid[0] count (the padding a shape bucket adds behind the collated batch stays out of the statistics)."""
    nb = L.load().asr_bn_stats_blocks(M)
    part = torch.empty((nb, 2 * C), device=y.device, dtype=torch.float32)
    mean = torch.empty(C, device=y.device, dtype=torch.float32)
    rstd = torch.empty(C, device=y.device, dtype=torch.float32)
    L.call("asr_bn_batch_stats_v", L.ptr(y), y.stride(0), M, C, L.ptr(part), L.ptr(mean), L.ptr(rstd), float(eps), float(momentum),
           L.ptr(running_mean), L.ptr(running_var), L.ptr(num_batches), ygrid[0], ygrid[1], L.ptr(valid[0]) if valid else None,
           int(valid[1]) if valid else 0, L.stream())
    return mean, rstd


def bn_act_fwd(y, M, C, mean, rstd, gamma, beta, lo, hi, out, tH=0, tW=0, ygrid=(0, 0)):
    ldo = 0 if tH else out.stride(0)
    L.call("asr_bn_act_fwd", L.ptr(y), y.stride(0), L.ptr(out), ldo, M, C, L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta),
           float(lo), float(hi), tH, tW, ygrid[0], ygrid[1], L.dt(out), L.stream())
    return out


def bn_act_bwd(dout, y, M, C, mean, rstd, gamma, beta, lo, hi, dy, tH=0, tW=0, ygrid=(0, 0), dygrid=(0, 0), valid=None):
    """-> sums (2C): [dbeta, dgamma]; writes dy[:M, :C] (the gradient w.r.t. the conv output) in dy's dtype.  dygrid = (Wg, OW): row m
    goes to dy row (m // OW) * Wg + m % OW (dy is the dense operand of the window gradients; the rows in between are left alone)."""
    assert dout.dtype == dy.dtype and dout.is_contiguous()
    ldo = 0 if tH else dout.stride(0)
    sums = torch.zeros(2 * C, device=y.device, dtype=torch.float32)
    vp, vw = (L.ptr(valid[0]), int(valid[1])) if valid else (None, 0)
    L.call("asr_bn_act_bwd_reduce_v", L.ptr(dout), ldo, L.ptr(y), y.stride(0), M, C, L.ptr(mean), L.ptr(rstd), L.ptr(gamma),
           L.ptr(beta), float(lo), float(hi), tH, tW, ygrid[0], ygrid[1], L.ptr(sums), vp, vw, L.dt(dout), L.stream())
    L.call("asr_bn_act_bwd_v", L.ptr(dout), ldo, L.ptr(y), y.stride(0), L.ptr(dy), dy.stride(0), M, C, L.ptr(mean), L.ptr(rstd),
           L.ptr(gamma), L.ptr(beta), float(lo), float(hi), tH, tW, ygrid[0], ygrid[1], dygrid[0], dygrid[1], L.ptr(sums), vp, vw,
           L.dt(dout), L.stream())
    return sums


# ------------------------------------------------------------------------------------------------ spectrogram front end
_stft_const = {}


def _stft_constants(n_fft, device):
    """Symmetric Hamming window (n_fft) and the [cos | -sin] DFT basis (2*(n_fft/2+1), n_fft), fp32, computed in float64."""
    key = (n_fft, str(device))
    c = _stft_const.get(key)
    if c is None:
        k = torch.arange(n_fft, dtype=torch.float64)
        win = (0.54 - 0.46 * torch.cos(2.0 * math.pi * k / (n_fft - 1))).float()
        f = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
        ang = 2.0 * math.pi * f * k[None, :] / n_fft
        basis = torch.cat([torch.cos(ang), -torch.sin(ang)], dim=0).float()
        c = (win.to(device), basis.to(device).contiguous())
        _stft_const[key] = c
    return c


def log_spectrogram(wav, lengths, n_fft=320, hop=160, normalize=True):
    """Padded waveforms wav (B, L) fp32 + lengths (B) int32 (samples), both on the device -> (spect (B, 1, n_fft/2+1, Tmax)
    fp32 zero padded along T, n_frames (B) int32): log1p(|STFT|) normalised per utterance, the reference loader's features
    (utils/data_loader.py:72-89) computed on the GPU: framing kernel -> fp32 MFMA GEMM against the DFT basis -> magnitude /
    log1p / mean / unbiased std kernels."""
    assert wav.dim() == 2 and wav.dtype == torch.float32 and wav.stride(1) == 1 and lengths.dtype == torch.int32
    B, Lmax = wav.shape
    F = n_fft // 2 + 1
    Tmax = 1 + max(Lmax, 2) // hop
    win, basis = _stft_constants(n_fft, wav.device)
    frames = torch.empty((B * Tmax, n_fft), device=wav.device, dtype=torch.float32)
    L.call("asr_stft_frames", L.ptr(wav), wav.stride(0), L.ptr(lengths), L.ptr(win), L.ptr(frames), B, Tmax, n_fft, hop,
           L.stream())
    ld = (2 * F + 3) // 4 * 4
    reim = torch.empty((B * Tmax, ld), device=wav.device, dtype=torch.float32)
    gemm_nt(frames, basis, out=reim[:, :2 * F])
    spect = torch.empty((B, 1, F, Tmax), device=wav.device, dtype=torch.float32)
    scratch = torch.zeros((2, B), device=wav.device, dtype=torch.float32)
    L.call("asr_spect_finish", L.ptr(reim), ld, L.ptr(lengths), L.ptr(spect), L.ptr(scratch[0]), L.ptr(scratch[1]), B, F, Tmax,
           hop, int(normalize), L.stream())
    n_frames = 1 + torch.clamp(lengths, min=2) // hop
    return spect, n_frames.to(torch.int32)


# ------------------------------------------------------------------------------------------------ profiling
def prof_enable(op, on=True):
    L.call("asr_prof_enable", int(op), int(on))


def prof_collect(op):
    import ctypes
    ms = ctypes.c_double(0.0)
    n = ctypes.c_int64(0)
    L.check(L.load().asr_prof_collect(int(op), ctypes.byref(ms), ctypes.byref(n)), "asr_prof_collect")
    return ms.value, n.value
