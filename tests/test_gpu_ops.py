"""Per-kernel parity: every C-ABI entry point against the torch-CPU fp32 expression it replaces (seeded inputs).
Tolerances: fp32 mode = fp32 round-off of a different summation order; bf16 mode = bf16 storage (8 mantissa bits) of
inputs/outputs with fp32 accumulation.  Runs on the MI355X (`-m gpu`)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def dev():
    return torch.device("cuda:0")


def tol(dtype, scale=1.0):
    return (2e-5 if dtype == torch.float32 else 1.6e-2) * scale


def close(name, got, ref, dtype, scale=1.0, atol=0.0):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), "%s: non-finite output" % name
    err = (got - ref).abs().max().item()
    bound = atol + tol(dtype, scale) * max(ref.abs().max().item(), 1e-6)
    assert err <= bound, "%s: max abs err %.3e > %.3e (ref max %.3e)" % (name, err, bound, ref.abs().max().item())


def q(x, dtype):
    """Round a CPU fp32 tensor to what the kernel will actually see."""
    return x.to(dtype).float()


@pytest.fixture(scope="module")
def ops():
    from asr_hip import ops as o
    return o


# ------------------------------------------------------------------------------------------------ MFMA layout probe
def test_mfma_fragment_layout_asymmetric(ops):
    """A = I (and a permutation) against an ASYMMETRIC B catches row/col swaps in the fragment maps."""
    for dtype, n in [(d, n) for d in DTYPES for n in (48, 128, 192)]:      # 48: generic kernel; 128/192: LDS-DMA kernel
        A = torch.eye(n)
        B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 / 16.0       # exactly representable in bf16
        C = ops.gemm_nt(A.to(dev(), dtype), B.to(dev(), dtype), out_dtype=torch.float32)
        assert torch.equal(C.cpu(), B.t().contiguous()), "gemm_nt(I, B) != B^T for %s" % dtype
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
        C2 = ops.gemm_nt(A[perm].to(dev(), dtype), B.to(dev(), dtype), out_dtype=torch.float32)
        assert torch.equal(C2.cpu(), B.t()[perm].contiguous())


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(200, 70, 96), (512, 384, 256), (130, 260, 40), (64, 4364, 512), (1000, 512, 5120),
                                   (37, 35, 161), (200, 70, 128), (129, 257, 192), (6400, 512, 512), (333, 2048, 64)])
def test_gemm_nt_bias_relu(ops, dtype, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = q(torch.randn(M, K, generator=g), dtype)
    B = q(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
    bias = torch.randn(N, generator=g)
    ref = A @ B.t() + bias
    out = ops.gemm_nt(A.to(dev(), dtype), B.to(dev(), dtype), bias=bias.to(dev()))
    close("gemm", out, ref, dtype)
    out = ops.gemm_nt(A.to(dev(), dtype), B.to(dev(), dtype), bias=bias.to(dev()), relu=True, out_dtype=torch.float32)
    close("gemm+relu fp32 out", out, ref.relu(), torch.float32 if dtype == torch.float32 else dtype, scale=0.1 if dtype != torch.float32 else 1)


@pytest.mark.parametrize("block,ns", [(256, 2), (128, 2), (128, 3), (128, 4)])
def test_gemm_nt_eight_wave_blocks(ops, block, ns):
    """csrc/gemm_big.hip (512-thread workgroups, 256 x 256 / 128 x 128 blocks, LDS-DMA ring) forced on shapes that exercise every
    edge: rows not a multiple of the block (3200 = 12.5 x 256, 777), N not a multiple of it (300 -> one partial block column, 4364),
    K = 64 (one step) .. 2048, a row stride larger than K, bias / ReLU / alpha, fp32 output, accumulate.  Reference: fp32 torch on
    the same bf16 operands; tolerance 2 bf16 ulp of the result (fp32 accumulation, another summation order)."""
    from asr_hip import lib as L
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(block + ns)
    L.set_tuning("GEMM_BIG", block)
    L.set_tuning("GEMM_BIG_NS", ns)
    try:
        for M, N, K, ldk, kind in [(3200, 512, 512, 512, "bias_relu"), (777, 304, 64, 64, "bias"), (1000, 1536, 2048, 2048, "plain"),
                                   (513, 4364, 512, 576, "f32"), (640, 512, 192, 192, "alpha_acc"), (256, 256, 128, 128, "f32acc")]:
            A = torch.randn(M, ldk, generator=g).to(dev()).to(bf)[:, :K]
            B = (torch.randn(N, ldk, generator=g) * K ** -0.5).to(dev()).to(bf)[:, :K]
            ref = A.float() @ B.float().t()
            if kind == "bias_relu":
                bias = torch.randn(N, generator=g).to(dev())
                out = ops.gemm_nt(A, B, bias=bias, relu=True)
                ref = (ref + bias).relu()
            elif kind == "bias":
                bias = torch.randn(N, generator=g).to(dev())
                out = ops.gemm_nt(A, B, bias=bias)
                ref = ref + bias
            elif kind == "plain":
                out = ops.gemm_nt(A, B)
            elif kind == "f32":
                out = ops.gemm_nt(A, B, out_dtype=torch.float32)
            elif kind == "alpha_acc":
                c0 = torch.randn(M, N, generator=g).to(dev()).to(bf)
                out = c0.clone()
                ops.gemm_nt(A, B, out=out, accumulate=True, alpha=0.5)
                ref = c0.float() + 0.5 * ref
            else:
                c0 = torch.randn(M, N, generator=g).to(dev())
                out = c0.clone()
                ops.gemm_nt(A, B, out=out, accumulate=True)
                ref = c0 + ref
            tol = (2.0 ** -7 if out.dtype == bf else 2e-5) * ref.abs().clamp_min(0.05 if out.dtype == bf else 1.0)
            bad = (out.float() - ref).abs() > tol
            assert not bad.any(), (M, N, K, kind, (out.float() - ref).abs().max().item(), int(bad.sum()))
    finally:
        L.set_tuning("GEMM_BIG", None)
        L.set_tuning("GEMM_BIG_NS", None)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_splitk_accumulate_and_mask(ops, dtype):
    g = torch.Generator().manual_seed(7)
    M, N, K = 96, 160, 4096
    A = q(torch.randn(M, K, generator=g), dtype)
    B = q(torch.randn(N, K, generator=g) / 64, dtype)
    C0 = torch.randn(M, N, generator=g)
    out = C0.clone().to(dev())
    ops.gemm_nt(A.to(dev(), dtype), B.to(dev(), dtype), out=out, accumulate=True, splits=8, alpha=0.5)
    close("split-K", out, C0 + 0.5 * (A @ B.t()), torch.float32 if dtype == torch.float32 else dtype, scale=0.2 if dtype != torch.float32 else 4)
    mask = q(torch.randn(M, N, generator=g), dtype)
    out2 = ops.gemm_nt(A.to(dev(), dtype), B.to(dev(), dtype), relu_mask=mask.to(dev(), dtype))
    close("relu-mask epilogue", out2, (A @ B.t()) * (mask > 0), dtype)
    acc = q(torch.randn(M, N, generator=g), dtype).to(dev(), dtype)
    ref = acc.float().cpu() + A @ B.t()
    ops.gemm_nt(A.to(dev(), dtype), B.to(dev(), dtype), out=acc, accumulate=True)
    close("accumulate in storage dtype", acc, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(256, 64, 64), (384, 200, 70), (1280, 512, 2048), (128, 35, 161), (100, 64, 64),
                                   (12720, 512, 512), (1, 8, 8), (333, 96, 40), (700, 1024, 2056)])
def test_gemm_tn_weight_gradient(ops, dtype, shape):
    """dW += dY^T X and db += colsum(dY) straight from the natural layouts (transposing LDS reads)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    ldy, ldx = (N + 63) // 64 * 64, (K + 7) // 8 * 8
    dy = torch.zeros(M, ldy); dy[:, :N] = q(torch.randn(M, N, generator=g), dtype)
    x = torch.zeros(M, ldx); x[:, :K] = q(torch.randn(M, K, generator=g), dtype)
    dw0 = torch.randn(N, K, generator=g)
    db0 = torch.randn(N, generator=g)
    dw, db = dw0.clone().to(dev()), db0.clone().to(dev())
    dyd, xd = dy.to(dev(), dtype), x.to(dev(), dtype)
    assert ops.gemm_tn_supported(dyd, xd)
    ops.gemm_tn(dyd, xd, dw, colsum_acc=db, N=N, K=K)
    sc = 8 if dtype == torch.float32 else 0.2
    close("tn dW", dw, dw0 + dy[:, :N].t() @ x[:, :K], torch.float32 if dtype == torch.float32 else dtype, scale=sc)
    close("tn db", db, db0 + dy[:, :N].sum(0), torch.float32, scale=8)
    dw1 = dw0.clone().to(dev())
    ops.gemm_tn(dyd, xd, dw1, N=N, K=K, splits=1)
    close("tn dW (no split)", dw1, dw0 + dy[:, :N].t() @ x[:, :K], torch.float32 if dtype == torch.float32 else dtype, scale=sc)
    assert not ops.gemm_tn_supported(dyd[:, 1:], xd)          # misaligned rows: callers fall back to explicit transposes


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(200, 64, 64), (130, 192, 72), (640, 512, 2048), (96, 1536, 512)])
def test_gemm_nn_data_gradient(ops, dtype, shape):
    """dx = dy @ W with W in its natural (N,K) layout; accumulate and ReLU-mask epilogues."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + K)
    dy = q(torch.randn(M, N, generator=g), dtype)
    w = q(torch.randn(N, K, generator=g) / math.sqrt(N), dtype)
    D = dev()
    assert ops.gemm_nn_supported(dy.to(D, dtype), w.to(D, dtype))
    out = ops.gemm_nn(dy.to(D, dtype), w.to(D, dtype))
    close("nn", out, dy @ w, dtype)
    base = q(torch.randn(M, K, generator=g), dtype)
    acc = base.to(D, dtype)
    ops.gemm_nn(dy.to(D, dtype), w.to(D, dtype), out=acc, accumulate=True)
    close("nn accumulate", acc, base + dy @ w, dtype)
    mask = q(torch.randn(M, K, generator=g), dtype)
    out = ops.gemm_nn(dy.to(D, dtype), w.to(D, dtype), relu_mask=mask.to(D, dtype))
    close("nn relu mask", out, (dy @ w) * (mask > 0), dtype)


@pytest.mark.parametrize("M,T,N,K,big", [(6400, 200, 512, 512, 1), (3200, 100, 512, 512, 0), (12720, 795, 512, 512, 1), (1590, 795, 512, 512, 0),
                                         (600, 100, 128, 192, 0), (600, 100, 128, 192, 2)])
@pytest.mark.parametrize("use_o32", [True, False])
def test_gemm_nn_rowdot_is_the_attention_delta(ops, M, T, N, K, big, use_o32):
    """asr_gemm_nn_rowdot: the output projection's data gradient (dO of the attention backward) and, from the same epilogue, delta =
    rowsum(dO * O) per head (reference: autograd of models/common_layers.py:211-225's softmax(QK^T)V -- the softmax backward's row
    term).  dO must be asr_gemm_nn's bits; delta must be what asr_attn_bwd's own delta launch computes from that dO (compared through
    the attention backward itself: the same dQ / dK / dV bits -- both GEMM kernels add in the delta kernel's lanes and order) and the
    fp64 row sums of the ROUNDED dO within fp32 summation error."""
    from asr_hip import lib as L
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + N)
    D = dev()
    H, B = N // 64, M // T
    dy = torch.randn(M, K, generator=g).to(D).to(bf)
    w = (torch.randn(K, N, generator=g) * K ** -0.5).to(D).to(bf)
    o32 = torch.randn(M, N, generator=g).to(D)
    o = o32.to(bf)
    L.set_tuning("GEMM_BIG_NN", big if big != 1 else None)
    try:
        want_dx = ops.gemm_nn(dy, w)
        got = ops.gemm_nn_rowdot(dy, w, o, o32 if use_o32 else None, T)
    finally:
        L.set_tuning("GEMM_BIG_NN", None)
    assert got is not None
    dx, delta = got
    assert torch.equal(dx, want_dx)
    ref = (dx.double() * (o32 if use_o32 else o).double()).view(B, T, H, 64).sum(-1).permute(0, 2, 1)      # (B, H, T)
    scale = (dx.double().abs() * (o32 if use_o32 else o).double().abs()).view(B, T, H, 64).sum(-1).permute(0, 2, 1)
    assert delta.shape == (B, H, T)
    assert ((delta.double() - ref).abs() <= 8e-6 * scale + 1e-30).all()           # 64 products and their fp32 sum
    # through the attention backward: delta handed in against delta computed by asr_attn_bwd's own launch
    q_, k_, v_ = (torch.randn(B, T, N, generator=g).to(D).to(bf) for _ in range(3))
    lse = torch.randn(B, H, T, generator=g).to(D).abs() + 6.0
    a = ops.attn_bwd(q_, k_, v_, o.view(B, T, N), dx.view(B, T, N), lse, H, 64, scale=0.125, o32=o32.view(B, T, N) if use_o32 else None)
    b = ops.attn_bwd(q_, k_, v_, o.view(B, T, N), dx.view(B, T, N), lse, H, 64, scale=0.125, o32=o32.view(B, T, N) if use_o32 else None,
                     delta=delta)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.fixture
def four_wave_nn():
    """asr_gemm_nn on the four-wave kernel only (GEMM_BIG_NN = 0) for the duration of a test."""
    from asr_hip import lib as L
    L.set_tuning("GEMM_BIG_NN", 0)
    yield
    L.set_tuning("GEMM_BIG_NN", None)


@pytest.mark.parametrize("ns", [2, 3, 4])
def test_gemm_nn_eight_wave_blocks(ops, ns):
    """csrc/gemm_big.hip gemm_big_nn_kernel (128 x 128 blocks, the weight read with transposing LDS reads) forced on shapes with
    ragged rows (777), an output width that is not a multiple of the block (200: partial block column; 264), contractions of 64 ..
    2048 and a weight row stride larger than its width; plain, `+=` into bf16 (one rounding) and ReLU-mask epilogues, alpha.
    Reference: fp32 torch on the same bf16 operands, 2 bf16 ulp."""
    from asr_hip import lib as L
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(ns)
    D = dev()
    L.set_tuning("GEMM_BIG_NN", 2)
    L.set_tuning("GEMM_BIG_NS", ns)
    try:
        for M, N, K, ldw in [(6400, 512, 512, 512), (777, 200, 64, 200), (1000, 2048, 512, 2048), (3200, 512, 2048, 512), (130, 264, 192, 320)]:
            # out (M, N) = dy (M, K) @ w (K, N): K is the contracted width here (ops.gemm_nn names it N)
            dy = torch.randn(M, K, generator=g).to(D).to(bf)
            w = (torch.randn(K, ldw, generator=g) * K ** -0.5).to(D).to(bf)[:, :N]
            ref = dy.float() @ w.float()
            out = ops.gemm_nn(dy, w)
            base = torch.randn(M, N, generator=g).to(D).to(bf)
            acc = base.clone()
            ops.gemm_nn(dy, w, out=acc, accumulate=True, alpha=0.5)
            mask = torch.randn(M, N, generator=g).to(D).to(bf)
            msk = ops.gemm_nn(dy, w, relu_mask=mask)
            for name, got, want in (("plain", out, ref), ("accumulate", acc, base.float() + 0.5 * ref), ("mask", msk, ref * (mask.float() > 0))):
                tol = 2.0 ** -7 * want.abs().clamp_min(0.05)
                bad = (got.float() - want).abs() > tol
                assert not bad.any(), (M, N, K, name, (got.float() - want).abs().max().item(), int(bad.sum()))
    finally:
        L.set_tuning("GEMM_BIG_NN", None)
        L.set_tuning("GEMM_BIG_NS", None)


def test_gemm_nt_private_ring_equals_single_stage(ops):
    """The four-wave forward GEMM (gemm_glds_kernel, 64 x 64 blocks) in its three-stage ring form -- taken by launches of at most NT_RING
    blocks, e.g. the decoder's 3200 x 512 projection over K = 2048 -- against the single-stage form (NT_RING = 0): the same MFMA sequence,
    bit-identical; bias + ReLU, fp32 output, `+=` into fp32, ReLU-mask epilogues, ragged rows and a partial block column; and against
    fp32 torch.  GEMM_BIG = 0 keeps the eight-wave blocks out of the way."""
    from asr_hip import lib as L
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(9)
    D = dev()
    L.set_tuning("GEMM_BIG", 0)
    try:
        for M, N, K in [(3200, 512, 2048), (1600, 64, 512), (777, 200, 256), (130, 264, 320), (64, 64, 4096), (3200, 512, 192)]:
            A = torch.randn(M, K, generator=g).to(D).to(bf)
            W = (torch.randn(N, K, generator=g) * K ** -0.5).to(D).to(bf)
            bias = torch.randn(N, generator=g).to(D)
            base = torch.randn(M, N, generator=g).to(D)
            mask = torch.randn(M, N, generator=g).to(D).to(bf)
            ref = A.float() @ W.float().t()
            got = {}
            for ring in (0, None):
                L.set_tuning("NT_RING", ring)
                try:
                    acc = base.clone()
                    ops.gemm_nt(A, W, out=acc, accumulate=True, alpha=0.5)
                    got[ring] = (ops.gemm_nt(A, W, bias=bias, relu=True), ops.gemm_nt(A, W, out_dtype=torch.float32), acc,
                                 ops.gemm_nt(A, W, relu_mask=mask))
                finally:
                    L.set_tuning("NT_RING", None)
            for a, b in zip(got[0], got[None]):
                assert torch.equal(a, b), (M, N, K)
            for name, out, want in (("bias+relu", got[None][0], (ref + bias).relu()), ("fp32", got[None][1], ref),
                                    ("accumulate", got[None][2], base + 0.5 * ref), ("mask", got[None][3], ref * (mask.float() > 0))):
                tol = 2.0 ** -7 * want.abs().clamp_min(0.05)
                bad = (out.float() - want).abs() > tol
                assert not bad.any(), (M, N, K, name, (out.float() - want).abs().max().item(), int(bad.sum()))
    finally:
        L.set_tuning("GEMM_BIG", None)


def test_gemm_nn_private_ring_equals_single_stage(ops, four_wave_nn):
    """The four-wave data-gradient kernel in its three-stage LDS-DMA ring form (launches of at most NN_RING blocks: the decoder's
    1600-row gradients) against its single-stage form (NN_RING = 0): the same MFMA sequence, so bit-identical -- plain, `+=` and
    ReLU-mask epilogues, ragged rows, a partial block column, a partial last K stage (4364 -> zero columns up to 4416) -- and
    against fp32 torch on the same operands."""
    from asr_hip import lib as L
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    D = dev()
    for M, N, K, ldw in [(1600, 512, 2048, 512), (1600, 512, 1536, 512), (1601, 512, 512, 512), (777, 200, 256, 200), (1600, 512, 4364, 512),
                         (130, 264, 320, 320), (64, 64, 4096, 64)]:
        Kp = (K + 63) // 64 * 64
        dy = torch.zeros(M, Kp)
        dy[:, :K] = torch.randn(M, K, generator=g)
        dy = dy.to(D).to(bf)
        w = (torch.randn(K, ldw, generator=g) * K ** -0.5).to(D).to(bf)[:, :N]
        base = torch.randn(M, N, generator=g).to(D).to(bf)
        mask = torch.randn(M, N, generator=g).to(D).to(bf)
        ref = dy[:, :K].float() @ w.float()
        got = {}
        for ring in (0, None):
            L.set_tuning("NN_RING", ring)
            try:
                acc = base.clone()
                ops.gemm_nn(dy, w, out=acc, accumulate=True, alpha=0.5)      # (the contracted width is w's row count: dy may be wider)
                got[ring] = (ops.gemm_nn(dy, w), acc, ops.gemm_nn(dy, w, relu_mask=mask))
            finally:
                L.set_tuning("NN_RING", None)
        for a, b in zip(got[0], got[None]):
            assert torch.equal(a, b), (M, N, K)
        for name, out, want in (("plain", got[None][0], ref), ("accumulate", got[None][1], base.float() + 0.5 * ref),
                                ("mask", got[None][2], ref * (mask.float() > 0))):
            tol = 2.0 ** -7 * want.abs().clamp_min(0.05)
            bad = (out.float() - want).abs() > tol
            assert not bad.any(), (M, N, K, name, (out.float() - want).abs().max().item(), int(bad.sum()))


@pytest.mark.parametrize("shape", [(6400, 512, 512), (3200, 512, 2048), (3200, 2048, 512), (200, 64, 64), (130, 192, 72),
                                   (37, 64, 200), (1000, 1536, 512)])
def test_gemm_nn_tn_one_launch(ops, shape, four_wave_nn):
    """A linear layer's dX and dW from ONE launch (asr_gemm_nn_tn) + the multi-layer fold (asr_tn_reduce_multi): dX bit-identical
    to the four-wave asr_gemm_nn (the same workgroup code), dW / db against fp32 torch and against asr_gemm_tn."""
    M, N, K = shape
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    D = dev()
    ldx = (K + 7) // 8 * 8
    dy = q(torch.randn(M, N, generator=g), dtype)
    w = q(torch.randn(N, K, generator=g) / math.sqrt(N), dtype)
    x = torch.zeros(M, ldx); x[:, :K] = q(torch.randn(M, K, generator=g), dtype)
    dw0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    dyd, wd, xd = dy.to(D, dtype), w.to(D, dtype), x.to(D, dtype)
    ref_dw = dw0 + dy.t() @ x[:, :K]
    # two "layers" pending at once: plain store, and accumulate + ReLU mask
    dw_a, db_a = dw0.clone().to(D), db0.clone().to(D)
    out_a = ops.gemm_nn_tn(dyd, wd, xd, dw_a, db_a)
    base, mask = q(torch.randn(M, K, generator=g), dtype), q(torch.randn(M, K, generator=g), dtype)
    dw_b = dw0.clone().to(D)
    out_b = base.to(D, dtype)
    ops.gemm_nn_tn(dyd, wd, xd, dw_b, None, out=out_b, accumulate=True, relu_mask=mask.to(D, dtype))
    assert torch.equal(dw_b.cpu(), dw0)                       # the second layer's slices are still pending ...
    assert len(ops._tn_pending) == 1                           # ... the first layer's were folded by the second launch
    ops.flush_tn_reduces()
    assert torch.equal(out_a, ops.gemm_nn(dyd, wd))
    chk = base.to(D, dtype)
    ops.gemm_nn(dyd, wd, out=chk, accumulate=True, relu_mask=mask.to(D, dtype))
    assert torch.equal(out_b, chk)
    close("nn_tn dW", dw_a, ref_dw, dtype, scale=0.2)
    assert torch.equal(dw_a, dw_b)                            # fixed summation order: reproducible
    close("nn_tn db", db_a, db0 + dy.sum(0), torch.float32, scale=8)
    dw_c = dw0.clone().to(D)
    ops.gemm_tn(dyd, xd, dw_c, N=N, K=K)
    close("nn_tn dW vs gemm_tn", dw_a, dw_c.cpu(), torch.float32, scale=64)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(300, 4364, 512), (130, 100, 72), (64, 32, 512)])
def test_gemm_nn_partial_reduction_stage(ops, dtype, shape):
    """dx = dy @ W when N (the contracted axis) is not a multiple of the kernel's stage: dy carries zero columns up to the next
    multiple of 64, W's rows are clamped (vocabulary projection, V = 4364 / 32); same for the one-launch dX + dW (bf16)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(N)
    D = dev()
    Np = (N + 63) // 64 * 64
    dy = torch.zeros(M, Np); dy[:, :N] = q(torch.randn(M, N, generator=g), dtype)
    # W sits inside a larger buffer whose following rows are NOT zero: they must not leak into the result
    wbuf = q(torch.randn(N + 64, K, generator=g) / math.sqrt(N), dtype)
    w = wbuf[:N]
    dyd, wd = dy.to(D, dtype), wbuf.to(D, dtype)[:N]
    assert ops.gemm_nn_supported(dyd, wd)
    if N % 64 != 0 and N % 8 == 0 and dtype == torch.bfloat16:
        assert not ops.gemm_nn_supported(dyd[:, :N].contiguous(), wd)      # no room for the zero columns
    out = ops.gemm_nn(dyd, wd)
    close("nn partial stage", out, dy[:, :N] @ w, dtype)
    if dtype == torch.bfloat16:
        x = q(torch.randn(M, K, generator=g), dtype)
        dw0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
        dw, db = dw0.clone().to(D), db0.clone().to(D)
        out2 = ops.gemm_nn_tn(dyd, wd, x.to(D, dtype), dw, db)
        ops.flush_tn_reduces()
        assert torch.equal(out2, out)
        close("nn_tn dW partial tile", dw, dw0 + dy[:, :N].t() @ x, dtype, scale=0.2)
        close("nn_tn db partial tile", db, db0 + dy[:, :N].sum(0), torch.float32, scale=8)


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_cast_colsum(ops, dtype):
    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(203, 77, generator=g), dtype)
    t = ops.transpose(x.to(dev(), dtype))
    assert torch.equal(t.float().cpu(), x.t())
    cs = torch.full((77,), 2.0, device=dev())
    tp = ops.transpose_padded(x.to(dev(), dtype), colsum_acc=cs)
    assert tp.shape == (77, 208) and (tp[:, 203:] == 0).all() and torch.equal(tp[:, :203].float().cpu(), x.t())
    close("transpose-fused colsum", cs, 2 + x.sum(0), torch.float32, scale=4)
    src = torch.randn(131, 45, generator=g)
    same, tr = ops.cast_and_transpose(src.to(dev()), dtype)
    assert torch.equal(same[:, :45].float().cpu(), q(src, dtype)) and (same[:, 45:] == 0).all()
    assert torch.equal(tr[:, :131].float().cpu(), q(src, dtype).t()) and (tr[:, 131:] == 0).all()
    acc = torch.ones(77, device=dev())
    ops.colsum_acc(x.to(dev(), dtype), acc)
    close("colsum", acc, 1 + x.sum(0), torch.float32, scale=4)


# ------------------------------------------------------------------------------------------------ LayerNorm epilogue
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [32, 512])
def test_add_ln_fwd_bwd(ops, dtype, D):
    g = torch.Generator().manual_seed(D)
    M, T = 50, 10
    y = q(torch.randn(M, D, generator=g), dtype)
    res = q(torch.randn(M, D, generator=g), dtype)
    gamma = 1 + 0.2 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    pe = torch.randn(T, D, generator=g)
    keep = (torch.rand(M, generator=g) > 0.3)
    dout = q(torch.randn(M, D, generator=g), dtype)

    yr, rr, gr, br = y.clone().requires_grad_(), res.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    z = q((yr + rr).detach(), dtype) + (yr + rr) - (yr + rr).detach()        # kernel normalises the ROUNDED z
    ref = (F.layer_norm(z, (D,), gr, br, 1e-5) + pe.repeat(M // T, 1)) * keep[:, None].float()
    ref.backward(dout)

    yz = y.to(dev(), dtype)
    out, mean, rstd = ops.add_ln_fwd(yz, res.to(dev(), dtype), gamma.to(dev()), beta.to(dev()), post_add=pe.to(dev()),
                                     row_keep=keep.to(torch.uint8).to(dev()))
    close("ln out", out, ref, dtype, scale=2)
    close("ln z (in place)", yz, q(y + res, dtype), dtype)
    dg = torch.zeros(D, device=dev()); db = torch.zeros(D, device=dev())
    d_res, d_y = ops.add_ln_bwd(dout.to(dev(), dtype), yz, mean, rstd, gamma.to(dev()), keep.to(torch.uint8).to(dev()), dg, db)
    assert d_y is d_res
    close("ln d_res", d_res, rr.grad, dtype, scale=4)
    close("ln dgamma", dg, gr.grad, torch.float32 if dtype == torch.float32 else dtype, scale=8 if dtype == torch.float32 else 2)
    close("ln dbeta", db, br.grad, torch.float32, scale=8)


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_ln_dropout_consistency(ops, dtype):
    """Dropout: keep-rate statistics, scaling, and the backward mask equals the forward mask."""
    M, D, p = 256, 512, 0.25
    y = torch.ones(M, D).to(dev(), dtype)
    gamma, beta = torch.ones(D, device=dev()), torch.zeros(D, device=dev())
    yz = y.clone()
    out, mean, rstd = ops.add_ln_fwd(yz, None, gamma, beta, p=p, seed=1234)
    zf = yz.float()
    kept = zf != 0
    rate = kept.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    assert torch.allclose(zf[kept], torch.full_like(zf[kept], 1 / (1 - p)), rtol=1e-2)
    # the mask must be uncorrelated along columns and rows (the counter hash is cheap 32-bit arithmetic)
    zc = (~kept).float().cpu() - p
    for sh, dim in ((1, 1), (2, 1), (8, 1), (1, 0), (4, 0)):
        corr = (zc * torch.roll(zc, sh, dim)).mean().item() / (p * (1 - p))
        assert abs(corr) < 0.02, (sh, dim, corr)
    dg = torch.zeros(D, device=dev()); db = torch.zeros(D, device=dev())
    dout = torch.randn(M, D, device=dev()).to(dtype)
    d_res, d_y = ops.add_ln_bwd(dout, yz, mean, rstd, gamma, None, dg, db, p=p, seed=1234)
    assert d_y is not d_res
    dyf, drf = d_y.float(), d_res.float()
    assert (dyf[~kept] == 0).all()
    assert torch.allclose(dyf[kept], drf[kept] / (1 - p), rtol=2e-2, atol=1e-6)
    yz2 = y.clone()
    ops.add_ln_fwd(yz2, None, gamma, beta, p=p, seed=99)
    assert not torch.equal(yz2, yz)


@pytest.mark.parametrize("tile", [0, 1, 128, 256])
def test_grouped_weight_gradients(ops, tile):
    """asr_gemm_tn_grouped: the weight and bias gradients of several linear layers in ONE launch (every linear layer's dW in the
    graph-replayed step, reference models/common_layers.py:136-142,181-187 via autograd) against fp32 torch.  Shapes: the model's
    (512 x 512 / 1536 x 512 / 2048 x 512 / 512 x 2048 over 400 .. 1700 rows), a vocabulary-like N = 300 with a padded leading dimension,
    K not a multiple of the block, M not a multiple of the 32-row stage, a problem without bias, an empty problem; both block
    sizes (TN_GROUP_TILE = 128: one block per contraction; 256: eight waves, rows cut in slices that meet in fp32 atomics; 1: the same
    blocks dealt out to one workgroup per CU in equal pieces of 32-row stages; 0, the default: equal pieces below 9 600 rows, slices
    from there on)."""
    from asr_hip import lib as L
    g = torch.Generator().manual_seed(17)
    D = dev()
    probs = []     # (M, N, K, ld_dy, ld_x, bias)
    for M, N, K, ldy, ldx, hb in [(1700, 512, 512, 512, 512, True), (933, 1536, 512, 1536, 512, True), (400, 2048, 512, 2048, 512, True),
                                 (1601, 512, 2048, 512, 2048, True), (640, 300, 512, 320, 512, True), (777, 512, 264, 512, 264, False),
                                 (0, 64, 64, 64, 64, True), (3300, 256, 256, 256, 256, True)]:
        dy = torch.zeros(max(M, 1), ldy)
        dy[:, :N] = torch.randn(max(M, 1), N, generator=g)
        x = torch.randn(max(M, 1), ldx, generator=g)
        dyd, xd = dy[:M].to(D).bfloat16(), x[:M].to(D).bfloat16()
        dw0 = torch.randn(N, K, generator=g)
        db0 = torch.randn(N, generator=g) if hb else None
        probs.append((dyd, xd, dw0.to(D), db0.to(D) if hb else None, N, K, dw0, db0))
    L.set_tuning("TN_GROUP_TILE", tile)
    try:
        ops.gemm_tn_grouped([pr[:6] for pr in probs])
        torch.cuda.synchronize()
    finally:
        L.set_tuning("TN_GROUP_TILE", None)
    for dyd, xd, dw, db, N, K, dw0, db0 in probs:
        dyf, xf = dyd.float().cpu(), xd.float().cpu()
        ref = dw0 + dyf[:, :N].t() @ xf[:, :K]
        tol = 2e-3 * max(1.0, float(ref.abs().max()))
        assert (dw.cpu() - ref).abs().max().item() < tol, (dyd.shape, N, K, (dw.cpu() - ref).abs().max().item())
        if db is not None:
            refb = db0 + dyf[:, :N].sum(0)
            assert (db.cpu() - refb).abs().max().item() < 2e-3 * max(1.0, float(refb.abs().max())), (N, K)


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(qx, kx, vx, H, d, key_len, key_pad, causal, scale):
    B, Tq, _ = qx.shape
    Tk = kx.shape[1]
    qh = qx.view(B, Tq, H, d).permute(0, 2, 1, 3)
    kh = kx.view(B, Tk, H, d).permute(0, 2, 1, 3)
    vh = vx.view(B, Tk, H, d).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) * scale
    mask = torch.zeros(B, 1, Tq, Tk, dtype=torch.bool)
    if key_len is not None:
        mask |= (torch.arange(Tk)[None, :] >= key_len[:, None].long())[:, None, None, :]
    if key_pad is not None:
        kp = key_pad.bool()
        mask |= kp[:, None, None, :] if kp.dim() == 2 else kp[:, None]
    if causal:
        mask |= torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), diagonal=1)[None, None]
    s = s.masked_fill(mask, float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = (a @ vh).permute(0, 2, 1, 3).reshape(B, Tq, H * d)
    return o, a


CASES = [
    dict(B=2, H=2, Tq=16, Tk=16, d=16, key_len=[16, 9], causal=False),
    dict(B=3, H=4, Tq=100, Tk=100, d=16, pad=True, causal=True),
    dict(B=2, H=8, Tq=100, Tk=200, d=64, key_len=[200, 37], causal=False),
    dict(B=2, H=2, Tq=130, Tk=75, d=32, key_len=[75, 60], causal=False),
    dict(B=1, H=2, Tq=70, Tk=70, d=64, full=True, causal=False),
    # the headline shapes: encoder self-attention of configs[1] (T'=... the north-star microbench T=800) and the
    # ragged T'=795 of configs[3]; decoder cross-attention Tq=100 over 795 keys; causal + key-pad decoder self-attention at Td=100
    dict(B=2, H=8, Tq=800, Tk=800, d=64, key_len=[800, 613], causal=False),
    dict(B=2, H=8, Tq=795, Tk=795, d=64, key_len=[795, 402], causal=False),
    dict(B=2, H=8, Tq=100, Tk=795, d=64, key_len=[700, 795], causal=False),
    dict(B=2, H=8, Tq=100, Tk=100, d=64, pad=True, causal=True),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_attention_fwd_bwd(ops, dtype, case):
    B, H, Tq, Tk, d = case["B"], case["H"], case["Tq"], case["Tk"], case["d"]
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    qx = q(torch.randn(B, Tq, H * d, generator=g), dtype)
    kx = q(torch.randn(B, Tk, H * d, generator=g), dtype)
    vx = q(torch.randn(B, Tk, H * d, generator=g), dtype)
    do = q(torch.randn(B, Tq, H * d, generator=g), dtype)
    key_len = torch.tensor(case["key_len"], dtype=torch.int32) if "key_len" in case else None
    key_pad = None
    if case.get("pad"):
        key_pad = torch.zeros(B, Tk, dtype=torch.uint8)
        for b in range(B):
            key_pad[b, Tk - 7 * (b + 1):] = 1
    if case.get("full"):
        key_pad = (torch.rand(B, Tq, Tk, generator=g) > 0.7).to(torch.uint8)
        key_pad[:, :, 0] = 0
    scale = 1.0 / math.sqrt(d)
    qr, kr, vr = qx.clone().requires_grad_(), kx.clone().requires_grad_(), vx.clone().requires_grad_()
    oref, aref = attn_ref(qr, kr, vr, H, d, key_len, key_pad, case["causal"], scale)
    oref.backward(do)
    D = dev()
    kl = key_len.to(D) if key_len is not None else None
    kp = key_pad.to(D) if key_pad is not None else None
    o, lse, attn = ops.attn_fwd(qx.to(D, dtype), kx.to(D, dtype), vx.to(D, dtype), H, d, key_len=kl, key_pad=kp,
                                causal=case["causal"], scale=scale, want_attn=True)
    close("attn out", o, oref, dtype, scale=2)
    aref_hb = aref.permute(1, 0, 2, 3).reshape(H * B, Tq, Tk)
    close("attn probs (H*B layout)", attn, aref_hb, dtype, scale=2, atol=1e-6)
    dq, dk, dv = ops.attn_bwd(qx.to(D, dtype), kx.to(D, dtype), vx.to(D, dtype), o, do.to(D, dtype), lse, H, d, key_len=kl,
                              key_pad=kp, causal=case["causal"], scale=scale)
    close("attn dq", dq, qr.grad, dtype, scale=4)
    close("attn dk", dk, kr.grad, dtype, scale=4)
    close("attn dv", dv, vr.grad, dtype, scale=4)


def test_attention_bwd_delta_from_unrounded_output(ops):
    """bf16 storage: delta = rowsum(dO * O) from the forward's fp32 copy of O (asr_attn_fwd o32).  When the value rows share a
    common component (LayerNorm bias / positional part of the stream) the rounding error of a bf16 O does not cancel against
    dP = dO.V^T and dominates dS = P (dP - delta); with the fp32 copy dQ / dK keep plain bf16 accuracy.  Truth: fp64 on the
    bf16-rounded operands.  Decoder self-attention shape of the benchmark (B=2, H=8, T=100, d=64, causal + key pad)."""
    B, H, T, d = 2, 8, 100, 64
    g = torch.Generator().manual_seed(11)
    dtype = torch.bfloat16
    qx = q(torch.randn(B, T, H * d, generator=g) * 0.5, dtype)
    kx = q(torch.randn(B, T, H * d, generator=g) * 0.5, dtype)
    vx = q(torch.randn(B, T, H * d, generator=g) + 6.0 * torch.randn(1, 1, H * d, generator=g), dtype)
    do = q(torch.randn(B, T, H * d, generator=g), dtype)
    key_pad = torch.zeros(B, T, dtype=torch.uint8)
    key_pad[1, 70:] = 1
    scale = 1.0 / math.sqrt(d)
    qr, kr, vr = (t.double().clone().requires_grad_() for t in (qx, kx, vx))
    oref, _ = attn_ref(qr, kr, vr, H, d, None, key_pad, True, scale)
    oref.backward(do.double())
    D = dev()
    a = [t.to(D, dtype) for t in (qx, kx, vx)]
    o32 = torch.empty(B, T, H * d, device=D, dtype=torch.float32)
    o, lse, _ = ops.attn_fwd(*a, H, d, key_pad=key_pad.to(D), causal=True, scale=scale, o32=o32)
    assert (o32.cpu().double() - oref.detach()).abs().max() < 5e-3 * oref.detach().abs().max() and torch.equal(o32.to(dtype), o)
    err = {}
    for tag, buf in (("bf16_O", None), ("fp32_O", o32)):
        dq, dk, dv = ops.attn_bwd(*a, o, do.to(D, dtype), lse, H, d, key_pad=key_pad.to(D), causal=True, scale=scale, o32=buf)
        rel = lambda x, r: float((x.double().cpu() - r).norm() / r.norm())
        err[tag] = (rel(dq, qr.grad), rel(dk, kr.grad), rel(dv, vr.grad))
    print("attention backward relative L2 error (dq, dk, dv):", err)
    assert max(err["fp32_O"]) < 1e-2, err
    assert err["fp32_O"][0] < 0.6 * err["bf16_O"][0] and err["fp32_O"][1] < 0.6 * err["bf16_O"][1], err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,T", [(32, 96), (64, 200)])
def test_attention_dropout_consistency(ops, dtype, d, T):
    """With dropout the returned attention matrix IS the dropped/scaled one; O must equal attn @ V and the backward
    must use the same mask (checked through dV = attn^T dO).  d = 64 / bf16 runs the specialised kernels."""
    B, H, p = 2, 2, 0.2
    g = torch.Generator().manual_seed(5)
    D = dev()
    qx = torch.randn(B, T, H * d, generator=g).to(D, dtype)
    kx = torch.randn(B, T, H * d, generator=g).to(D, dtype)
    vx = torch.randn(B, T, H * d, generator=g).to(D, dtype)
    do = torch.randn(B, T, H * d, generator=g).to(D, dtype)
    o, lse, attn = ops.attn_fwd(qx, kx, vx, H, d, scale=0.2, p=p, seed=77, want_attn=True)
    a = attn.view(H, B, T, T).permute(1, 0, 2, 3).float().cpu()
    drop_rate = (a == 0).float().mean().item()
    assert abs(drop_rate - p) < 0.02, drop_rate
    # the mask must not be correlated along keys, queries or heads (a weak hash would show up here)
    z = (a == 0).float() - p
    for sh in ((0, 0, 0, 1), (0, 0, 1, 0), (0, 1, 0, 0), (0, 0, 0, 2)):
        zz = torch.roll(z, shifts=sh, dims=(0, 1, 2, 3))
        corr = (z * zz).mean().item() / (p * (1 - p))
        assert abs(corr) < 0.02, (sh, corr)
    vh = vx.float().cpu().view(B, T, H, d).permute(0, 2, 1, 3)
    close("O == attn_dropped @ V", o, (a @ vh).permute(0, 2, 1, 3).reshape(B, T, H * d), dtype, scale=3)
    _, _, dv = ops.attn_bwd(qx, kx, vx, o, do, lse, H, d, scale=0.2, p=p, seed=77)
    doh = do.float().cpu().view(B, T, H, d).permute(0, 2, 1, 3)
    close("dV == attn_dropped^T @ dO", dv, (a.transpose(-1, -2) @ doh).permute(0, 2, 1, 3).reshape(B, T, H * d), dtype, scale=4)


@pytest.mark.parametrize("B,H,Tq,Tk,causal,mask,p", [
    (2, 8, 200, 200, False, "len", 0.1), (2, 8, 100, 200, False, "len", 0.1), (2, 8, 100, 100, True, "pad", 0.1), (3, 4, 130, 75, False, "len", 0.0),
    (2, 2, 70, 256, False, "full", 0.25), (1, 8, 333, 131, False, None, 0.1), (2, 4, 64, 64, True, None, 0.0), (1, 2, 5, 3, False, None, 0.1)])
def test_attention_backward_in_one_pass_equals_the_two_halves(ops, B, H, Tq, Tk, causal, mask, p):
    """Short key sequences (Tk <= 256, bf16, d = 64): the whole backward of a (batch, head) in one workgroup and one pass over the scores
    (csrc/attention_fast.hip attn_bwd_fused_body; reference: autograd of models/common_layers.py:211-225) against the two-half launch it
    replaces (tuning ATTN_BWD_FUSED = 0) on the same inputs, masks, dropout seed and delta: both form P, the dropout mask and dS from
    the same fp32 scores with the same instructions, so dK / dV (same contraction, same order) are EQUAL and dQ -- contracted over the
    keys from the bf16 dS^T through LDS instead of from fp32-accumulated per-wave fragments -- is within bf16 rounding of it."""
    from asr_hip import lib as L
    d, bf = 64, torch.bfloat16
    g = torch.Generator().manual_seed(Tq * 3 + Tk)
    D = dev()
    qx, kx, vx = (torch.randn(B, T, H * d, generator=g).to(D, bf) for T in (Tq, Tk, Tk))
    do = torch.randn(B, Tq, H * d, generator=g).to(D, bf)
    kl = kp = None
    if mask == "len":
        kl = torch.randint(max(1, Tk // 3), Tk + 1, (B,), generator=g).to(torch.int32).to(D)
    elif mask == "pad":
        kp = torch.zeros(B, Tk, dtype=torch.uint8)
        for b in range(B):
            kp[b, Tk - 7 * (b + 1):] = 1
        kp = kp.to(D)
    elif mask == "full":
        kp = (torch.rand(B, Tq, Tk, generator=g) > 0.7).to(torch.uint8)
        kp[:, :, 0] = 0
        kp = kp.to(D)
    o32 = torch.empty(B, Tq, H * d, device=D, dtype=torch.float32)
    o, lse, _ = ops.attn_fwd(qx, kx, vx, H, d, key_len=kl, key_pad=kp, causal=causal, scale=0.125, p=p, seed=91, o32=o32)
    got = {}
    try:
        for fused in (1, 0):
            L.set_tuning("ATTN_BWD_FUSED", fused)
            got[fused] = ops.attn_bwd(qx, kx, vx, o, do, lse, H, d, key_len=kl, key_pad=kp, causal=causal, scale=0.125, p=p, seed=91, o32=o32)
    finally:
        L.set_tuning("ATTN_BWD_FUSED", None)
    torch.cuda.synchronize()
    assert torch.equal(got[1][1], got[0][1]), "dK"
    assert torch.equal(got[1][2], got[0][2]), "dV"
    a, b_ = got[1][0].float(), got[0][0].float()
    assert torch.isfinite(a).all()
    # both contract bf16-rounded dS (2^-9 relative per term) over the keys, in different orders and with the dropout rescale on different
    # sides of the rounding: a few 2^-9 of the LARGEST terms, i.e. of the tensor's scale, and a small relative L2 error
    assert (a - b_).abs().max() <= 2.0 ** -7 * b_.abs().max(), ((a - b_).abs().max().item(), b_.abs().max().item())
    assert (a - b_).norm() <= 4e-3 * b_.norm(), ((a - b_).norm().item(), b_.norm().item())


# ------------------------------------------------------------------------------------------------ decoder input side
def test_decoder_preprocess_matches_oracle(ops):
    from oracle import asr_oracle as O
    g = torch.Generator().manual_seed(11)
    B, Lw, Td = 5, 14, 16
    tgt = torch.randint(3, 30, (B, Lw), generator=g)
    tgt[0, 10:] = 0
    tgt[1, 3] = 0           # interior PAD
    tgt[2, :] = 0           # empty target
    tgt[3, 14:] = 0
    si_ref, so_ref = O.decoder_preprocess(tgt, Td)
    si, so, key_pad, row_keep, ovf = ops.decoder_preprocess(tgt.to(dev()), Td)
    assert torch.equal(si.cpu(), si_ref) and torch.equal(so.cpu(), so_ref)
    assert torch.equal(key_pad.cpu().bool(), si_ref == O.EOS) and torch.equal(row_keep.cpu().bool(), si_ref != O.EOS)
    assert int(ovf.item()) == 0
    _, _, _, _, ovf = ops.decoder_preprocess(torch.randint(3, 30, (2, 20), generator=g).to(dev()), Td)
    assert int(ovf.item()) == 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding_fwd_bwd(ops, dtype):
    g = torch.Generator().manual_seed(2)
    B, T, D, V = 3, 12, 64, 35
    tok = torch.randint(0, V, (B, T), generator=g)
    table = torch.randn(V, D, generator=g)
    pe = torch.randn(T, D, generator=g)
    out = ops.embed_fwd(tok.to(dev()), table.to(dev()), pe.to(dev()), 0.5, 0.0, 0, dtype)
    close("embed", out, table[tok] * 0.5 + pe, dtype)
    dout = q(torch.randn(B, T, D, generator=g), dtype)
    dt = torch.zeros(V, D, device=dev())
    ops.embed_bwd(tok.to(dev()), dout.to(dev(), dtype), dt, 0.5, 0.0, 0, 0)
    ref = torch.zeros(V, D).index_add_(0, tok.reshape(-1), dout.reshape(-1, D) * 0.5)
    ref[0] = 0                                      # padding_idx row gets no gradient
    close("embed bwd", dt, ref, torch.float32, scale=4)


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("smoothing", [0.0, 0.1])
@pytest.mark.parametrize("V", [35, 4364])
def test_ce_matches_oracle(ops, smoothing, V):
    from oracle import asr_oracle as O
    g = torch.Generator().manual_seed(V)
    M = 60
    logits = torch.randn(M, V, generator=g)
    gold = torch.randint(1, V, (M,), generator=g)
    gold[::5] = 0
    logits[7] = 0.0                                  # exact tie over the whole row -> argmax must be 0
    logits[8, 5] = logits[8, 20] = 9.0               # two-way tie -> lowest index
    lr = logits.clone().requires_grad_()
    loss_ref, ncorrect, nword = O.smoothed_ce(lr.view(1, M, V), gold.view(1, M), smoothing)
    loss_ref.backward()
    lse, am, sums = ops.ce_fwd(logits.to(dev()), gold.to(dev()), smoothing, 0)
    s = sums.cpu()
    assert int(s[1]) == nword and int(s[2]) == ncorrect
    assert abs(s[0].item() / s[1].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item()))
    assert torch.equal(am.cpu(), logits.argmax(1)) and am[7].item() == 0 and am[8].item() == 5
    assert torch.equal(ops.argmax_rows(logits.to(dev())).cpu(), logits.argmax(1))
    gout = torch.tensor([1.7], device=dev())
    dl = ops.ce_bwd(logits.to(dev()), gold.to(dev()), lse, smoothing, 0, gout, sums[1:2])
    close("dlogits", dl, 1.7 * lr.grad, torch.float32, scale=8, atol=1e-9)
    assert dl.stride(0) % 8 == 0


@pytest.mark.parametrize("M,V", [(60, 35), (3200, 4364), (7, 4364)])
def test_ce_statistics_in_a_fixed_order(ops, M, V):
    """asr_ce_fwd_partials + asr_ce_finish (round 6): the three statistics and the mean loss are the SAME BITS run to run (per-block partial
    sums added in a fixed order; asr_ce_fwd's fp32 atomics are equal only up to their order), equal to the atomics' to rounding, exact in
    the two counts, and the loss is sums[0] / sums[1] -- or / den when a device scalar is given."""
    g = torch.Generator().manual_seed(M + V)
    logits = (torch.randn(M, V, generator=g) * 3).to(dev())
    gold = torch.randint(1, V, (M,), generator=g)
    gold[::5] = 0
    gold = gold.to(dev())
    runs = [ops.ce_fwd_det(logits, gold, 0.1, 0) for _ in range(4)]
    lse0, am0, s0, l0 = runs[0]
    for lse, am, s, l in runs[1:]:
        assert torch.equal(s, s0) and torch.equal(l, l0) and torch.equal(lse, lse0) and torch.equal(am, am0)
    lse_a, am_a, s_a = ops.ce_fwd(logits, gold, 0.1, 0)
    assert torch.equal(lse_a, lse0) and torch.equal(am_a, am0)
    assert s0[1].item() == s_a[1].item() == float((gold != 0).sum()) and s0[2].item() == s_a[2].item()
    assert abs(s0[0].item() - s_a[0].item()) <= 1e-5 * abs(s_a[0].item())
    assert l0.item() == (s0[0] / s0[1]).item()
    den = torch.tensor([123.0], device=dev())
    assert ops.ce_fwd_det(logits, gold, 0.1, 0, den=den)[3].item() == (s0[0] / den[0]).item()


# ------------------------------------------------------------------------------------------------ optimiser
def test_adam_matches_oracle(ops):
    from oracle import asr_oracle as O
    g = torch.Generator().manual_seed(9)
    n = 1003
    p0 = torch.randn(n, generator=g)
    params = {"w": p0.clone()}
    opt = O.NoamAdam(params, model_size=5120)
    p = torch.zeros(1008, device=dev()); p[:n] = p0.to(dev())
    gbuf = torch.zeros(1008, device=dev()); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for t in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** (t - 2))
        opt.step({"w": grad})
        gbuf[:n] = grad.to(dev())
        lr = O.noam_rate(t, 5120, 1.0, 4000, 1e-5)
        ops.adam_step(p[:n + 1], gbuf[:n + 1], m[:n + 1], v[:n + 1], lr, 0.9, 0.98, 1e-9, t)    # odd length: scalar tail
        assert torch.allclose(p[:n].cpu(), params["w"], rtol=0, atol=2e-7), t
    acc = torch.zeros(1, device=dev())
    ops.sumsq_acc(gbuf, acc)
    coef = torch.zeros(1, device=dev())
    ops.clip_coef(acc, 0.5, coef)
    nrm = gbuf.norm().item()
    assert abs(acc.sqrt().item() - nrm) < 1e-3 * nrm and abs(coef.item() - min(1.0, 0.5 / (nrm + 1e-6))) < 1e-6


def test_device_side_step_is_cancelled_by_a_non_finite_guard(ops):
    """asr_adam_noam_step(guard_dev): a non-finite loss sum (or gradient scale) leaves parameters and both moments untouched --
    the reference trainer's `if loss == inf: continue` (trainer/asr/trainer.py:102-104) for a step replayed from a hipGraph."""
    from asr_hip import ops as O
    D = dev()
    n = 1001
    g = torch.Generator().manual_seed(5)
    p = torch.randn(n, generator=g).to(D); grad = torch.randn(n, generator=g).to(D)
    m = torch.rand(n, generator=g).to(D); v = torch.rand(n, generator=g).to(D)
    O.step_state(D)[1] = 3
    for bad in (float("inf"), float("nan"), float("-inf")):
        p0, m0, v0 = p.clone(), m.clone(), v.clone()
        O.adam_noam_step(p, grad, m, v, 0.9, 0.98, 1e-9, 0.01, 4000.0, 1e-5, guard=torch.tensor([bad, 1.0], device=D))
        assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0), bad
        O.adam_noam_step(p, grad, m, v, 0.9, 0.98, 1e-9, 0.01, 4000.0, 1e-5, grad_scale=torch.tensor([bad], device=D))
        assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0), bad
    p0 = p.clone()
    O.adam_noam_step(p, grad, m, v, 0.9, 0.98, 1e-9, 0.01, 4000.0, 1e-5, guard=torch.tensor([2.5, 1.0], device=D))
    assert not torch.equal(p, p0)


# ------------------------------------------------------------------------------------------------ conv front end
def nhwc(x):       # (B,C,H,W) -> (B,H,W,C)
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 21, 37), (5, 161, 331)])     # the second: several tiles per workgroup (4-stage DMA pipeline)
def test_conv1_fwd_wgrad(ops, dtype, shape):
    g = torch.Generator().manual_seed(4)
    (B, H, W), C0 = shape, 64
    x = torch.randn(B, 1, H, W, generator=g)
    w = torch.randn(C0, 1, 3, 3, generator=g) / 3
    b = torch.randn(C0, generator=g) / 3
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.relu(F.conv2d(x, wr, br, padding=1))
    y = ops.conv1_fwd(x.to(dev()), w.to(dev()), b.to(dev()), dtype)
    close("conv1", y, nhwc(ref), dtype)
    dy = q(torch.randn(B, H, W, C0, generator=g), dtype) * (nhwc(ref) > 0)
    ref.backward(nchw(dy))
    dw = torch.zeros_like(w, device=dev()); db = torch.zeros(C0, device=dev())
    ops.conv1_wgrad(x.to(dev()), dy.to(dev(), dtype), dw, db)
    close("conv1 dw", dw, wr.grad, torch.float32, scale=16)
    close("conv1 db", db, br.grad, torch.float32, scale=16)


def test_conv1_wgrad_fp32_accuracy_at_the_benchmark_image(ops):
    """fp32 parity mode at 161 x 800 (the benchmark's spectrogram): the first layer's weight gradient is a sum of ~10^5
    uncorrelated products per batch element (|g| ~ sqrt(N) |term|), so accumulation order matters.  Truth: fp64; the bound is
    a small multiple of what torch's own fp32 convolution backward loses on the same tensors."""
    g = torch.Generator().manual_seed(8)
    B, H, W, C0 = 2, 161, 800, 64
    x = torch.randn(B, 1, H, W, generator=g)
    x[1, :, :, 170:] = 0
    dy = torch.randn(B, H, W, C0, generator=g) * (torch.rand(B, H, W, C0, generator=g) > 0.5)
    w64 = torch.zeros(C0, 1, 3, 3, dtype=torch.float64, requires_grad=True)
    b64 = torch.zeros(C0, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, b64, padding=1).backward(nchw(dy).double())
    w32 = torch.zeros(C0, 1, 3, 3, requires_grad=True)
    b32 = torch.zeros(C0, requires_grad=True)
    F.conv2d(x, w32, b32, padding=1).backward(nchw(dy))
    dw = torch.zeros(C0, 1, 3, 3, device=dev()); db = torch.zeros(C0, device=dev())
    ops.conv1_wgrad(x.to(dev()), dy.to(dev()), dw, db)
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    e = dict(ours_dw=rel(dw, w64.grad), torch_dw=rel(w32.grad, w64.grad), ours_db=rel(db, b64.grad), torch_db=rel(b32.grad, b64.grad))
    print("conv1 wgrad fp32 relative L2 error vs fp64:", e)
    assert e["ours_dw"] <= max(2e-5, 4 * e["torch_dw"]) and e["ours_db"] <= max(2e-5, 4 * e["torch_db"]), e


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 21, 37, 64, 64), (1, 16, 32, 64, 128), (2, 9, 50, 128, 128), (1, 11, 19, 128, 64),
                                 (1, 80, 400, 64, 128), (1, 80, 400, 128, 128)])       # the last two: configs[1] image size
def test_conv3x3_fwd_dgrad_wgrad(ops, dtype, cfg):
    B, H, W, Cin, Cout = cfg
    g = torch.Generator().manual_seed(H * W + Cin)
    x = q(torch.randn(B, Cin, H, W, generator=g).relu(), dtype)         # an upstream ReLU output
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g) / 3
    D = dev()
    wk = torch.empty(Cout, 9, Cin, device=D, dtype=dtype); wd = torch.empty(Cin, 9, Cout, device=D, dtype=dtype)
    ops.conv_pack_weight(w.to(D), wk, wd)
    wq = q(w, dtype)
    xr, wr, br = x.clone().requires_grad_(), wq.clone().requires_grad_(), b.clone().requires_grad_()
    pre = F.conv2d(xr, wr, br, padding=1)
    ref = F.relu(pre)
    y = ops.conv3x3(nhwc(x).to(D, dtype), wk, b.to(D), Cout, relu=True)
    close("conv3x3 fwd", y, nhwc(ref), dtype)
    dy = q(torch.randn(B, H, W, Cout, generator=g), dtype) * (nhwc(ref) > 0)
    ref.backward(nchw(dy))
    # dgrad, masked by the ReLU that produced x
    dx = ops.conv3x3(dy.to(D, dtype), wd, None, Cin, relu=False, mask_src=nhwc(x).to(D, dtype))
    close("conv3x3 dgrad", dx, nhwc(xr.grad * (x > 0)), dtype, scale=2)
    db = torch.zeros(Cout, device=D)
    ops.colsum_acc(dy.to(D, dtype).view(-1, Cout), db)
    close("conv3x3 db", db, br.grad, torch.float32, scale=16)
    # NHWC-native kernel (transposing LDS reads), with the bias gradient fused
    dwn = torch.zeros(Cout, Cin, 3, 3, device=D); dbn = torch.zeros(Cout, device=D)
    ops.conv3x3_wgrad_nhwc(nhwc(x).to(D, dtype), dy.to(D, dtype), dwn, dbn)
    close("conv3x3 wgrad (NHWC native)", dwn, wr.grad, torch.float32 if dtype == torch.float32 else dtype, scale=16 if dtype == torch.float32 else 1)
    close("conv3x3 db (NHWC native)", dbn, br.grad, torch.float32, scale=16)
    # accumulation into existing gradients, with and without the two-stage workspace
    ops.conv3x3_wgrad_nhwc(nhwc(x).to(D, dtype), dy.to(D, dtype), dwn, dbn)
    close("conv3x3 wgrad accumulates", dwn, 2 * wr.grad, torch.float32 if dtype == torch.float32 else dtype, scale=16 if dtype == torch.float32 else 1)


@pytest.mark.parametrize("shape", [(3, 161, 232), (2, 21, 37), (1, 8, 16), (2, 9, 50)])
def test_conv3x3_relu_pool_fused(ops, shape):
    """conv.2 + ReLU + MaxPool2d from one epilogue (asr_conv3x3_relu_pool) == the convolution followed by asr_maxpool_fwd, bit for
    bit (same accumulators, same rounding; the pooled maximum is taken on the rounded bf16 values in both)."""
    B, H, W = shape
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H * 7 + W)
    x = q(torch.randn(B, H, W, 64, generator=g).relu(), dtype)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g) / 3
    D = dev()
    wk = torch.empty(64, 9, 64, device=D, dtype=dtype); wd = torch.empty(64, 9, 64, device=D, dtype=dtype)
    ops.conv_pack_weight(w.to(D), wk, wd)
    xd = x.to(D, dtype)
    y_ref = ops.conv3x3(xd, wk, b.to(D), 64, relu=True)
    p_ref = ops.maxpool_fwd(y_ref)
    y, pool = ops.conv3x3_relu_pool(xd, wk, b.to(D), 64)
    assert torch.equal(y, y_ref)
    assert pool.shape == p_ref.shape and torch.equal(pool, p_ref)
    # the same epilogue with selection codes, with and without the un-pooled output: same pool; the codes route a gradient exactly
    # like the pooling backward that re-reads y (first maximum in scan order, nothing where the maximum is 0)
    dy = q(torch.randn(B, H // 2, W // 2, 64, generator=g), dtype).to(D, dtype)
    dx_ref = ops.maxpool_bwd(y_ref, dy)
    for keep_y in (True, False):
        y2, pool2, code = ops.conv3x3_relu_pool_code(xd, wk, b.to(D), 64, keep_y=keep_y)
        assert torch.equal(pool2, p_ref) and (y2 is None) == (not keep_y) and (y2 is None or torch.equal(y2, y_ref))
        assert code.dtype == torch.uint8 and int(code.max()) <= 4 and torch.equal(code == 0, p_ref == 0)
        assert torch.equal(ops.maxpool_bwd_code(code, dy, tuple(y_ref.shape)), dx_ref)


@pytest.mark.parametrize("cfg", [(2, 32, 48, 128), (1, 16, 16, 64), (3, 80, 64, 128)])
def test_conv3x3_relu_pool_tcf_fused(ops, cfg):
    """conv.7 + ReLU + MaxPool2d + the (B, T', C F') transpose from the convolution's epilogue (asr_conv3x3_relu_pool_tcf_code) ==
    the convolution followed by asr_maxpool_fwd_code in the encoder layout, bit for bit: pooled values AND selection bytes."""
    B, H, W, Cin = cfg
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H * 3 + W + Cin)
    x = q(torch.randn(B, H, W, Cin, generator=g).relu(), dtype)
    w = torch.randn(128, Cin, 3, 3, generator=g) / (3 * math.sqrt(Cin))
    b = torch.randn(128, generator=g) / 3
    D = dev()
    wk = torch.empty(128, 9, Cin, device=D, dtype=dtype); wd = torch.empty(Cin, 9, 128, device=D, dtype=dtype)
    ops.conv_pack_weight(w.to(D), wk, wd)
    xd = x.to(D, dtype)
    y_ref = ops.conv3x3(xd, wk, b.to(D), 128, relu=True)
    p_ref, c_ref = ops.maxpool_fwd_code(y_ref, tcf=True)
    pool, code = ops.conv3x3_relu_pool_tcf_code(xd, wk, b.to(D), 128)
    assert pool.shape == p_ref.shape and torch.equal(pool, p_ref)
    assert torch.equal(code, c_ref)
    assert ops.conv3x3_relu_pool_tcf_code(xd[:, :H - 1].contiguous(), wk, b.to(D), 128) is None      # H not a multiple of 16: not taken


@pytest.mark.parametrize("cfg", [(3, 97, 130, 64, 128), (2, 50, 200, 128, 64)])
def test_conv3x3_wgrad_dma_pipeline(ops, cfg):
    """bf16 weight gradient on the LDS-DMA pipelined kernel (conv_wgrad_dma.hip) at sizes with interior AND border patches, several
    patches per workgroup and more than one dW block; reference = fp32 autograd of the same bf16-rounded operands."""
    B, H, W, Cin, Cout = cfg
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H + W)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    dy = q(torch.randn(B, Cout, H, W, generator=g) / 8, dtype)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    bias = torch.zeros(Cout, requires_grad=True)
    F.conv2d(x, w, bias, padding=1).backward(dy)
    D = dev()
    dw = torch.zeros(Cout, Cin, 3, 3, device=D); db = torch.zeros(Cout, device=D)
    ops.conv3x3_wgrad_nhwc(nhwc(x).to(D, dtype), nhwc(dy).to(D, dtype), dw, db)
    close("wgrad dma dW", dw, w.grad, dtype)
    close("wgrad dma db", db, bias.grad, torch.float32, scale=64)


@pytest.mark.parametrize("tw", ["0", "1", "2"])
@pytest.mark.parametrize("cfg", [(3, 161, 232, True), (5, 97, 401, False), (1, 5, 7, True)])
def test_conv3x3_c64_persistent(ops, tw, cfg, monkeypatch):
    """64 -> 64 channel bf16 layer on the persistent kernel (conv_c64.hip): more tiles than workgroups, so that the three-deep
    patch pipeline, both tile shapes, image borders and the masked (dgrad) epilogue all run; reference = fp32 conv of the same
    bf16-rounded operands."""
    B, H, W, masked = cfg
    monkeypatch.setenv("ASR_C64_SHAPE", tw)
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H * W)
    x = q(torch.randn(B, 64, H, W, generator=g).relu(), dtype)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g) / 3
    D = dev()
    wk = torch.empty(64, 9, 64, device=D, dtype=dtype); wd = torch.empty(64, 9, 64, device=D, dtype=dtype)
    ops.conv_pack_weight(w.to(D), wk, wd)
    ref = F.conv2d(x, q(w, dtype), b, padding=1)
    if masked:
        m = q(torch.randn(B, 64, H, W, generator=g), dtype)
        y = ops.conv3x3(nhwc(x).to(D, dtype), wk, b.to(D), 64, relu=False, mask_src=nhwc(m).to(D, dtype))
        close("c64 masked", y, nhwc(ref * (m > 0)), dtype, scale=2)
    else:
        y = ops.conv3x3(nhwc(x).to(D, dtype), wk, b.to(D), 64, relu=True)
        close("c64 relu", y, nhwc(F.relu(ref)), dtype, scale=2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 21, 38, 64), (1, 161, 16, 64), (2, 40, 16, 128), (2, 80, 12, 128)])
def test_maxpool_fwd_bwd_both_layouts(ops, dtype, shape):
    B, H, W, C = shape
    g = torch.Generator().manual_seed(H + W)
    x = q(torch.randn(B, C, H, W, generator=g).relu(), dtype)
    xr = x.clone().requires_grad_()
    ref = F.max_pool2d(xr, 2, stride=2)
    D = dev()
    xd = nhwc(x).to(D, dtype)
    y = ops.maxpool_fwd(xd)
    assert torch.equal(y.float().cpu(), nhwc(ref.detach()))
    ytcf = ops.maxpool_fwd(xd, tcf=True)
    Bq, Cq, H2, W2 = ref.shape
    ref_tcf = ref.detach().reshape(Bq, Cq * H2, W2).transpose(1, 2).contiguous()      # transformer.py:74-76
    assert torch.equal(ytcf.float().cpu(), ref_tcf)
    dy = q(torch.randn(B, C, H2, W2, generator=g), dtype)
    ref.backward(dy)
    want = nhwc(xr.grad * (x > 0))
    dx = ops.maxpool_bwd(xd, nhwc(dy).to(D, dtype))
    assert torch.equal(dx.float().cpu(), want)
    dy_tcf = dy.reshape(B, C * H2, W2).transpose(1, 2).contiguous()
    dx2 = ops.maxpool_bwd(xd, dy_tcf.to(D, dtype), tcf=True)
    assert torch.equal(dx2.float().cpu(), want)
    # selection codes instead of the activations in backward (asr_maxpool_fwd_code / asr_maxpool_bwd_code), both layouts
    for tcf, gy in ((False, nhwc(dy)), (True, dy_tcf)):
        res = ops.maxpool_fwd_code(xd, tcf=tcf)
        if res is None:                     # no 16-byte form for this shape: the library said so, callers keep the pair above
            assert C % (4 if dtype == torch.float32 else 8) != 0 or (tcf and H2 % (4 if dtype == torch.float32 else 8) != 0)
            continue
        yc, code = res
        assert torch.equal(yc, ytcf if tcf else y)
        dxc = ops.maxpool_bwd_code(code, gy.to(D, dtype), tuple(xd.shape), tcf=tcf)
        assert torch.equal(dxc.float().cpu(), want)


def test_window_sum_folds_the_taps_along_time(ops):
    """asr_window_sum: y[g*OW + j, co] = bias[co] + sum_kx Z[g*Wg + j + kx, kx*Cout + co]; the pad columns of y are zeroed."""
    G, Wg, KW, Cout, ldz, ldy = 5, 23, 11, 32, 352, 64
    OW = Wg - KW + 1
    g = torch.Generator().manual_seed(3)
    Z = torch.randn(G * Wg + KW - 1, ldz, generator=g)
    bias = torch.randn(Cout, generator=g)
    want = torch.zeros(G * OW, ldy)
    for gi in range(G):
        for j in range(OW):
            acc = bias.clone()
            for kx in range(KW):
                acc = acc + Z[gi * Wg + j + kx, kx * Cout:(kx + 1) * Cout]
            want[gi * OW + j, :Cout] = acc
    D = dev()
    y = torch.full((G * OW + 3, ldy), 7.0, device=D)
    ops.window_sum(Z.to(D), y, bias.to(D), G, Wg, OW, KW, Cout)
    assert torch.equal(y[:G * OW].cpu(), want)
    assert bool((y[G * OW:] == 7.0).all())


@pytest.mark.parametrize("shift", ["1", "0"])
@pytest.mark.parametrize("cfg", [(2, 61, 37, 32, 21, 11, 2, 1, 0), (2, 161, 50, 1, 41, 11, 2, 2, 10)])
def test_window_convolutions_against_conv2d(ops, shift, cfg, monkeypatch):
    """The two convolutions of emb_cnn (reference transformer.py:33-40: 1 -> 32, 41 x 11, stride (2,2), time padding 10, and 32 -> 32,
    21 x 11, stride (2,1)) in their bf16 forms -- shift = 1: dense product over single-step patches + window sum (unit stride) and the
    packet-of-rows weight-gradient contraction; shift = 0: the window-view GEMMs -- against F.conv2d in fp32 on the same bf16-rounded
    operands: output, weight and bias gradient, and (unit stride) the data gradient."""
    from asr_hip import functions as Fn
    monkeypatch.setattr(Fn, "_emb_shift_fwd", shift == "1")
    monkeypatch.setattr(Fn, "_emb_shift_wgrad", shift == "1")
    B, H, W, C, KH, KW, SH, SW, PW = cfg
    Cout = 32
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, C, KH, KW, generator=g) * (C * KH * KW) ** -0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv2d(xr, wr, br, stride=(SH, SW), padding=(0, PW))
    OH, OW = ref.shape[2], ref.shape[3]
    dy = torch.randn(B, Cout, OH, OW, generator=g).bfloat16().float()
    ref.backward(dy)
    D = dev()
    prev = ops.compute_dtype()
    ops.set_compute_dtype(torch.bfloat16)
    try:
        geo = ops.conv_geom(B, H, W, C, KH, KW, SH, SW, 0, PW)
        assert Fn._window_ok(geo)
        tag = "t_win%d" % SW
        wd, bd = torch.nn.Parameter(w.to(D)), torch.nn.Parameter(b.to(D))
        xin = nhwc(x).to(D, torch.bfloat16 if C > 1 else torch.float32).contiguous()
        X2, A, y, M, (yW, yOW) = Fn._conv_window_fwd(xin, geo, wd, bd, tag)
        # the strided form leaves y on the GEMM's row grid (groups of yW rows, the first yOW are outputs), the unit-stride form compact
        yc = y[:M // yOW * yW].view(M // yOW, yW, -1)[:, :yOW].reshape(M, -1) if yOW else y[:M]
        got = yc[:, :Cout].reshape(B, OH, OW, Cout).permute(0, 3, 1, 2).cpu()
        assert (got - ref.detach()).abs().max().item() < 2e-3 * max(1.0, ref.detach().abs().max().item())
        assert bool((yc[:, Cout:] == 0).all())
        Dd, dview, (dW, dOW) = Fn._window_dy(geo, Cout, tag, D)
        assert dOW == OW
        dview.view(B * OH, dW, Cout)[:, :OW] = nhwc(dy).reshape(B * OH, OW, Cout).to(D, torch.bfloat16)
        bg = torch.zeros(Cout, device=D)
        dw, dx = Fn._conv_window_bwd(Dd, A, wd, bg, geo, tag, SW == 1)
        assert tuple(dw.shape) == tuple(w.shape)
        assert (dw.cpu() - wr.grad).abs().max().item() < 3e-3 * wr.grad.abs().max().item()
        assert (bg.cpu() - br.grad).abs().max().item() < 1e-3 * br.grad.abs().max().item()
        if SW == 1:
            dxr = nhwc(xr.grad)
            assert (dx.float().cpu() - dxr).abs().max().item() < 1.2e-2 * dxr.abs().max().item()     # dx is stored in bf16
    finally:
        ops.set_compute_dtype(prev)


@pytest.mark.parametrize("grid", [(0, 0), (13, 9)])
def test_batchnorm_hardtanh_on_a_row_grid(ops, grid):
    """asr_bn_batch_stats / asr_bn_act_fwd / asr_bn_act_bwd against torch's BatchNorm2d (training mode: batch statistics, running buffers with
    the unbiased variance, reference transformer.py:35-36) + Hardtanh(0, 20), with the convolution output y either compact or on the row
    grid of a window GEMM (groups of 13 rows of which 9 are outputs; the rows in between hold garbage that must not be read) and the
    gradient written through the same kind of grid into a buffer whose other rows must stay untouched."""
    D = dev()
    C, groups = 32, 57
    gW, gOW = grid
    M = groups * (gOW or 9)
    g = torch.Generator().manual_seed(4)
    yc = (torch.randn(M, C, generator=g) * 6 + 5)                                     # compact conv output, some of it outside (0, 20)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dz = torch.randn(M, C, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
        bn.running_mean.normal_(generator=g); bn.running_var.uniform_(0.5, 2.0, generator=g)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    yr = yc.clone().requires_grad_()
    ref = F.hardtanh(bn(yr.t().reshape(1, C, M, 1)), 0.0, 20.0).reshape(C, M).t()
    ref.backward(dz)
    if gOW:
        y = torch.full((groups * gW, 64), 1e30)
        y.view(groups, gW, 64)[:, :gOW, :C] = yc.view(groups, gOW, C)
    else:
        y = torch.zeros(M, 64)
        y[:, :C] = yc
    y = y.to(D)
    rm, rv, nb = rm0.clone().to(D), rv0.clone().to(D), torch.zeros((), dtype=torch.int64, device=D)
    mean, rstd = ops.bn_train_stats(y, M, C, bn.eps, 0.1, rm, rv, nb, ygrid=grid)
    assert (mean.cpu() - yc.mean(0)).abs().max().item() < 1e-4
    assert (rstd.cpu() - torch.rsqrt(yc.var(0, unbiased=False) + bn.eps)).abs().max().item() < 1e-4
    assert (rm.cpu() - bn.running_mean).abs().max().item() < 1e-5 and (rv.cpu() - bn.running_var).abs().max().item() < 1e-4 and int(nb) == 1
    out = torch.empty(M, C, device=D)
    ops.bn_act_fwd(y, M, C, mean, rstd, gamma.to(D), beta.to(D), 0.0, 20.0, out, ygrid=grid)
    assert (out.cpu() - ref.detach()).abs().max().item() < 1e-4
    dgrid = (17, gOW) if gOW else (0, 0)                                                # the gradient goes to yet another grid
    dy = torch.full(((groups * 17 if gOW else M), C), 7.0, device=D)
    sums = ops.bn_act_bwd(dz.to(D), y, M, C, mean, rstd, gamma.to(D), beta.to(D), 0.0, 20.0, dy, ygrid=grid, dygrid=dgrid)
    assert (sums[:C].cpu() - bn.bias.grad).abs().max().item() < 2e-3 and (sums[C:].cpu() - bn.weight.grad).abs().max().item() < 2e-3
    got = dy.view(groups, 17, C)[:, :gOW].reshape(M, C) if gOW else dy
    assert (got.cpu() - yr.grad).abs().max().item() < 1e-4 * max(1.0, yr.grad.abs().max().item())
    if gOW:
        assert bool((dy.view(groups, 17, C)[:, gOW:] == 7.0).all())                  # rows between the groups: not written


@pytest.mark.parametrize("grid", [(0, 0), (13, 9)])
@pytest.mark.parametrize("nv", [9, 6, 1])
def test_batchnorm_length_masked_statistics(ops, grid, nv):
    """The *_v forms (round 6): rows are (group, t) with t = m % 9; only t < valid[0] -- a DEVICE int -- take part in the statistics and get
    a gradient.  Truth: torch's BatchNorm2d (training) + Hardtanh on the COMPACTED valid rows -- what the reference's BatchNorm sees when
    the batch is not padded to a shape bucket -- and zeros in the gradient of the masked rows.  nv = 9: everything valid = the unmasked
    result."""
    D = dev()
    C, groups, Wt = 32, 41, 9
    gW, gOW = grid
    M = groups * Wt
    g = torch.Generator().manual_seed(40 + nv)
    yc = (torch.randn(M, C, generator=g) * 6 + 5)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dz = torch.randn(M, C, generator=g)
    keep = (torch.arange(M) % Wt) < nv
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    yv = yc[keep].clone().requires_grad_()
    Mv = int(keep.sum())
    ref = F.hardtanh(bn(yv.t().reshape(1, C, Mv, 1)), 0.0, 20.0).reshape(C, Mv).t()
    ref.backward(dz[keep])
    if gOW:
        y = torch.full((groups * gW, 64), 1e30)
        y.view(groups, gW, 64)[:, :gOW, :C] = yc.view(groups, gOW, C)
    else:
        y = torch.zeros(M, 64)
        y[:, :C] = yc
    y = y.to(D)
    valid = (torch.tensor([nv], dtype=torch.int32, device=D), Wt)
    rm, rv, nb = rm0.clone().to(D), rv0.clone().to(D), torch.zeros((), dtype=torch.int64, device=D)
    mean, rstd = ops.bn_train_stats(y, M, C, bn.eps, 0.1, rm, rv, nb, ygrid=grid, valid=valid)
    assert (mean.cpu() - yc[keep].mean(0)).abs().max().item() < 1e-4
    assert (rstd.cpu() - torch.rsqrt(yc[keep].var(0, unbiased=False) + bn.eps)).abs().max().item() < 1e-4
    assert (rm.cpu() - bn.running_mean).abs().max().item() < 1e-5 and (rv.cpu() - bn.running_var).abs().max().item() < 1e-4 and int(nb) == 1
    dgrid = (17, gOW) if gOW else (0, 0)
    dy = torch.full(((groups * 17 if gOW else M), C), 7.0, device=D)
    sums = ops.bn_act_bwd(dz.to(D), y, M, C, mean, rstd, gamma.to(D), beta.to(D), 0.0, 20.0, dy, ygrid=grid, dygrid=dgrid, valid=valid)
    assert (sums[:C].cpu() - bn.bias.grad).abs().max().item() < 2e-3 and (sums[C:].cpu() - bn.weight.grad).abs().max().item() < 2e-3
    got = (dy.view(groups, 17, C)[:, :gOW].reshape(M, C) if gOW else dy).cpu()
    assert (got[keep] - yv.grad).abs().max().item() < 1e-4 * max(1.0, yv.grad.abs().max().item())
    assert bool((got[~keep] == 0).all())                                              # masked rows: no gradient


def test_ops_refuse_host_tensors(ops):
    from asr_hip.lib import AsrHipError
    with pytest.raises(AsrHipError):
        ops.gemm_nt(torch.zeros(8, 8), torch.zeros(8, 8))


# ------------------------------------------------------------------------------------------------ spectrogram front end
def test_gpu_spectrogram_matches_host_convention(ops):
    """asr_stft_frames + fp32 MFMA DFT + asr_spect_finish against utils/audio.log_spectrogram (the numpy restatement of
    the reference's librosa/scipy convention, data_loader.py:72-89) on ragged utterances, including very short ones.
    Tolerance: fp32 DFT by GEMM vs numpy's float64-internally rfft: 2e-4 absolute on the normalised features (O(1))."""
    from utils.audio import log_spectrogram
    rng = np.random.RandomState(3)
    lens = [16000, 12345, 4001, 700, 161, 100, 1]
    Lmax = max(lens)
    wav = np.zeros((len(lens), Lmax), dtype=np.float32)
    for i, n in enumerate(lens):
        t = np.arange(n) / 16000.0
        wav[i, :n] = (0.3 * np.sin(2 * np.pi * (200 + 37 * i) * t) + 0.05 * rng.randn(n)).astype(np.float32)
    D = dev()
    spect, nfr = ops.log_spectrogram(torch.from_numpy(wav).to(D), torch.tensor(lens, dtype=torch.int32, device=D))
    assert spect.shape == (len(lens), 1, 161, 1 + Lmax // 160)
    sp = spect.cpu().numpy()
    for i, n in enumerate(lens):
        ref = log_spectrogram(wav[i, :n])
        T = ref.shape[1]
        assert int(nfr[i]) == T
        if n > 1:       # a 1-sample utterance has 1 frame x 161 bins of a constant: std of near-equal values, skip the values
            np.testing.assert_allclose(sp[i, 0, :, :T], ref, rtol=0, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=str(n))
        assert (sp[i, 0, :, T:] == 0).all()
    raw, _ = ops.log_spectrogram(torch.from_numpy(wav[:2]).to(D), torch.tensor(lens[:2], dtype=torch.int32, device=D), normalize=False)
    ref = log_spectrogram(wav[1, :lens[1]], normalize=False)
    np.testing.assert_allclose(raw[1, 0, :, :ref.shape[1]].cpu().numpy(), ref, rtol=0, atol=1e-4)


def test_gpu_spectrogram_matches_scipy_restatement_and_known_answer(ops):
    """The GPU front end against the two independent pins of the convention (tests/test_host.py): the scipy.signal.stft restatement of
    SpectrogramParser.parse_audio (reference utils/data_loader.py:72-89) on ragged noise, and the hand-computed impulse spectrogram
    (symmetric Hamming w[k] = 0.54 - 0.46 cos(2 pi k / 319), reflect centring, hop 160, log1p) -- no FFT library on that side."""
    import test_host as TH
    rng = np.random.RandomState(11)
    lens = [4807, 16000, 321]
    wav = np.zeros((4, max(lens)), dtype=np.float32)
    for i, n in enumerate(lens):
        wav[i, :n] = (0.1 * rng.randn(n)).astype(np.float32)
    y, exp = TH.spectrogram_known_answer()
    wav[3, :y.size] = y
    lens.append(int(y.size))
    D = dev()
    for norm in (False, True):
        sp, nfr = ops.log_spectrogram(torch.from_numpy(wav).to(D), torch.tensor(lens, dtype=torch.int32, device=D), normalize=norm)
        sp = sp.cpu().numpy()
        for i, n in enumerate(lens[:3]):
            ref = TH._scipy_spectrogram(wav[i, :n], norm)
            assert int(nfr[i]) == ref.shape[1] == 1 + n // 160
            np.testing.assert_allclose(sp[i, 0, :, :ref.shape[1]], ref, rtol=0, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg="%d %s" % (n, norm))
        assert int(nfr[3]) == 3
        if not norm:
            np.testing.assert_allclose(sp[3, 0, :, :3], exp, rtol=0, atol=2e-5)
        else:
            mean = exp.sum() / exp.size
            std = np.sqrt(((exp - mean) ** 2).sum() / (exp.size - 1))
            np.testing.assert_allclose(sp[3, 0, :, :3], (exp - mean) / std, rtol=0, atol=2e-4)


def test_device_prefetcher_delivers_device_batches():
    from utils.data_loader import DevicePrefetcher
    batches = [(torch.randn(4, 1, 161, 50 + i), torch.randint(0, 9, (4, 7)), torch.ones(4), torch.full((4,), 50 + i), None) for i in range(6)]
    out = []
    for b in DevicePrefetcher(batches, device=dev()):
        assert b[0].is_cuda and b[1].is_cuda and not b[2].is_cuda
        out.append((b[0] * 2).sum().item())          # consume on the compute stream
    assert len(out) == 6
    for i, v in enumerate(out):
        assert abs(v - float((batches[i][0] * 2).sum())) < 1e-2 * max(1.0, abs(v))


@pytest.mark.parametrize("M,V,k", [(5, 4364, 4), (1, 35, 2), (32, 4364, 8), (3, 17, 16)])
def test_logsoftmax_topk_matches_torch(M, V, k):
    """asr_logsoftmax_topk against the reference's two calls (F.log_softmax + torch.topk, transformer.py:446-449)."""
    from asr_hip import ops
    g = torch.Generator().manual_seed(M * 7 + V)
    x = (3 * torch.randn(M, V + 5, generator=g)).cuda()[:, :V]          # a row stride larger than V
    x[0, 3] = x[0, 11] = x.max() + 1.0                                  # a tie at the top: lowest index first
    vals, idx = ops.logsoftmax_topk(x, k)
    ref = torch.log_softmax(x.float().cpu(), dim=1)
    rv, ri = torch.topk(ref, k, dim=1)
    assert (vals.cpu() - rv).abs().max().item() <= 2e-6 * max(1.0, rv.abs().max().item())
    assert idx[0, 0].item() == 3 and idx[0, 1].item() == 11
    # same index sets, and the same order wherever the values are distinct
    assert torch.equal(torch.sort(idx.cpu(), 1).values, torch.sort(ri, 1).values)
    distinct = (rv[:, 1:] != rv[:, :-1]).all(1)
    assert torch.equal(idx.cpu()[distinct], ri[distinct])
