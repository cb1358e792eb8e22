"""The second max-pool's backward inside the encoder input projection's data gradient (round 6; reference: the autograd of
models/asr/transformer.py:50-52 MaxPool2d, :74-76 view / transpose, :172 encoder.input_linear).

Three launches became one: asr_gemm_nn (the data gradient into the pooled tensor's (B, W/2, C, H/2) layout) + asr_maxpool_bwd_code
(expansion through the selection bytes into the gradient of conv.7's un-pooled output) -> asr_gemm_nn_poolbwd, the same GEMM on a
column-permuted weight with the expansion in its epilogue.  Everything here is EQUALITY: the products, their summation order and the
single bf16 rounding are those of asr_gemm_nn (same kernel, same tiles); the rest is a permutation and a selection.

  * asr_permute_cols_tcf against torch indexing;
  * conv.7's pooled epilogue with channel-last selection bytes against the (B, W/2, C, H/2) form it replaces (pooled values untouched);
  * asr_gemm_nn_poolbwd against asr_gemm_nn + asr_maxpool_bwd_code, incl. code 0 ("no gradient": a pooled value of zero) and ragged M;
  * the whole model step (vgg_tiny, bf16) with the hand-over on and off: every gradient equal bit for bit.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = "cuda:0"
BF = torch.bfloat16


@pytest.mark.parametrize("rows,C,H2", [(512, 128, 40), (7, 16, 3), (64, 128, 8)])
@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_permute_cols_tcf(rows, C, H2, dtype):
    from asr_hip import ops
    g = torch.Generator().manual_seed(rows + C)
    src = torch.randn(rows, C * H2, generator=g).to(dtype).to(D)
    dst = torch.empty_like(src)
    ops.permute_cols_tcf(src, dst, C, H2)
    want = src.view(rows, C, H2).permute(0, 2, 1).reshape(rows, H2 * C)
    assert torch.equal(dst, want)


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 80, 48), (3, 24, 16)])
def test_pooled_epilogue_channel_last_codes(B, H, W):
    from asr_hip import ops
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randint(-3, 4, (B, H, W, 128), generator=g).float().to(D).to(BF)
    w = torch.randint(-2, 3, (128, 128, 3, 3), generator=g).float()
    wk = w.permute(0, 2, 3, 1).reshape(128, 9, 128).contiguous().to(D).to(BF)
    bias = (torch.randint(-8, 9, (128,), generator=g).float() * 0.5).to(D)
    a = ops.conv3x3_relu_pool_tcf_code(x, wk, bias, 128)
    b = ops.conv3x3_relu_pool_tcf_code(x, wk, bias, 128, code_cl=True)
    assert a is not None and b is not None
    assert torch.equal(a[0], b[0])
    H2, W2 = H // 2, W // 2
    assert tuple(b[1].shape) == (B, W2, H2, 128)
    assert torch.equal(b[1].permute(0, 1, 3, 2).reshape(B, W2, 128 * H2), a[1])
    assert int(a[1].max()) <= 4 and int((a[1] == 0).sum()) > 0 and int((a[1] == 4).sum()) > 0      # every kind of byte is exercised


@pytest.mark.parametrize("B,H,W,K", [(2, 16, 32, 128), (32, 80, 400, 512), (3, 16, 16, 64), (5, 32, 48, 256)])
def test_gemm_nn_poolbwd_equals_gemm_then_pool_backward(B, H, W, K):
    from asr_hip import ops
    C, H2, W2 = 128, H // 2, W // 2
    M, N = B * W2, C * H2
    g = torch.Generator().manual_seed(B * 7 + K)
    dy = torch.randn(M, K, generator=g).to(BF).to(D)
    wgt = (torch.randn(K, N, generator=g) / 8).to(BF).to(D)                  # (out_features, in_features): columns in c * H2 + h2 order
    code = torch.randint(0, 5, (B, W2, C * H2), generator=g, dtype=torch.uint8).to(D)     # pooled layout (B, W2, C, H2)
    # the two launches (GEMM_BIG_NN = 2: the eight-wave kernel at every size -- the fused form has no other, and "same bits" is a
    # statement about one kernel's summation order)
    from asr_hip import lib as L
    try:
        L.set_tuning("GEMM_BIG_NN", 2)
        d_pool = ops.gemm_nn(dy, wgt)
    finally:
        L.set_tuning("GEMM_BIG_NN", None)
    assert d_pool.dtype == BF and tuple(d_pool.shape) == (M, N)
    want = ops.maxpool_bwd_code(code, d_pool.view(B, W2, N).contiguous(), (B, H, W, C), tcf=True)
    # the one launch
    w_perm = torch.empty_like(wgt)
    ops.permute_cols_tcf(wgt, w_perm, C, H2)
    code_cl = code.view(B, W2, C, H2).permute(0, 1, 3, 2).contiguous()
    got = ops.gemm_nn_poolbwd(dy, w_perm, code_cl, (B, H, W, C))
    assert got is not None, "the library is expected to take this shape"
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert float(want.float().abs().max()) > 0


def test_whole_step_with_and_without_the_hand_over(golden_dir):
    from test_gpu_model import build
    from asr_hip import functions as F_
    from utils.metrics import calculate_loss
    grads = {}
    for on in (True, False):
        z, args, model, opt = build(golden_dir, "vgg_tiny", "bf16")
        src, tgt = torch.from_numpy(z["src"]).to(D), torch.from_numpy(z["tgt"]).to(D)
        src_len = torch.from_numpy(z["src_len"])
        old = F_._pool_handover_on
        F_._pool_handover_on = on
        try:
            opt.zero_grad()
            pred, gold, _, _ = model(src, src_len, tgt)
            loss = calculate_loss(pred, gold, smoothing=float(z["smoothing"]))
            loss.backward()
            torch.cuda.synchronize()
        finally:
            F_._pool_handover_on = old
        grads[on] = (float(loss.item()), {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()})
    (l1, g1), (l0, g0) = grads[True], grads[False]
    # (bit for bit since the loss statistics are added in a fixed order, asr_ce_finish: with fp32 atomics one ulp was seen once in ~15 runs)
    assert l1 == l0, (l1, l0)
    for k in g1:
        if k.startswith("conv.") and k.endswith(".bias"):
            # the conv bias gradients are summed with fp32 atomics (csrc/conv.hip, conv_wgrad_dma.hip): equal up to their order, run to run
            assert torch.allclose(g1[k], g0[k], rtol=1e-5, atol=1e-7), (k, float((g1[k] - g0[k]).abs().max()))
            continue
        assert torch.equal(g1[k], g0[k]), (k, float((g1[k] - g0[k]).abs().max()))
    assert float(g1["conv.7.weight"].abs().max()) > 0 and float(g1["conv.0.weight"].abs().max()) > 0
