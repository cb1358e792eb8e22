"""The full-resolution level of vgg_cnn without its full-resolution activations (csrc/conv_level0.hip; reference:
models/asr/transformer.py:42-47 and its autograd): forward, data-side backward (dW0 / db0) and weight-side backward (dW2 / db2)

  * against the launch chain they replace -- asr_conv1_fwd -> asr_conv3x3_relu_pool_code; asr_maxpool_bwd_code -> asr_conv3x3_igemm
    with the ReLU mask -> asr_conv1_wgrad; asr_conv3x3_wgrad_nhwc -- on inputs for which conv.0 is EXACT in both formulations (small
    integer frames and weights: every partial sum is an integer below 256), where pooled values, selection codes and conv.2's weight
    gradient must agree bit for bit and the first layer's gradient to fp32 summation order;
  * against torch's fp32 convolutions on random inputs, with the bf16 tolerance of the other convolution tests.
Shapes: odd height (the benchmark's 161 rows: the last tile row holds one image row, the last image row is not pooled), widths that
are not a multiple of the 16-pixel tile, several tiles per workgroup, a single tile.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
D = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from asr_hip import ops as o
    return o


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _packed(ops, w):
    wk = torch.empty(64, 9, 64, device=D, dtype=BF)
    wd = torch.empty(64, 9, 64, device=D, dtype=BF)
    ops.conv_pack_weight(w.to(D), wk, wd)
    return wk, wd


def _inputs(shape, exact, seed):
    B, H, W = shape
    g = torch.Generator().manual_seed(seed)
    if exact:
        src = torch.randint(-4, 5, (B, 1, H, W), generator=g).float()
        w0 = torch.randint(-2, 3, (64, 1, 3, 3), generator=g).float()
        b0 = torch.randint(-3, 4, (64,), generator=g).float()
    else:
        src = torch.randn(B, 1, H, W, generator=g)
        w0 = torch.randn(64, 1, 3, 3, generator=g) / 3
        b0 = torch.randn(64, generator=g) / 3
    w2 = torch.randn(64, 64, 3, 3, generator=g) / 24
    b2 = torch.randn(64, generator=g) / 3
    dp = torch.randn(B, H // 2, W // 2, 64, generator=g).to(BF)
    return src, w0, b0, w2, b2, dp


SHAPES = [(2, 21, 37), (3, 161, 232), (1, 8, 16), (2, 9, 50), (1, 40, 160), (5, 97, 401)]


@pytest.mark.parametrize("shape", SHAPES)
def test_level0_equals_the_launch_chain_it_replaces(ops, shape):
    B, H, W = shape
    src, w0, b0, w2, b2, dp = _inputs(shape, True, H * 13 + W)
    wk, wd = _packed(ops, w2)
    sd, w0d, b0d, b2d, dpd = src.to(D), w0.to(D), b0.to(D), b2.to(D), dp.to(D)
    # ---- forward
    y1 = ops.conv1_fwd(sd, w0d, b0d, BF)
    _, p_ref, c_ref = ops.conv3x3_relu_pool_code(y1, wk, b2d, 64)
    out = ops.vgg_level0_fwd(sd, w0d, b0d, wk, b2d)
    assert out is not None
    pool, code = out
    assert torch.equal(pool, p_ref), float((pool.float() - p_ref.float()).abs().max())
    assert torch.equal(code, c_ref)
    # ---- backward, weight side of conv.2
    dy2 = ops.maxpool_bwd_code(c_ref, dpd, (B, H, W, 64))
    dw2_ref = torch.zeros(64, 64, 3, 3, device=D); db2_ref = torch.zeros(64, device=D)
    ops.conv3x3_wgrad_nhwc(y1, dy2, dw2_ref, db2_ref)
    dw2 = torch.zeros(64, 64, 3, 3, device=D); db2 = torch.zeros(64, device=D)
    ops.vgg_level0_wgrad(sd, w0d, b0d, dpd, code, dw2, db2)
    assert torch.equal(dw2, dw2_ref), float((dw2 - dw2_ref).abs().max())
    assert float((db2 - db2_ref).abs().max()) <= 1e-4 * max(1.0, float(db2_ref.abs().max()))       # (atomics: order only)
    # ---- backward, data side: dW0 / db0
    dy1 = ops.conv3x3(dy2, wd, None, 64, relu=False, mask_src=y1)
    dw0_ref = torch.zeros(64, 1, 3, 3, device=D); db0_ref = torch.zeros(64, device=D)
    ops.conv1_wgrad(sd, dy1, dw0_ref, db0_ref)
    dw0 = torch.zeros(64, 1, 3, 3, device=D); db0 = torch.zeros(64, device=D)
    ops.vgg_level0_dgrad(dpd, code, sd, w0d, b0d, wd, dw0, db0)
    sc = max(1.0, float(dw0_ref.abs().max()))
    assert float((dw0 - dw0_ref).abs().max()) <= 2e-5 * sc * math.sqrt(B * H * W / 1000 + 1), (float((dw0 - dw0_ref).abs().max()), sc)
    assert float((db0 - db0_ref).abs().max()) <= 2e-5 * max(1.0, float(db0_ref.abs().max())) * math.sqrt(B * H * W / 1000 + 1)
    # accumulation into existing gradients (+=), second call
    ops.vgg_level0_dgrad(dpd, code, sd, w0d, b0d, wd, dw0, db0)
    assert float((dw0 - 2 * dw0_ref).abs().max()) <= 4e-5 * sc * math.sqrt(B * H * W / 1000 + 1)


@pytest.mark.parametrize("shape", [(2, 21, 37), (2, 161, 120)])
@pytest.mark.parametrize("wsplit", ["1", "0"])
def test_level0_against_torch_convolutions(ops, shape, wsplit):
    """Random inputs: the whole level against F.conv2d / max_pool2d in fp32 on the bf16-rounded conv.2 operands.  wsplit = 0 runs
    conv.0 on its bf16-rounded weights alone (the wlo slots of the one MFMA left zero): conv.0's pre-activations then move by 2^-9 relative, ~0.3 % of
    its ReLU decisions differ from the fp32 reference's and the gradients -- compared under the REFERENCE's selections here -- by ~5 %
    (measured 4.5 - 6.7 %); the split weights (default) keep the decisions and stay at the bf16 floor of 3 %."""
    from asr_hip import lib as L
    L.set_tuning("L0_WSPLIT", int(wsplit))
    try:
        B, H, W = shape
        src, w0, b0, w2, b2, dp = _inputs(shape, False, H * 3 + W)
        wk, wd = _packed(ops, w2)
        sd, w0d, b0d, b2d, dpd = src.to(D), w0.to(D), b0.to(D), b2.to(D), dp.to(D)
        w0r, b0r = w0.clone().requires_grad_(), b0.clone().requires_grad_()
        w2r, b2r = w2.to(BF).float().requires_grad_(), b2.clone().requires_grad_()
        y1 = F.relu(F.conv2d(src, w0r, b0r, padding=1))
        y1q = y1 + (y1.detach().to(BF).float() - y1.detach())                     # bf16 storage, straight-through
        y2 = F.relu(F.conv2d(y1q, w2r, b2r, padding=1))
        y2q = y2 + (y2.detach().to(BF).float() - y2.detach())
        p_ref = F.max_pool2d(y2q, 2, 2)
        pool, code = ops.vgg_level0_fwd(sd, w0d, b0d, wk, b2d)
        pf = pool.float().cpu()
        ref = nhwc(p_ref.detach())
        err = float((pf - ref).abs().max())
        assert err <= 2.5e-2 * float(ref.abs().max()), (err, float(ref.abs().max()))
        assert torch.equal((code == 0).cpu(), pf == 0) and int(code.max()) <= 4
        p_ref.backward(nchw(dp.float()))
        dw2 = torch.zeros(64, 64, 3, 3, device=D); db2 = torch.zeros(64, device=D)
        ops.vgg_level0_wgrad(sd, w0d, b0d, dpd, code, dw2, db2)
        dw0 = torch.zeros(64, 1, 3, 3, device=D); db0 = torch.zeros(64, device=D)
        ops.vgg_level0_dgrad(dpd, code, sd, w0d, b0d, wd, dw0, db0)
        rel = lambda a, r: float((a.float().cpu() - r).norm() / (r.norm() + 1e-30))
        e = dict(dw2=rel(dw2, w2r.grad), db2=rel(db2, b2r.grad), dw0=rel(dw0, w0r.grad), db0=rel(db0, b0r.grad))
        print("level0 vs torch fp32 (relative L2):", e)
        assert max(e.values()) <= (3e-2 if wsplit == "1" else 1e-1), e
    finally:
        L.set_tuning("L0_WSPLIT", None)
