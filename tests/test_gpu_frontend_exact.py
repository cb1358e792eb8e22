"""The whole vgg_cnn front end (reference: models/asr/transformer.py:42-52, :70-76 and its autograd), benched kernels against the
stored-activation launch chain, BIT FOR BIT (VERDICT r5 #6a).

Every fp32-tight parity statement of tests/test_gpu_baseline_shapes.py runs the front end with the activation tap ON, i.e. on the chain
conv1_fwd -> conv.2 (+ pool codes) -> conv.5 -> conv.7 -> pooling kernel.  The benched step runs other kernels: conv_level0.hip (conv.0
recomputed, nothing stored at full resolution), conv_ws.hip (conv.5 in one pass writing its ReLU mask as bits, conv.7 with its pooled
epilogue, the data gradients from the bit mask).  The two were tied per kernel (tests/test_gpu_level0.py, tests/test_gpu_conv_ws.py)
and, for the whole model, only to bf16 noise.  Here they are tied end to end on data for which the summation order cannot matter:

  * frames, all four conv weights / biases and the incoming gradient are small INTEGERS; a bf16 rounding of an integer is an integer, a
    product of two bf16 values is exact in fp32, so every partial sum of every layer -- forward, data gradients, weight gradients -- is
    an integer; with sparse weights they stay below 2^24, where fp32 addition is exact and therefore associative;
  * hence both chains must produce the SAME BITS: `torch.equal` on the features `Transformer._features` returns and on all eight conv
    parameter gradients, tap on vs tap off;
  * the premise itself is checked: the forward features equal a float64 torch restatement (conv2d / ReLU / max_pool2d with a bf16
    rounding wherever the kernels store bf16) exactly, and the largest |partial-sum bound| of every layer is asserted below 2^24.

Shapes: odd height like the benchmark's 161 rows (H % 32 == 1: the pooled second level has H2 % 16 == 0 -> the vertical tile pairs of
the benched conv.7 epilogue), widths with W % 32 == 0 (the benched path's own requirement for the pooled epilogue).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
D = "cuda:0"
BF = torch.bfloat16


def _sparse_int(g, shape, density, lo=-1, hi=1):
    """Integer tensor in [lo, hi], zero except for a `density` fraction of entries."""
    w = torch.randint(lo, hi + 1, shape, generator=g).float()
    keep = torch.rand(shape, generator=g) < density
    return w * keep


def _build(seed, B, H, W):
    from utils import constant
    from utils.functions import init_transformer_model
    g = torch.Generator().manual_seed(seed)
    args = constant.parse(["--num-layers", "1", "--num-heads", "2", "--dim-model", "128", "--dim-key", "64", "--dim-value", "64", "--dim-inner",
                           "128", "--dim-emb", "128", "--feat_extractor", "vgg_cnn", "--tgt-max-len", "16", "--src-max-len", str(W),
                           "--dropout", "0.0", "--precision", "bf16", "--cuda"])          # (only Transformer._features is exercised: the encoder's input width is irrelevant)
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + list("abcdefgh")
    l2i = {c: i for i, c in enumerate(chars)}
    model = init_transformer_model(args, l2i, {i: c for c, i in l2i.items()})
    core = model.module if hasattr(model, "module") else model
    conv = core.conv
    ws = {"0": (64, 1, 0.6, 2), "2": (64, 64, 0.04, 1), "5": (128, 64, 0.04, 1), "7": (128, 128, 0.02, 1)}
    with torch.no_grad():
        for idx, (co, ci, dens, amp) in ws.items():
            layer = conv[int(idx)]
            layer.weight.copy_(_sparse_int(g, (co, ci, 3, 3), dens, -amp, amp))
            layer.bias.copy_(torch.randint(-1, 2, (co,), generator=g).float())
    src = torch.randint(-3, 4, (B, 1, H, W), generator=g).float()
    return model, core, src, g


def _reference_features(core, src):
    """float64 restatement with the kernels' bf16 storage points: y1, pooled p1, y3, pooled output."""
    c = core.conv
    w = [c[i].weight.detach().double().cpu() for i in (0, 2, 5, 7)]
    b = [c[i].bias.detach().double().cpu() for i in (0, 2, 5, 7)]
    rb = lambda t: t.to(BF).double()
    bound = []
    x = src.double()
    y1 = rb(F.relu(F.conv2d(x, w[0], b[0], padding=1)))
    bound.append(float(F.conv2d(x.abs(), w[0].abs(), b[0].abs(), padding=1).max()))
    bound.append(float(F.conv2d(y1, w[1].abs(), b[1].abs(), padding=1).max()))
    p1 = rb(F.max_pool2d(F.relu(F.conv2d(y1, w[1], b[1], padding=1)), 2, 2))
    bound.append(float(F.conv2d(p1, w[2].abs(), b[2].abs(), padding=1).max()))
    y3 = rb(F.relu(F.conv2d(p1, w[2], b[2], padding=1)))
    bound.append(float(F.conv2d(y3, w[3].abs(), b[3].abs(), padding=1).max()))
    y4 = rb(F.max_pool2d(F.relu(F.conv2d(y3, w[3], b[3], padding=1)), 2, 2))
    Bn, C, Hh, Ww = y4.shape
    feats = y4.reshape(Bn, C * Hh, Ww).transpose(1, 2).contiguous()         # transformer.py:74-76
    return feats, bound


@pytest.mark.parametrize("shape", [(2, 33, 64), (1, 161, 96), (3, 65, 160)])
def test_front_end_benched_kernels_equal_the_stored_activation_chain_bit_for_bit(shape):
    from asr_hip import functions as F_
    from asr_hip import params as P
    from utils.functions import init_optimizer
    from utils import constant
    B, H, W = shape
    runs = {}
    feats_ref = None
    for tap in (True, False):
        model, core, src, g = _build(1000 + H + W, B, H, W)
        model = model.cuda().train()
        opt = init_optimizer(constant.args, model, "noam")       # flat fp32 gradient buffers (the kernels accumulate into them)
        opt.zero_grad()
        if feats_ref is None:
            feats_ref, bound = _reference_features(core, src)
            assert max(bound) < 2 ** 24, ("a partial sum may leave fp32's exact-integer range", bound)
        Tp, Din = W // 4, 128 * ((H // 2) // 2)
        dout = torch.randint(-2, 3, (B, Tp, Din), generator=g).to(BF).to(D)
        F_.capture_selections = [] if tap else None
        try:
            feats = core._features(src.to(D))
            assert feats.shape == (B, Tp, Din) and feats.dtype == BF
            feats.backward(dout)
            torch.cuda.synchronize()
        finally:
            taps, F_.capture_selections = F_.capture_selections, None
        if tap:
            assert any(kind == "vgg" for kind, _ in taps), "tap on: the stored-activation chain"
        grads = {"conv.%d.%s" % (i, n): getattr(core.conv[i], n).grad.detach().float().cpu().clone() for i in (0, 2, 5, 7) for n in ("weight", "bias")}
        runs[tap] = (feats.detach().float().cpu(), grads)
    f1, g1 = runs[True]
    f0, g0 = runs[False]
    # the premise: both equal the exact (float64) result rounded where the kernels round
    assert torch.equal(f1.double(), feats_ref), float((f1.double() - feats_ref).abs().max())
    assert float(f1.abs().max()) > 0 and float((f1 != 0).float().mean()) > 0.02, "degenerate features: the test would prove nothing"
    # the statement: benched kernels == stored-activation chain, bit for bit
    assert torch.equal(f0, f1), float((f0 - f1).abs().max())
    for k in g1:
        assert float(g1[k].abs().max()) > 0, (k, "zero gradient: nothing compared")
        assert torch.equal(g0[k], g1[k]), (k, float((g0[k] - g1[k]).abs().max()), float(g1[k].abs().max()))
