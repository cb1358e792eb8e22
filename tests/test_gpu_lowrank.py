"""Low-Rank Transformer variant (BASELINE configs[4]; arXiv:1910.13923 cited by the reference README, no
code in the reference tree: PARITY UNPINNED).  The product model built with --rank r against the oracle's restatement
(oracle/asr_oracle.py:proj -- y = V (U x) + b for every attention / feed-forward projection) from the same state_dict:
logits, loss and every gradient in fp32 mode; bf16 mode within the bf16 bounds of the full-rank model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLAGS = ["--num-layers", "2", "--num-heads", "4", "--dim-model", "128", "--dim-key", "32", "--dim-value", "32", "--dim-inner", "256",
         "--dim-emb", "128", "--feat_extractor", "vgg_cnn", "--tgt-max-len", "24", "--src-max-len", "64", "--label-smoothing", "0.1",
         "--dropout", "0.0", "--rank", "16"]


def _build(precision):
    from utils import constant
    from utils.functions import init_transformer_model
    args = constant.parse(FLAGS + ["--precision", precision, "--cuda"])
    V = 40
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x4E00 + i) for i in range(V - 3)]
    l2i = {c: i for i, c in enumerate(chars)}
    torch.manual_seed(7)
    model = init_transformer_model(args, l2i, {i: c for c, i in l2i.items()})
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in sorted(model.named_parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return args, model, V


def _batch(V):
    g = torch.Generator().manual_seed(11)
    src = torch.randn(3, 1, 161, 64, generator=g)
    src_len = torch.tensor([64, 48, 20], dtype=torch.int32)
    for b in range(3):
        src[b, :, :, int(src_len[b]):] = 0
    tgt = torch.zeros(3, 20, dtype=torch.int64)
    for b, L in enumerate([20, 11, 4]):
        tgt[b, :L] = torch.randint(3, V, (L,), generator=g)
    return src, src_len, tgt


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_lowrank_model_matches_oracle(precision):
    from oracle import asr_oracle as O
    from utils.functions import init_optimizer
    from utils.metrics import calculate_metrics
    args, model, V = _build(precision)
    names = [k for k in model.state_dict() if ".u.weight" in k]
    assert len(names) == 2 * (4 + 2) + 2 * (4 + 4 + 2) and model.encoder.layers[0].self_attn.query_linear.u.weight.shape == (16, 128)
    src, src_len, tgt = _batch(V)
    w = {k: v.detach().double().cpu() if v.dtype.is_floating_point else v.cpu() for k, v in model.state_dict().items()}
    cfg = O.Cfg.from_flags(" ".join(FLAGS[:-2]))
    ref = O.train_step(w, cfg, src.double(), src_len, tgt, 0.1)
    model = model.cuda().train()
    opt = init_optimizer(args, model, "noam")
    opt.zero_grad()
    pred, gold, hyp, _ = model(src.cuda(), src_len, tgt.cuda())
    loss, _ = calculate_metrics(pred, gold, smoothing=0.1, loss_type="ce")
    loss.backward()
    amax = float(ref["pred"].abs().max())
    perr = float((pred.detach().double().cpu() - ref["pred"]).abs().max())
    rel = {}
    for k, p in model.named_parameters():
        if k.endswith("key_linear.v.bias"):
            continue                                  # exact gradient is zero (softmax shift invariance)
        a, b = p.grad.detach().double().cpu().reshape(-1), ref["grads"][k].reshape(-1)
        rel[k] = float((a - b).norm() / (b.norm() + 1e-30))
    worst = max(rel, key=rel.get)
    print("low-rank %s: logits max err %.3e (max |logit| %.2f), loss err %.2e, gradient rel L2 worst %s %.3e median %.3e"
          % (precision, perr, amax, abs(loss.item() - ref["loss"]), worst, rel[worst], float(np.median(list(rel.values())))))
    if precision == "fp32":
        assert perr <= 5e-5 * max(1.0, amax) and abs(loss.item() - ref["loss"]) < 2e-5
        assert rel[worst] <= 2e-3 and float(np.median(list(rel.values()))) <= 2e-5, (worst, rel[worst])     # worst: one flipped ReLU / pool arg-max (test_gpu_baseline_shapes.py)
    else:
        assert perr <= 6e-2 * amax and abs(loss.item() - ref["loss"]) < 3e-2
        assert float(np.median(list(rel.values()))) <= 6e-2 and rel[worst] <= 0.25, (worst, rel[worst])
    opt.step()


def _e4m3_decode(b):
    """OCP e4m3fn byte -> float (CPU restatement of the format: 1-4-3, bias 7, no infinities, 0x7f/0xff = NaN)."""
    b = b.to(torch.int32)
    s = torch.where((b & 0x80) != 0, -1.0, 1.0).double()
    e = ((b >> 3) & 0xF).double()
    m = (b & 7).double()
    v = torch.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e - 7))
    return s * v


@pytest.mark.parametrize("shape", [(200, 64, 512), (6400, 512, 64), (130, 70, 128), (64, 2048, 64), (333, 64, 2048)])
def test_fp8_gemm_is_exact_on_its_quantised_operands(shape):
    """asr_quant_fp8 + asr_gemm_nt_fp8: (1) the bytes decode (OCP e4m3fn) to within half an fp8 step of x * 448 / row amax;
    (2) the GEMM equals the float64 product of the DECODED operands to 1e-4 of the largest output (measured 3e-5: the
    block-scaled matrix core aligns the 32 products of a block to their largest before adding -- about 2^-15 relative -- so it
    is not bit-exact fp32 accumulation, but any operand-layout error of v_mfma_scale_f32_16x16x128_f8f6f4 would be O(1)); (3) against the un-quantised product the
    relative L2 error is the e4m3 quantisation noise (3 mantissa bits: <= 4e-2 stated tolerance)."""
    from asr_hip import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g)
    qa, sa = ops.quant_fp8(A.cuda())
    qb, sb = ops.quant_fp8(B.cuda())
    da, db = _e4m3_decode(qa.cpu())[:, :K], _e4m3_decode(qb.cpu())[:, :K]
    ra = sa.double().cpu()[:, None]
    assert (ra[:, 0] - A.float().abs().amax(1).double() / 448).abs().max() < 1e-6 * ra.max()
    xs = A.double() / ra
    step = torch.clamp(2.0 ** (torch.floor(torch.log2(xs.abs().clamp_min(2.0 ** -6))) - 3), min=2.0 ** -9)
    assert ((da - xs).abs() <= 0.5 * step + 1e-5 * xs.abs() + 1e-9).all()          # (exact ties round to even; the scale is fp32)
    out = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=bias.cuda(), out_dtype=torch.float32, K=qa.shape[1])
    exact = (da @ db.t()) * (ra * sb.double().cpu()[None, :]) + bias.double()
    err = (out.double().cpu() - exact).abs().max().item()
    assert err <= 1e-4 * exact.abs().max().item(), err
    full = A.double() @ B.double().t() + bias.double()
    rel = float((out.double().cpu() - full).norm() / full.norm())
    print("fp8 gemm %s: rel L2 vs the unquantised product %.3e" % (shape, rel))
    assert rel <= 3.2e-2, rel            # measured 2.5e-2 .. 2.7e-2 on the five shapes (e4m3: 3 mantissa bits on both operands)
    relu = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=bias.cuda(), relu=True, out_dtype=torch.bfloat16, K=qa.shape[1])
    assert (relu.float().cpu() - exact.clamp_min(0).float()).abs().max().item() <= 1.6e-2 * exact.abs().max().item()


def test_lowrank_fp8_forward_close_to_bf16():
    """--precision fp8: the r-rank projections' forward GEMMs on the fp8 MFMA.  Stated tolerance against the bf16 run of the
    same model: logits within 0.10 * max|logit| and loss within 3e-2 (measured 0.075 and 1.5e-2; e4m3 has 3 mantissa bits, ~20
    chained projections; the parity of this model family is unpinned -- there is no reference code for it); the
    backward pass is the bf16 straight-through one and must stay finite."""
    from utils import constant
    from utils.functions import init_optimizer
    from utils.metrics import calculate_metrics
    args, model, V = _build("bf16")
    src, src_len, tgt = _batch(V)
    model = model.cuda().train()
    from asr_hip import ops
    outs = {}
    for mode in ("bf16", "fp8"):
        ops.set_fp8(mode == "fp8")
        opt = init_optimizer(args, model, "noam")
        opt.zero_grad()
        pred, gold, _, _ = model(src.cuda(), src_len, tgt.cuda())
        loss, _ = calculate_metrics(pred, gold, smoothing=0.1, loss_type="ce")
        loss.backward()
        gn = float(sum(p.grad.float().pow(2).sum() for p in model.parameters()).sqrt())
        outs[mode] = (pred.detach().float().cpu(), loss.item(), gn)
    ops.set_fp8(False)
    amax = outs["bf16"][0].abs().max().item()
    d = (outs["fp8"][0] - outs["bf16"][0]).abs().max().item()
    print("fp8 vs bf16 low-rank model: max |dlogit| %.3e (max |logit| %.2f), loss %.4f vs %.4f, |grad| %.3e vs %.3e"
          % (d, amax, outs["fp8"][1], outs["bf16"][1], outs["fp8"][2], outs["bf16"][2]))
    assert d > 0 and d <= 0.10 * amax and abs(outs["fp8"][1] - outs["bf16"][1]) < 3e-2
    assert np.isfinite(outs["fp8"][2]) and abs(outs["fp8"][2] - outs["bf16"][2]) < 0.3 * outs["bf16"][2]


def test_lowrank_model_decodes(monkeypatch):
    """a --rank > 0 checkpoint must go through Transformer.evaluate (greedy and beam) -- the KV-cached decoders read
    full-rank .weight tensors that LowRankLinear does not have, so the low-rank model decodes by re-running the layer modules over
    the prefix (Decoder._kv_cache_supported).  Same strings whether the caller asks for the cache or not."""
    from utils import constant
    from utils.functions import init_transformer_model
    flags = [f if f != "24" else "301" for f in FLAGS]
    args = constant.parse(flags + ["--precision", "fp32", "--cuda"])
    V = 40
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x4E00 + i) for i in range(V - 3)]
    l2i = {c: i for i, c in enumerate(chars)}
    torch.manual_seed(7)
    model = init_transformer_model(args, l2i, {i: c for c, i in l2i.items()}).cuda().eval()
    assert not model.decoder._kv_cache_supported()
    src, src_len, tgt = _batch(V)
    _, hyp_greedy, gold = model.evaluate(src.cuda(), src_len, tgt.cuda())
    assert len(hyp_greedy) == 3 and len(gold) == 3
    feats = model._features(src.cuda())
    enc_out, _ = model.encoder(feats, src_len)
    assert model.decoder.greedy_search(enc_out, use_cache=False) == hyp_greedy
    assert model.decoder.greedy_search(enc_out, use_cache="graph") == hyp_greedy
    _, hyp_beam, _ = model.evaluate(src.cuda(), src_len, tgt.cuda(), beam_search=True, beam_width=2, beam_nbest=1)
    assert len(hyp_beam) == 3
