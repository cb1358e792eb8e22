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
