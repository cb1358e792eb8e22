"""KV-cached decoding (asr_hip/decode.py) against the reference-style full re-run of the decoder at every step
(models/asr/transformer.py:316-517): same kernels, same per-row arithmetic -> identical tokens / strings / scores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V = 40


def _model(precision, layers=2):
    from utils import constant
    from utils.functions import init_transformer_model
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x61 + i) for i in range(V - 3)]
    l2i = {c: i for i, c in enumerate(chars)}
    i2l = {i: c for c, i in l2i.items()}
    args = constant.parse(["--num-layers", str(layers), "--num-heads", "8", "--dim-model", "512", "--dim-key", "64",
                           "--dim-value", "64", "--dim-inner", "256", "--dim-emb", "512", "--feat_extractor", "vgg_cnn",
                           "--tgt-max-len", "301", "--src-max-len", "64", "--dropout", "0.1", "--precision", precision,
                           "--cuda"])
    torch.manual_seed(7)
    model = init_transformer_model(args, l2i, i2l).cuda().eval()
    # make EOS reachable at different steps: bias the output layer a little towards EOS
    with torch.no_grad():
        model.decoder.output_linear.weight[2] += 0.02
    return model


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cached_step_equals_full_rerun_row(precision):
    """logits of position t from the cache == row t of the decoder run over the whole prefix (teacher forcing)."""
    model = _model(precision)
    dec = model.decoder
    g = torch.Generator().manual_seed(3)
    B, Te, T = 3, 37, 21
    enc = torch.randn(B, Te, 512, generator=g).cuda()
    ys = torch.randint(3, V, (B, T), generator=g).cuda()
    ys[:, 0] = 1
    from asr_hip.decode import DecoderKVCache
    full = dec._step_logits(ys, enc).float()
    cache = DecoderKVCache(dec, enc, max_len=T)
    tol = 2e-5 if precision == "fp32" else 2e-2
    for t in range(T):
        lg = cache.step(ys[:, t].contiguous())
        err = (lg - full[:, t]).abs().max().item()
        assert err <= tol * max(1.0, full[:, t].abs().max().item()), (t, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_greedy_and_beam_cached_match_uncached(precision):
    model = _model(precision, layers=1)
    dec = model.decoder
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(2, 12, 512, generator=g).cuda()
    a = dec.greedy_search(enc, use_cache=True)
    b = dec.greedy_search(enc, use_cache=False)
    assert a == b and len(a) == 2
    ia, sa = dec.beam_search(enc, beam_width=3, nbest=2, use_cache=True)
    ib, sb = dec.beam_search(enc, beam_width=3, nbest=2, use_cache=False)
    assert sa == sb and ia == ib
