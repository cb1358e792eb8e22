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
    a = dec.greedy_search(enc, use_cache="graph")        # one hipGraph replay per token, kernel-per-op step
    b = dec.greedy_search(enc, use_cache=False)
    c = dec.greedy_search(enc, use_cache="eager")
    assert a == b == c and len(a) == 2
    ia, sa = dec.beam_search(enc, beam_width=3, nbest=2, use_cache=True)
    ib, sb = dec.beam_search(enc, beam_width=3, nbest=2, use_cache=False)
    assert sa == sb and ia == ib


# ------------------------------------------------------------------------------------------------ vs the reference
def _reference_case(golden_dir, precision):
    import os
    from utils import constant
    from utils.functions import init_transformer_model
    z = np.load(os.path.join(golden_dir, "dec_tiny.npz"))
    chars = constant.PAD_CHAR + constant.SOS_CHAR + constant.EOS_CHAR + "_'abcdefghijklmnopqrstuvwxyz "
    l2i = {c: i for i, c in enumerate(chars)}
    i2l = {i: c for c, i in l2i.items()}
    args = constant.parse(str(z["flags"]).split() + ["--precision", precision, "--cuda"])
    model = init_transformer_model(args, l2i, i2l)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}, strict=True)
    return z, model.cuda().eval()


def _error_counts(strs_hyps, strs_gold):
    """The accumulation of test.py:evaluate (reference test.py:42-58) with the product's metric functions."""
    from utils import constant
    from utils.metrics import calculate_cer, calculate_wer
    tc = tch = tw = twd = 0
    for h, g in zip(strs_hyps, strs_gold):
        for ch in (constant.EOS_CHAR, constant.SOS_CHAR, constant.PAD_CHAR):
            h, g = h.replace(ch, ""), g.replace(ch, "")
        tw += calculate_wer(h, g)
        tc += calculate_cer(h.strip(), g.strip())
        twd += len(g.split(" "))
        tch += len(g)
    return [tc, tch, tw, twd]


@pytest.mark.parametrize("use_cache", [True, "eager", False])
def test_decode_strings_and_cer_match_the_reference(golden_dir, use_cache):
    """fp32 mode: Transformer.evaluate() greedy and beam-4 produce the REFERENCE's strings (tests/golden/dec_tiny.npz: the
    reference's own evaluate() on a model the reference trained for 170 steps; greedy CER 17/34, beam CER 30/34) and
    therefore its CER / WER counts -- with the KV cache and with the reference-style full re-run."""
    z, model = _reference_case(golden_dir, "fp32")
    src, src_len, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["src_len"]), torch.from_numpy(z["tgt"]).cuda()
    dec = model.decoder
    with torch.no_grad():
        feats = model._features(src)
        enc, _ = model.encoder(feats, src_len)
        assert (enc.float().cpu() - torch.from_numpy(z["enc_out"])).abs().max().item() < 5e-5
        _, gold, *_ = dec(tgt, enc, src_len)
    strs_gold = ["".join(model.id2label[int(x)] for x in row) for row in gold.cpu().tolist()]
    assert strs_gold == [str(s) for s in z["gold_strs"]]
    greedy = dec.greedy_search(enc, use_cache=use_cache)
    assert greedy == [str(s) for s in z["greedy"]]
    _, beam = dec.beam_search(enc, beam_width=int(z["beam_width"]), nbest=1, c_weight=0.1, use_cache=use_cache)
    assert beam == [str(s) for s in z["beam"]]
    assert _error_counts(greedy, strs_gold) == [int(v) for v in z["greedy_cer"]]
    assert _error_counts(beam, strs_gold) == [int(v) for v in z["beam_cer"]]
    if use_cache is True:       # the public entry point (reference transformer.py:87-124)
        _, hyps, golds = model.evaluate(src, src_len, tgt, beam_search=True, beam_width=int(z["beam_width"]), beam_nbest=1,
                                        c_weight=0.1)
        assert hyps == beam and golds == strs_gold


def test_decode_bf16_cer_close_to_reference(golden_dir):
    """bf16 mode decodes the same utterances; near-tied steps may flip, so the bound is on the error COUNTS:
    |CER chars - reference| <= 3 of 34 gold characters for greedy and for beam-4."""
    z, model = _reference_case(golden_dir, "bf16")
    src, src_len, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["src_len"]), torch.from_numpy(z["tgt"]).cuda()
    _, greedy, golds = model.evaluate(src, src_len, tgt, beam_search=False)
    _, beam, _ = model.evaluate(src, src_len, tgt, beam_search=True, beam_width=int(z["beam_width"]), beam_nbest=1, c_weight=0.1)
    gc, bc = _error_counts(greedy, golds), _error_counts(beam, golds)
    assert gc[1] == int(z["greedy_cer"][1]) and abs(gc[0] - int(z["greedy_cer"][0])) <= 3, (gc, greedy)
    assert abs(bc[0] - int(z["beam_cer"][0])) <= 3, (bc, beam)
