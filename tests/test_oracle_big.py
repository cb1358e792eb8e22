"""Pins oracle/asr_oracle.py to the reference at the BASELINE shapes and on its decode path (CPU only).

  * cfg0 / cfg1_b2 / cfg3_shape: the oracle's train step on weights rebuilt from the reference's seed vs the summaries
    the executed reference left in tests/golden/ (logit columns, log-sum-exp rows, loss, arg-max rows, per-parameter
    gradient norm / random projection / samples).  Measured: logits 3e-6, loss < 1e-7, gradient samples 5e-5 relative.
  * dec_tiny: oracle greedy / beam-4 strings and the CER / WER counts of reference test.py:42-58 vs the reference's
    own Transformer.evaluate() on a model the reference trained for 170 steps.
"""
import numpy as np
import pytest
import torch

import big_cases as BC
from oracle import asr_oracle as O


@pytest.mark.parametrize("name", BC.BIG_CASES)
def test_oracle_matches_reference_at_baseline_shape(golden_dir, name):
    torch.set_num_threads(8)
    z = BC.load(golden_dir, name)
    args, model, l2i, i2l = BC.build_product(z, "fp32", False)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = BC.oracle_cfg(z)
    src, src_len, tgt = BC.batch(z)
    r = O.train_step(w, cfg, src, src_len, tgt, float(z["smoothing"]), bn_state={})
    e = BC.summary_errors(z, r["pred"], r["loss"], r["grads"])
    assert e["pred_sub"] < 2e-5 and e["pred_lse"] < 2e-5 and e["loss"] < 2e-6, e
    assert np.array_equal(r["gold"].numpy(), z["gold"])
    assert r["num_correct"] == int(z["num_correct"])
    miss, n = BC.argmax_agreement(z, r["hyp"], 1e-4)
    assert miss == 0 and n > 100
    emb = cfg.feat_extractor == "emb_cnn"
    for k in r["grads"]:
        if BC.noise_driven(k, emb):
            continue
        # two fp32 evaluations (the oracle's and the reference's) of a gradient whose fp32 floor against fp64 is e32 (stored by the
        # generator): they may differ by a few floors -- matters only for the near-zero cross-attention query / key gradients of the
        # 12 / 6-layer case (e32 up to 3.5e-4)
        f = 4 * float(z["e32/" + k])
        assert e["gn"][k] < max(2e-4, f) and e["gp"][k] < max(1e-3, f) and e["gs"][k] < max(5e-4, f), (k, e["gn"][k], e["gp"][k], e["gs"][k], f)


def _dec(golden_dir):
    z = BC.load(golden_dir, "dec_tiny")
    w = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
    cfg = O.Cfg.from_flags(str(z["flags"]))
    chars = "¶§¤" + "_'abcdefghijklmnopqrstuvwxyz "
    i2l = {i: c for i, c in enumerate(chars)}
    assert len(i2l) == int(z["V"])
    return z, w, cfg, i2l


def test_oracle_decode_matches_reference_strings_and_cer(golden_dir):
    torch.set_num_threads(8)
    z, w, cfg, i2l = _dec(golden_dir)
    src, src_len, tgt = torch.from_numpy(z["src"]), torch.from_numpy(z["src_len"]), torch.from_numpy(z["tgt"])
    with torch.no_grad():
        x = O.conv_front_end(w, src, cfg.feat_extractor, training=False)
        enc = O.encoder_forward(w, cfg, x, src_len)
        np.testing.assert_allclose(enc.numpy(), z["enc_out"], rtol=0, atol=2e-5)
        _, gold = O.decoder_forward(w, cfg, tgt, enc, src_len)
    strs_gold = ["".join(i2l[int(t)] for t in row) for row in gold.tolist()]
    assert strs_gold == [str(s) for s in z["gold_strs"]]
    greedy = O.greedy_search(w, cfg, enc, i2l)
    assert greedy == [str(s) for s in z["greedy"]]
    _, beam = O.beam_search(w, cfg, enc, i2l, beam_width=int(z["beam_width"]), nbest=1, c_weight=0.1)
    assert beam == [str(s) for s in z["beam"]]
    assert list(O.eval_error_counts(greedy, strs_gold)) == [int(v) for v in z["greedy_cer"]]
    assert list(O.eval_error_counts(beam, strs_gold)) == [int(v) for v in z["beam_cer"]]


def test_oracle_ctc_is_the_reference_expression():
    """oracle.ctc_loss restates utils/metrics.py:133-154 (log_softmax + F.ctc_loss(mean)); pinned by a brute-force sum over all
    alignments of a tiny case: p(l|x) = sum over paths that collapse to l."""
    import itertools
    import torch
    from oracle import asr_oracle as O
    g = torch.Generator().manual_seed(0)
    T, V = 4, 3
    pred = torch.randn(1, T, V, generator=g).double()
    tgt = torch.tensor([[1, 2]])
    lp = torch.log_softmax(pred[0], dim=1)
    total = 0.0
    for path in itertools.product(range(V), repeat=T):
        col, prev = [], None
        for s in path:
            if s != prev and s != 0:
                col.append(s)
            prev = s
        if col == [1, 2]:
            total += float(torch.exp(sum(lp[t, s] for t, s in enumerate(path))))
    ref = -torch.log(torch.tensor(total, dtype=torch.float64)) / 2          # reduction="mean": / target length, mean over the batch of 1
    got = O.ctc_loss(pred, tgt, torch.tensor([T]), torch.tensor([2]))
    assert abs(float(got) - float(ref)) < 1e-12
