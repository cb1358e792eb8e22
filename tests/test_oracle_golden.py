"""Pins oracle/asr_oracle.py against the golden vectors produced by executing the real reference
(oracle/gen_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

CASES = ["vgg_tiny", "emb_tiny", "raw_tiny", "dkdv_tiny"]       # dkdv_tiny: dim_key 16 != dim_value 24 (round 6)


def noise_driven(k, cfg):
    """Parameters whose true gradient is identically zero: key biases (softmax shift invariance) and, for emb_cnn,
    conv biases feeding BatchNorm (mean subtraction).  Autograd returns rounding noise for them."""
    return k.endswith("key_linear.bias") or (cfg.feat_extractor == "emb_cnn" and k in ("conv.0.bias", "conv.3.bias"))


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    w0 = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}
    return z, w0


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_two_steps(golden_dir, name):
    torch.set_num_threads(4)
    z, w = load(golden_dir, name)
    cfg = O.Cfg.from_flags(str(z["flags"]))
    src, src_len, tgt = torch.from_numpy(z["src"]), torch.from_numpy(z["src_len"]), torch.from_numpy(z["tgt"])
    sm = float(z["smoothing"])
    names = O.trainable_names(w, cfg)
    params = {k: w[k] for k in names}
    opt = O.NoamAdam(params, model_size=int(z["dim_input"]))
    bn = {}
    r = O.train_step(w, cfg, src, src_len, tgt, sm, opt=opt, bn_state=bn)
    np.testing.assert_allclose(r["pred"].numpy(), z["pred"], rtol=0, atol=2e-5)
    assert np.array_equal(r["gold"].numpy(), z["gold"])
    # Rows whose decoder input is EOS are zeroed (transformer.py:282,536) -> all V logits are exactly 0.
    # torch.topk(pred, 1) (transformer.py:80) breaks that exact tie in a kernel-specific way (index 22 for
    # V=35 on this CPU build); utils/metrics.py:89 (`pred.max(1)[1]`) and the oracle/product return the
    # LOWEST index.  hyp is therefore compared on non-degenerate rows, and tied rows must give index 0.
    tied = (z["pred"].max(-1) == z["pred"].min(-1))
    assert np.array_equal(r["hyp"].numpy()[~tied], z["hyp_seq"][~tied])
    assert (r["hyp"].numpy()[tied] == 0).all()
    assert r["num_correct"] == int(z["num_correct"])
    assert abs(r["loss"] - float(z["loss"])) < 2e-6
    for k in names:
        ref = z["g0/" + k]
        tol = 1e-6 + 2e-5 * np.abs(ref).max()
        np.testing.assert_allclose(r["grads"][k].numpy(), ref, rtol=0, atol=tol, err_msg=k)
    assert abs(opt.rate - float(z["lr1"])) < 1e-12
    if cfg.emb_trg_sharing:
        w["decoder.output_linear.weight"] = w["decoder.trg_embedding.weight"]
    r2 = O.train_step(w, cfg, src, src_len, tgt, sm, opt=opt, bn_state=bn)
    assert abs(r2["loss"] - float(z["loss2"])) < 5e-6
    assert abs(opt.rate - float(z["lr2"])) < 1e-12
    for k in names:
        # key_linear.bias gradients are mathematically zero (softmax is shift invariant); what autograd returns is
        # ~1e-9 rounding noise that Adam normalises to a full +-lr update, so each side moves by up to lr per step: only |delta| <= 2*(lr1+lr2) is pinned.
        atol = 2.1 * (float(z["lr1"]) + float(z["lr2"])) if noise_driven(k, cfg) else 5e-6
        np.testing.assert_allclose(w[k].numpy(), z["w2/" + k], rtol=0, atol=atol, err_msg=k)
    for k, v in bn.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), z["w2/" + k].astype(np.float64), rtol=1e-5, atol=1e-5)  # running_mean carries the noise-driven conv bias
    # eval-mode forward with the updated weights (BatchNorm running stats for emb_cnn)
    # (uses the reference's own post-step weights so that noise-driven parameters do not enter the comparison)
    wl = dict(w)
    wl.update({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w2/")})
    pred, _, _ = O.transformer_forward(wl, cfg, src, src_len, tgt, training=False, bn_state={})
    np.testing.assert_allclose(pred.detach().numpy(), z["pred_eval"], rtol=0, atol=5e-5)


def test_preprocess_strips_interior_pad(golden_dir):
    z, _ = load(golden_dir, "raw_tiny")
    tgt = torch.from_numpy(z["tgt"])
    assert (tgt[1, 2] == 0) and (tgt[1, 3] != 0)
    seq_in, seq_out = O.decoder_preprocess(tgt, 16)
    assert np.array_equal(seq_out.numpy(), z["gold"])
    assert seq_in[1, 0] == O.SOS and (seq_in[1] == O.EOS).sum() == 16 - 1 - 8


def test_edit_distance_known_answers():
    assert O.edit_distance("kitten", "sitting") == 3
    assert O.edit_distance("", "abc") == 3
    assert O.edit_distance("flaw", "lawn") == 2
    assert O.edit_distance("abc", "abc") == 0
