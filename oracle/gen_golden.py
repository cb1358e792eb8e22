#!/usr/bin/env python3
"""Generate golden fixtures by EXECUTING the unmodified reference on CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference exists);
the GPU box never runs this script -- it only reads the committed tests/golden/*.npz.

Recipe follows SURVEY.md Appendix A: stub Levenshtein/torchaudio, preset sys.argv before the first
reference import (utils/constant.py:99 parses argv at import time), never write bytecode into the
reference tree.  One interpreter per configuration (the reference Namespace is a process global).

usage:  python oracle/gen_golden.py <case>      case in: vgg_tiny | emb_tiny | raw_tiny
        python oracle/gen_golden.py all         (spawns one subprocess per case)
"""
import os
import subprocess
import sys
import types

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (argv flags, B, T_src, src_len list, tgt real lengths)
    "vgg_tiny": dict(
        flags=["--num-layers", "2", "--num-heads", "2", "--dim-model", "32", "--dim-key", "16", "--dim-value", "16",
               "--dim-inner", "64", "--dim-emb", "32", "--feat_extractor", "vgg_cnn", "--tgt-max-len", "12",
               "--src-max-len", "64", "--label-smoothing", "0.1", "--dropout", "0.0"],
        B=3, T=64, src_len=[64, 40, 9], tgt_len=[11, 7, 3], smoothing=0.1),
    "emb_tiny": dict(
        flags=["--num-layers", "1", "--num-heads", "2", "--dim-model", "32", "--dim-key", "16", "--dim-value", "16",
               "--dim-inner", "64", "--dim-emb", "32", "--feat_extractor", "emb_cnn", "--tgt-max-len", "10",
               "--src-max-len", "96", "--label-smoothing", "0.0", "--dropout", "0.0"],
        B=2, T=96, src_len=[96, 30], tgt_len=[9, 4], smoothing=0.0),
    "raw_tiny": dict(  # --feat_extractor "" : no CNN, T' = T, D_in = 161 ; exercises the length masks fully
        flags=["--num-layers", "2", "--num-heads", "4", "--dim-model", "64", "--dim-key", "16", "--dim-value", "16",
               "--dim-inner", "128", "--dim-emb", "64", "--feat_extractor", "", "--tgt-max-len", "16",
               "--src-max-len", "50", "--label-smoothing", "0.1", "--dropout", "0.0", "--emb_trg_sharing"],
        B=4, T=50, src_len=[50, 37, 20, 5], tgt_len=[15, 9, 4, 1], smoothing=0.1),
}


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def run_case(name):
    import json
    import numpy as np
    cfg = CASES[name]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.argv = ["train.py"] + cfg["flags"]
    lev = types.ModuleType("Levenshtein"); lev.distance = edit_distance
    sys.modules["Levenshtein"] = lev
    sys.modules["torchaudio"] = types.ModuleType("torchaudio")
    import torch
    torch.set_num_threads(4)
    from utils import constant                                    # argparse runs here
    from utils.functions import init_transformer_model, init_optimizer
    from utils.metrics import calculate_metrics

    labels = json.load(open(os.path.join(REF, "data/labels/labels.json")))
    labels = constant.PAD_CHAR + constant.SOS_CHAR + constant.EOS_CHAR + "".join(labels)
    label2id = {c: i for i, c in enumerate(labels)}
    id2label = {i: c for c, i in label2id.items()}
    V = len(label2id)

    torch.manual_seed(123456)
    model = init_transformer_model(constant.args, label2id, id2label)
    opt = init_optimizer(constant.args, model, "noam")
    model.train()
    # make biases / LN affine non-trivial so that bias & affine paths are really checked
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))

    B, T = cfg["B"], cfg["T"]
    g = torch.Generator().manual_seed(1234)
    src = torch.randn(B, 1, 161, T, generator=g)
    src_len = torch.tensor(cfg["src_len"], dtype=torch.int32)
    for b in range(B):
        src[b, :, :, int(src_len[b]):] = 0.0                      # loader zero-pads (data_loader.py:196-209)
    Lmax = max(cfg["tgt_len"])
    tgt = torch.zeros(B, Lmax, dtype=torch.int64)
    for b, L in enumerate(cfg["tgt_len"]):
        tgt[b, :L] = torch.randint(3, V, (L,), generator=g)
    if name == "raw_tiny":
        tgt[1, 2] = 0                                             # an interior PAD: preprocess strips it (transformer.py:258)

    out = {"src": src.numpy(), "src_len": src_len.numpy(), "tgt": tgt.numpy(),
           "V": np.int64(V), "dim_input": np.int64(constant.args.dim_input),
           "smoothing": np.float64(cfg["smoothing"]), "flags": np.array(" ".join(cfg["flags"]))}
    for k, v in model.state_dict().items():
        out["w0/" + k] = v.detach().numpy().copy()

    opt.zero_grad()
    pred, gold, hyp_seq, gold_seq = model(src, src_len, tgt)
    loss, ncorrect = calculate_metrics(pred, gold, smoothing=cfg["smoothing"], loss_type="ce")
    loss.backward()
    out["pred"] = pred.detach().numpy().copy()
    out["gold"] = gold.numpy().copy()
    out["hyp_seq"] = hyp_seq.numpy().copy()
    out["loss"] = np.float64(loss.item())
    out["num_correct"] = np.int64(ncorrect)
    for k, p in model.named_parameters():
        out["g0/" + k] = p.grad.detach().numpy().copy()
    opt.step()
    out["lr1"] = np.float64(opt._rate)
    # second step (Adam moments + Noam counter exercised twice)
    opt.zero_grad()
    pred2, gold2, _, _ = model(src, src_len, tgt)
    loss2, _ = calculate_metrics(pred2, gold2, smoothing=cfg["smoothing"], loss_type="ce")
    loss2.backward(); opt.step()
    out["loss2"] = np.float64(loss2.item())
    out["lr2"] = np.float64(opt._rate)
    for k, v in model.state_dict().items():
        if k.endswith(".pe"):
            continue
        out["w2/" + k] = v.detach().numpy().copy()
    # encoder output after the two steps, eval mode (exercises BatchNorm running stats for emb_cnn)
    model.eval()
    with torch.no_grad():
        pred_eval, _, _, _ = model(src, src_len, tgt)
    out["pred_eval"] = pred_eval.numpy().copy()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", out["loss"], "loss2", out["loss2"], "ncorrect", ncorrect, "->", path,
          "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "all":
        for c in CASES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), c],
                                  env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    else:
        run_case(which)
