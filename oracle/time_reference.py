#!/usr/bin/env python3
"""CPU baseline leg of bench.py (`cpu_baseline.kind == "reference"`): times the UNMODIFIED reference training step
imported from /root/reference, per BASELINE.md section 2.  TEST / MEASUREMENT INFRASTRUCTURE, not product code; it
only runs where the reference tree exists (the build container), never on the GPU box.

Prints one JSON line: {"times": [seconds per timed step], "threads": torch.get_num_threads()}.
"""
import argparse
import json
import os
import sys
import time
import types

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--budget", type=float, default=150.0)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.argv = ["train.py", "--num-layers", "4", "--num-heads", "8", "--dim-model", "512", "--dim-key", "64", "--dim-value", "64",
            "--dim-inner", "2048", "--dim-emb", "512", "--feat_extractor", "vgg_cnn", "--tgt-max-len", "100", "--src-max-len",
            "800", "--label-smoothing", "0.1", "--dropout", "0.1"]
lev = types.ModuleType("Levenshtein")
lev.distance = lambda x, y: 0
sys.modules["Levenshtein"] = lev
sys.modules["torchaudio"] = types.ModuleType("torchaudio")
import torch  # noqa: E402
from utils import constant  # noqa: E402
from utils.functions import init_optimizer, init_transformer_model  # noqa: E402
from utils.metrics import calculate_metrics  # noqa: E402

V = 4364
chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x4E00 + i) for i in range(V - 3)]
l2i = {c: i for i, c in enumerate(chars)}
i2l = {i: c for c, i in l2i.items()}
torch.manual_seed(123456)
model = init_transformer_model(constant.args, l2i, i2l)
opt = init_optimizer(constant.args, model, "noam")
model.train()
g = torch.Generator().manual_seed(1234)
src = torch.randn(a.batch, 1, 161, 800, generator=g)
tgt = torch.randint(3, V, (a.batch, 99), generator=g)
src_len = torch.full((a.batch,), 800, dtype=torch.int32)
times = []
t_start = time.time()
for i in range(a.steps + 1):
    t0 = time.time()
    opt.zero_grad()
    pred, gold, _, _ = model(src, src_len, tgt)
    loss, _ = calculate_metrics(pred, gold, smoothing=0.1, loss_type="ce")
    loss.backward()
    opt.step()
    dt = time.time() - t0
    if i > 0 or dt > a.budget / 2:
        times.append(dt)
    if time.time() - t_start > a.budget:
        break
print(json.dumps({"times": times, "threads": torch.get_num_threads()}))
