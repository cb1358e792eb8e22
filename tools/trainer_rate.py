#!/usr/bin/env python3
"""Steps per second of the TRAINER's step body (trainer/asr/trainer.py:_run_batch: H2D through the prefetcher, zero_grad, forward,
loss, backward, optimiser, one D2H copy of the token ids, strings, CER / WER) on the benchmark's workload -- configs[1], B = 32,
synthetic batches that arrive as host tensors like the collate function's -- next to bench.py's graph-replayed step.
usage: [RATE_MODE=prefetch|resident|pinned] [RATE_BUCKETS=N] python tools/trainer_rate.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
sys.path.insert(0, ROOT)
import bench as Bn                                   # noqa: E402
from utils import constant                           # noqa: E402
from utils.data_loader import DevicePrefetcher       # noqa: E402
from utils.functions import init_optimizer, init_transformer_model   # noqa: E402
from trainer.asr.trainer import Trainer              # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 32
_flags = Bn.MODEL_FLAGS + ["--dropout", "0.1", "--precision", "bf16", "--cuda", "--batch-size", str(B)]
if "RATE_BUCKETS" in os.environ:                      # an explicit --graph-buckets N (0 = the eager loop)
    _flags += ["--graph-buckets", os.environ["RATE_BUCKETS"]]
args = constant.parse(_flags)
import train as _train                               # noqa: E402
_train.resolve_graph_buckets(args, constant.explicit)        # unset: what `python train.py --cuda ...` resolves to (round 5: 64 for vgg_cnn)
print("# --graph-buckets resolved to %d (%s)" % (args.graph_buckets, "typed" if "RATE_BUCKETS" in os.environ else "train.py's default"))
l2i, i2l = Bn.labels(Bn.V)
model = init_transformer_model(args, l2i, i2l).cuda()
opt = init_optimizer(args, model, "noam")
src, src_len, tgt = Bn.synthetic_batch(B, torch)
batch = (src, tgt, torch.ones(B), src_len, torch.full((B,), tgt.shape[1], dtype=torch.int32))
tr = Trainer()
model.train()
resident = (src.cuda(), tgt.cuda()) + batch[2:]
mode = os.environ.get("RATE_MODE", "prefetch")        # prefetch | resident (batch already on the device) | pinned (pre-pinned host batch)
if mode == "pinned":
    batch = (src.pin_memory(), tgt.pin_memory()) + batch[2:]
for name, n in (("warm-up", 5), ("timed", steps)):
    feed = [resident] * n if mode == "resident" else DevicePrefetcher([batch] * n, torch.device("cuda", 0))
    torch.cuda.synchronize()
    t0 = time.time()
    pending = None
    for data in feed:
        r = tr._run_batch(model, data, 0.1, "ce", i2l, opt)
        if pending is not None:                       # like Trainer.train: the previous step's results after this step's launches
            last = pending.result()
        pending = r if hasattr(r, "result") else None
        if pending is None:
            last = r
    if pending is not None:
        last = pending.result()
    r = last
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    if name == "timed":
        print(mode, "graph-buckets", args.graph_buckets, "trainer step body: %.2f ms/step = %.2f M frames/s (loss %.4f)" % (dt * 1e3, B * Bn.T_SRC / dt / 1e6, r[0]))
