"""Development aid: where do the device-to-device copies of one eager training step come from?"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end2end-asr-pytorch_amd")]
import torch
import bench
from utils import constant
from utils.functions import init_optimizer, init_transformer_model
from utils.metrics import calculate_loss
args = constant.parse(bench.MODEL_FLAGS + ["--dropout", "0.1", "--cuda", "--batch-size", "32"])
l2i, i2l = bench.labels()
model = init_transformer_model(args, l2i, i2l).cuda().train()
opt = init_optimizer(args, model, "noam")
src, src_len, tgt = bench.synthetic_batch(32, torch)
src, tgt = src.cuda(), tgt.cuda()
def step():
    opt.zero_grad()
    pred, gold, hyp, _ = model(src, src_len, tgt)
    loss = calculate_loss(pred, gold, smoothing=0.1, loss_type="ce")
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::zeros", "aten::fill_", "aten::zero_"):
        st = [s for s in (e.stack or []) if "end2end-asr-pytorch_amd" in s or "bench.py" in s]
        key = (e.name, str(e.input_shapes)[:60], st[0][-70:] if st else "?")
        cnt[key] += 1
for k, v in cnt.most_common(40):
    print(v, k)
