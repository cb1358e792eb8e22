"""Diagnostic (round 6): how reproducible are the two arms of GraphedTrainStep._verify_one_graph?  Run under ASR_FORCE_DDP=1."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
import bench
dist.init_process_group("nccl", rank=0, world_size=1)
from asr_hip import ops, params as P
from utils import constant
from utils.functions import init_optimizer, init_transformer_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
drop = sys.argv[2] if len(sys.argv) > 2 else "0.1"
args = constant.parse(bench.MODEL_FLAGS + ["--dropout", drop, "--precision", "bf16", "--cuda", "--batch-size", str(B), "--parallel"])
l2i, i2l = bench.labels()
torch.manual_seed(123456)
model = init_transformer_model(args, l2i, i2l).cuda(); model.train()
opt = init_optimizer(args, model, "noam")
src, src_len, tgt = bench.synthetic_batch(B, torch); src, tgt = src.cuda(), tgt.cuda()
from asr_hip.graph import GraphedTrainStep
gs = GraphedTrainStep(model, opt, 0.1, src, src_len, tgt, warmup_steps=2, ddp_graph="one")
print("mode", gs.ddp_graph_mode)
adam = opt.optimizer; flat = adam.flat; st = ops.step_state(src.device)
shadow = flat.shadow_for_step(torch.bfloat16)
snap = [t.clone() for t in (flat.data, adam._m, adam._v, st, shadow)]
def restore():
    for d, s in zip((flat.data, adam._m, adam._v, st, shadow), snap): d.copy_(s)
    P._state["seed_ctr"] = gs._seed_ctr_at_capture
def probe(run):
    restore(); run(); torch.cuda.synchronize()
    return [float(x) for x in flat.stats[:3].double().cpu()], float(flat.grad.double().pow(2).sum())
for name, run in (("eager", gs._eager_step), ("eager", gs._eager_step), ("replay", gs.graph.replay), ("replay", gs.graph.replay), ("eager", gs._eager_step)):
    print(name, probe(run), "st", st.tolist())
dist.destroy_process_group()
