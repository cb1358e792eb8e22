"""Microbenchmark of the full-resolution level's three kernels at the benchmark shape (B=32, 161 x 800), optional ablations."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "end2end-asr-pytorch_amd"))
import torch
from asr_hip import ops, lib as L
D = "cuda:0"
B, H, W = 32, 161, 800
g = torch.Generator().manual_seed(0)
src = torch.randn(B, 1, H, W, generator=g).to(D)
w0 = (torch.randn(64, 1, 3, 3, generator=g) / 3).to(D); b0 = (torch.randn(64, generator=g) / 3).to(D)
w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(D); b2 = (torch.randn(64, generator=g) / 3).to(D)
wk = torch.empty(64, 9, 64, device=D, dtype=torch.bfloat16); wd = torch.empty_like(wk)
ops.conv_pack_weight(w2, wk, wd)
dp = torch.randn(B, H // 2, W // 2, 64, generator=g).to(D, torch.bfloat16)
pool, code = ops.vgg_level0_fwd(src, w0, b0, wk, b2)
dw2 = torch.zeros(64, 64, 3, 3, device=D); db2 = torch.zeros(64, device=D); dw0 = torch.zeros(64, 1, 3, 3, device=D); db0 = torch.zeros(64, device=D)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fns = {"fwd": lambda: ops.vgg_level0_fwd(src, w0, b0, wk, b2), "wgrad": lambda: ops.vgg_level0_wgrad(src, w0, b0, dp, code, dw2, db2),
       "dgrad": lambda: ops.vgg_level0_dgrad(dp, code, src, w0, b0, wd, dw0, db0)}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    print("level 0: " + "  ".join("%s %.1f us" % (k, t(f)) for k, f in fns.items()), flush=True)
