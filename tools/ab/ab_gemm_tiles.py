"""A/B of the NT GEMM's tile size x LDS stages on the model's forward shapes (tuning switches GEMM_TILE / GEMM_NS of csrc/gemm.hip).
Round 3 question: 1-KB LDS-DMA pieces per flop, not occupancy, look like the bound of the 64 x 64 tile (6400 x 2048 x 512: 400 pieces per
SIMD in 24.8 us = 149 cycles per piece; the same constant reproduces 6400 x 1536 x 512 and 6400 x 512 x 512) -- does a 128 x 128 tile
with a real pipeline (2 - 3 stages, now that its epilogue no longer reserves an fp32 tile of LDS) halve the time?
usage: python tools/ab/ab_gemm_tiles.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
from asr_hip import lib as L  # noqa: E402
from asr_hip import ops  # noqa: E402

D = torch.device("cuda")


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


shapes = [(6400, 512, 512), (6400, 1536, 512), (6400, 2048, 512), (6400, 512, 2048), (6400, 512, 5120), (3200, 1536, 512), (3200, 2048, 512),
          (3200, 512, 2048), (12720, 2048, 512), (12720, 512, 2048)]
print("== gemm_nt (M,N,K) bf16 -> bf16, bias + ReLU epilogue; us (TF/s) per tile x stages")
for M, N, K in shapes:
    A = torch.randn(M, K, device=D).bfloat16()
    B = torch.randn(N, K, device=D).bfloat16()
    bias = torch.randn(N, device=D)
    out = torch.empty(M, N, device=D, dtype=torch.bfloat16)
    ref = None
    row = []
    for tile, tn in ((2, "64x64"), (1, "128x64"), (0, "128x128")):
        for ns in (1, 2, 3):
            L.set_tuning("GEMM_TILE", tile)
            L.set_tuning("GEMM_NS", ns)
            us = timeit(lambda: ops.gemm_nt(A, B, out=out, bias=bias, relu=True))
            if ref is None:
                ref = out.float().clone()
            err = (out.float() - ref).abs().max().item()
            row.append("%s/%d %5.1f (%4.0f)%s" % (tn, ns, us, 2.0 * M * N * K / us / 1e6, "" if err < 0.1 else " ERR %.2g" % err))
    L.set_tuning("GEMM_TILE", None)
    L.set_tuning("GEMM_NS", None)
    print("  %5d %5d %5d : %s" % (M, N, K, " | ".join(row)))
