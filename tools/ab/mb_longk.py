#!/usr/bin/env python3
"""Forward GEMMs with few output tiles and a long contraction (FFN2, the encoder input projection): LDS stages x tile shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from asr_hip import lib as L, ops
from microbench import timeit, D

for M, N, K in [(6400, 512, 2048), (3200, 512, 2048), (6400, 512, 5120), (6400, 2048, 512), (3200, 512, 512)]:
    A = torch.randn(M, K, device=D).bfloat16(); B = torch.randn(N, K, device=D).bfloat16()
    bias = torch.randn(N, device=D); out = torch.empty(M, N, device=D, dtype=torch.bfloat16)
    res = []
    for ns in (1, 2, 3):
        for tile in (2, 1, 0):
            L.set_tuning("GEMM_NS", ns); L.set_tuning("GEMM_TILE", tile)
            us = timeit(lambda: ops.gemm_nt(A, B, out=out, bias=bias), iters=50)
            res.append("ns%d %s %5.1f" % (ns, ["128x128", "128x64", "64x64"][tile], us))
    L.set_tuning("GEMM_NS", None); L.set_tuning("GEMM_TILE", None)
    print("%5d %5d %5d : %s" % (M, N, K, " | ".join(res)))
