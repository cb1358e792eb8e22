"""Decoder-sized GEMMs (1600 / 3200 rows): eight-wave 128 x 128 blocks against the four-wave 64 x 64 kernel with and without its
private three-stage ring, NT (forward) and NN (data gradient).  us per launch, back-to-back launches.  python tools/ab/mb_small_gemm.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "ab"))
from asr_hip import lib as L  # noqa: E402
from asr_hip import ops  # noqa: E402
from mb_gemm_big import timeit  # noqa: E402

D = torch.device("cuda")


def main():
    g = torch.Generator().manual_seed(1)
    print("== gemm_nt (M, N, K), bias epilogue: us per launch")
    for M, N, K in [(3200, 512, 512), (3200, 1536, 512), (3200, 2048, 512), (3200, 512, 2048), (3200, 1024, 512), (1600, 512, 512),
                    (1600, 1536, 512), (1600, 2048, 512), (1600, 512, 2048), (6400, 1024, 512), (6400, 64, 512), (6400, 512, 64)]:
        A = torch.randn(M, K, generator=g).to(D).bfloat16()
        B = (torch.randn(N, K, generator=g) * K ** -0.5).to(D).bfloat16()
        bias = torch.randn(N, generator=g).to(D)
        out = torch.empty(M, N, device=D, dtype=torch.bfloat16)
        row = []
        for name, tv in [("8-wave 128/2", dict(GEMM_BIG=128, GEMM_BIG_NS=2)), ("8-wave 128/4", dict(GEMM_BIG=128, GEMM_BIG_NS=4)),
                         ("4-wave 1 stage", dict(GEMM_BIG=0, NT_RING=0)), ("4-wave ring", dict(GEMM_BIG=0, NT_RING=100000)), ("auto", dict())]:
            for kk in ("GEMM_BIG", "GEMM_BIG_NS", "NT_RING"):
                L.set_tuning(kk, tv.get(kk))
            us = timeit(lambda: ops.gemm_nt(A, B, out=out, bias=bias))
            row.append("%s %5.1f" % (name, us))
        print("  %5d %5d %5d : %s" % (M, N, K, " | ".join(row)))
    for kk in ("GEMM_BIG", "GEMM_BIG_NS", "NT_RING"):
        L.set_tuning(kk, None)
    print("== gemm_nn out (M, N) = dy (M, K) @ w (K, N): us per launch")
    for M, N, K in [(3200, 512, 512), (3200, 512, 1536), (3200, 512, 2048), (3200, 2048, 512), (3200, 512, 1024), (1600, 512, 512),
                    (1600, 512, 1536), (1600, 512, 2048), (1600, 2048, 512), (3200, 512, 4416), (6400, 512, 64), (6400, 64, 512)]:
        dy = torch.randn(M, K, generator=g).to(D).bfloat16()
        w = (torch.randn(K, N, generator=g) * K ** -0.5).to(D).bfloat16()
        out = torch.empty(M, N, device=D, dtype=torch.bfloat16)
        row = []
        for name, tv in [("8-wave", dict(GEMM_BIG_NN=2)), ("4-wave 1 stage", dict(GEMM_BIG_NN=0, NN_RING=0)),
                         ("4-wave ring", dict(GEMM_BIG_NN=0, NN_RING=100000)), ("auto", dict())]:
            for kk in ("GEMM_BIG_NN", "NN_RING"):
                L.set_tuning(kk, tv.get(kk))
            us = timeit(lambda: ops.gemm_nn(dy, w, out=out))
            row.append("%s %5.1f" % (name, us))
        print("  %5d %5d %5d : %s" % (M, N, K, " | ".join(row)))
    for kk in ("GEMM_BIG_NN", "NN_RING"):
        L.set_tuning(kk, None)


if __name__ == "__main__":
    main()
