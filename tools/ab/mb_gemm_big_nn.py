"""Eight-wave data-gradient blocks (csrc/gemm_big.hip, NN form) against the four-wave asr_gemm_nn on the data-gradient shapes of the model:
out (M, N) = dy (M, K) @ w (K, N), plain / accumulate / ReLU-mask epilogues.  python tools/ab/mb_gemm_big_nn.py"""
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


def main():
    shapes = [(6400, 512, 1536, "acc"), (6400, 512, 512, "plain"), (6400, 512, 2048, "acc"), (6400, 2048, 512, "mask"), (6400, 5120, 512, "plain"),
              (6400, 512, 1024, "acc"), (3200, 512, 1536, "acc"), (3200, 512, 512, "plain"), (3200, 512, 2048, "acc"), (3200, 2048, 512, "mask"),
              (3200, 512, 4416, "plain"), (12720, 512, 1536, "acc"), (12720, 512, 2048, "acc"), (12720, 2048, 512, "mask"), (12720, 512, 512, "plain")]
    variants = [("4-wave", dict(GEMM_BIG_NN=0)), ("8w/2", dict(GEMM_BIG_NN=2, GEMM_BIG_NS=2)), ("8w/3", dict(GEMM_BIG_NN=2, GEMM_BIG_NS=3)),
                ("8w/4", dict(GEMM_BIG_NN=2, GEMM_BIG_NS=4)), ("auto", dict(GEMM_BIG_NN=1))]
    print("== gemm_nn out (M, N) = dy (M, K) w (K, N), bf16: us per launch (TF/s)")
    g = torch.Generator().manual_seed(1)
    for M, N, K, kind in shapes:
        dy = torch.randn(M, K, generator=g).to(D).bfloat16()
        w = (torch.randn(K, N, generator=g) * K ** -0.5).to(D).bfloat16()
        out = torch.zeros(M, N, device=D, dtype=torch.bfloat16)
        mask = torch.randn(M, N, generator=g).to(D).bfloat16() if kind == "mask" else None
        row = []
        for name, tv in variants:
            for kk in ("GEMM_BIG_NN", "GEMM_BIG_NS"):
                L.set_tuning(kk, tv.get(kk))
            us = timeit(lambda: ops.gemm_nn(dy, w, out=out, accumulate=kind == "acc", relu_mask=mask))
            row.append("%s %5.1f (%4.0f)" % (name, us, 2.0 * M * N * K / us / 1e6))
        print("  %5d %5d %5d %-5s : %s" % (M, N, K, kind, " | ".join(row)))
    for kk in ("GEMM_BIG_NN", "GEMM_BIG_NS"):
        L.set_tuning(kk, None)


if __name__ == "__main__":
    main()
