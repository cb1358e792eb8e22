"""Eight-wave GEMM blocks (csrc/gemm_big.hip) against the four-wave kernels on the linear-layer shapes of the model: time per launch
(back-to-back launches, hot caches) and the largest deviation from an fp32 torch product.  python tools/ab/mb_gemm_big.py"""
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
    shapes = [(6400, 512, 512, "bf16"), (6400, 1536, 512, "bf16"), (6400, 2048, 512, "bf16"), (6400, 512, 2048, "bf16"),
              (6400, 512, 5120, "bf16"), (6400, 4096, 512, "bf16"), (3200, 512, 512, "bf16"), (3200, 1536, 512, "bf16"),
              (3200, 2048, 512, "bf16"), (3200, 512, 2048, "bf16"), (3200, 4364, 512, "f32"), (12720, 2048, 512, "bf16"),
              (12720, 512, 2048, "bf16"), (12720, 1536, 512, "bf16"), (12720, 512, 512, "bf16")]
    variants = [("4-wave", dict(GEMM_BIG=0)), ("256", dict(GEMM_BIG=256)), ("128/2", dict(GEMM_BIG=128, GEMM_BIG_NS=2)),
                ("128/3", dict(GEMM_BIG=128, GEMM_BIG_NS=3)), ("128/4", dict(GEMM_BIG=128, GEMM_BIG_NS=4)), ("auto", dict(GEMM_BIG=1))]
    print("== gemm_nt (M, N, K) bf16 operands, bias + ReLU epilogue: us per launch (TF/s) [max abs deviation from fp32 torch]")
    g = torch.Generator().manual_seed(1)
    for M, N, K, od in shapes:
        A = torch.randn(M, K, generator=g).to(D).bfloat16()
        B = (torch.randn(N, K, generator=g) * K ** -0.5).to(D).bfloat16()
        bias = torch.randn(N, generator=g).to(D) if od == "bf16" else None
        out = torch.empty(M, N, device=D, dtype=torch.bfloat16 if od == "bf16" else torch.float32)
        ref = A.float() @ B.float().t()
        if bias is not None:
            ref = (ref + bias).relu()
        row = []
        for name, tv in variants:
            for kk in ("GEMM_BIG", "GEMM_BIG_NS"):
                L.set_tuning(kk, tv.get(kk))
            out.zero_()
            ops.gemm_nt(A, B, out=out, bias=bias, relu=bias is not None)
            err = (out.float() - ref).abs().max().item()
            us = timeit(lambda: ops.gemm_nt(A, B, out=out, bias=bias, relu=bias is not None))
            row.append("%s %5.1f (%4.0f) [%.0e]" % (name, us, 2.0 * M * N * K / us / 1e6, err))
        print("  %5d %5d %5d %s : %s" % (M, N, K, od, " | ".join(row)))
    for kk in ("GEMM_BIG", "GEMM_BIG_NS"):
        L.set_tuning(kk, None)


if __name__ == "__main__":
    main()
