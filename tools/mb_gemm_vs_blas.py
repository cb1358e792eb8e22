"""Forward / data-gradient GEMM shapes of configs[1] and configs[3]: this library against torch.mm (hipBLASLt / rocBLAS), bf16.
Development tool: says where the hand-written kernels leave time on the table.  python tools/mb_gemm_vs_blas.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "end2end-asr-pytorch_amd"))
from asr_hip import ops  # noqa: E402

D = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ops.set_compute_dtype(torch.bfloat16)
    shapes = [("enc input", 6400, 512, 5120), ("qkv", 6400, 1536, 512), ("out proj", 6400, 512, 512), ("ffn1", 6400, 2048, 512),
              ("ffn2", 6400, 512, 2048), ("dec qkv", 3200, 1536, 512), ("dec ffn1", 3200, 2048, 512), ("vocab", 3200, 4416, 512),
              ("cfg3 qkv", 12720, 1536, 512), ("cfg3 ffn1", 12720, 2048, 512), ("cfg3 ffn2", 12720, 512, 2048), ("cfg3 out", 12720, 512, 512)]
    print("%-10s %6s %5s %5s | nt ours   blas   | nn ours   blas   (us)" % ("shape", "M", "N", "K"))
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=D).bfloat16()
        w = torch.randn(N, K, device=D).bfloat16()
        bias = torch.randn(N, device=D)
        out = torch.empty(M, N, device=D, dtype=torch.bfloat16)
        t_nt = timeit(lambda: ops.gemm_nt(a, w, out=out, bias=bias))
        wt = w.t()
        t_nt_b = timeit(lambda: torch.mm(a, wt, out=out))
        dy = torch.randn(M, N, device=D).bfloat16()
        dx = torch.empty(M, K, device=D, dtype=torch.bfloat16)
        t_nn = timeit(lambda: ops.gemm_nn(dy, w, out=dx)) if ops.gemm_nn_supported(dy, w) else float("nan")
        t_nn_b = timeit(lambda: torch.mm(dy, w, out=dx))
        print("%-10s %6d %5d %5d | %7.1f %7.1f | %7.1f %7.1f" % (name, M, N, K, t_nt, t_nt_b, t_nn, t_nn_b))


if __name__ == "__main__":
    main()
