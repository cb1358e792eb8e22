"""Attention at the north-star shape (B*H = 256, T = 800, d = 64, bf16): this library against torch's scaled_dot_product_attention
(the flash kernels the ROCm build ships), forward and backward, dropout 0 and 0.1.  Development tool."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "end2end-asr-pytorch_amd"))
from asr_hip import ops  # noqa: E402

D = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
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
    for B, H, T, d in ((32, 8, 800, 64), (32, 8, 200, 64), (16, 8, 795, 64)):
        for p in (0.0, 0.1):
            q = torch.randn(B, T, H * d, device=D).bfloat16()
            k = torch.randn(B, T, H * d, device=D).bfloat16()
            v = torch.randn(B, T, H * d, device=D).bfloat16()
            do = torch.randn(B, T, H * d, device=D).bfloat16()
            kw = dict(scale=0.125, p=p, seed=1234)
            tf = timeit(lambda: ops.attn_fwd(q, k, v, H, d, **kw))
            o, lse, _ = ops.attn_fwd(q, k, v, H, d, **kw)
            tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d, **kw))
            qh, kh, vh = (t.view(B, T, H, d).transpose(1, 2).detach().requires_grad_() for t in (q, k, v))
            doh = do.view(B, T, H, d).transpose(1, 2)
            try:
                sf = timeit(lambda: F.scaled_dot_product_attention(qh, kh, vh, dropout_p=p))
                out = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=p)
                sb = timeit(lambda: torch.autograd.grad(out, (qh, kh, vh), doh, retain_graph=True))
            except Exception as e:
                sf = sb = float("nan")
                print("  sdpa failed:", repr(e)[:150])
            print("(B,H,T,d) = %s p = %.1f | ours fwd %7.1f bwd %7.1f us | torch sdpa fwd %7.1f bwd %7.1f us" % ((B, H, T, d), p, tf, tb, sf, sb))


if __name__ == "__main__":
    main()
