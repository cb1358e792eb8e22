"""Attention backward at one shape, timed per part (delta / dQ / dK+dV / all in one call).  Development tool.
usage: python tools/mb_attn_bwd.py [B H Tq Tk p]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "end2end-asr-pytorch_amd"))
from asr_hip import lib as L, ops  # noqa: E402

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
    a = sys.argv[1:]
    B, H, Tq, Tk = (int(x) for x in a[:4]) if len(a) >= 4 else (32, 8, 800, 800)
    p = float(a[4]) if len(a) > 4 else 0.1
    d = 64
    ops.set_compute_dtype(torch.bfloat16)
    q = torch.randn(B, Tq, H * d, device=D).bfloat16()
    k = torch.randn(B, Tk, H * d, device=D).bfloat16()
    v = torch.randn(B, Tk, H * d, device=D).bfloat16()
    do = torch.randn(B, Tq, H * d, device=D).bfloat16()
    kl = torch.full((B,), Tk, device=D, dtype=torch.int32)
    o, lse, _ = ops.attn_fwd(q, k, v, H, d, key_len=kl, scale=0.125, p=p, seed=1234)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty((B, H, Tq), device=D, dtype=torch.float32)
    qs, ks, vs, os_ = (ops._bt_strides(t, H, d) for t in (q, k, v, o))

    def launch(parts):
        L.call("asr_attn_bwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(None), L.ptr(do), L.ptr(lse), L.ptr(delta), L.ptr(dq),
               L.ptr(dk), L.ptr(dv), B, H, Tq, Tk, d, qs[0], qs[1], ks[0], ks[1], vs[0], vs[1], os_[0], os_[1],
               L.ptr(kl), L.ptr(None), 0, 0, 0, 0.125, p, 1234, L.ptr(None), parts, L.dt(q), L.stream())

    launch(L.ATTN_ALL)
    fl = 4.0 * B * H * Tq * Tk * d
    res = []
    for name, parts, f in (("delta", L.ATTN_DELTA, 0), ("dq", L.ATTN_DQ, 1.0), ("dkv", L.ATTN_DKV, 1.5), ("dq+dkv", L.ATTN_DQ | L.ATTN_DKV, 2.5)):
        us = timeit(lambda: launch(parts))
        res.append("%s %.1f us%s" % (name, us, (" (%.0f TF/s)" % (f * fl / us / 1e6)) if f else ""))
    print("attn bwd (%d,%d,%d,%d,64) p=%.1f: " % (B, H, Tq, Tk, p) + " | ".join(res))


if __name__ == "__main__":
    main()
