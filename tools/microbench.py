#!/usr/bin/env python3
"""Kernel micro-benchmarks on the shapes of BASELINE config 2 (B=32): TFLOP/s per launch, measured with HIP events over
back-to-back launches.  Development tool (not part of the test suite):  python tools/microbench.py [gemm|conv|wgrad|attn|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
import torch

from asr_hip import lib as L
from asr_hip import ops

D = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def gemm():
    print("== gemm_nt  (M, N, K)  bf16")
    shapes = [(6400, 512, 512, torch.bfloat16), (6400, 1536, 512, torch.bfloat16), (6400, 2048, 512, torch.bfloat16),
              (6400, 512, 2048, torch.bfloat16), (6400, 512, 5120, torch.bfloat16), (3200, 512, 512, torch.bfloat16),
              (3200, 4364, 512, torch.float32), (3200, 512, 4368, torch.bfloat16), (6400, 5120, 512, torch.bfloat16)]
    for M, N, K, od in shapes:
        A = torch.randn(M, K, device=D).bfloat16()
        B = torch.randn(N, K, device=D).bfloat16()
        bias = torch.randn(N, device=D)
        out = torch.empty(M, N, device=D, dtype=od)
        res = []
        for tile in "012":
            L.set_tuning("GEMM_TILE", int(tile))
            us = timeit(lambda: ops.gemm_nt(A, B, out=out, bias=bias))
            res.append("%s %6.1fus %5.0fTF" % (["128x128", "128x64", "64x64"][int(tile)], us, 2 * M * N * K / us / 1e6))
        L.set_tuning("GEMM_TILE", None)
        print("  fwd   %5d %5d %5d -> %-8s %s" % (M, N, K, str(od)[6:], " | ".join(res)))
    print("== wgrad TN (natural layouts, tr reads)  dW(N,K) over M")
    for N, K, M in [(512, 512, 6400), (2048, 512, 6400), (512, 2048, 6400), (512, 5120, 6400), (4364, 512, 3200), (512, 512, 3200)]:
        dy = torch.randn(M, (N + 63) // 64 * 64, device=D).bfloat16()
        x = torch.randn(M, K, device=D).bfloat16()
        g = torch.zeros(N, K, device=D); gb = torch.zeros(N, device=D)
        res = []
        for sp in (1, 2, 4, 8, 0):
            us = timeit(lambda: ops.gemm_tn(dy, x, g, colsum_acc=gb, N=N, K=K, splits=sp))
            res.append("s%d %6.1fus %5.0fTF" % (sp, us, 2 * M * N * K / us / 1e6))
        print("  tn    %5d %5d %5d : %s" % (N, K, M, " | ".join(res)))
    print("== wgrad (split-K, fp32 atomics)  dW(N,K) over M")
    for N, K, M in [(512, 512, 6400), (2048, 512, 6400), (512, 2048, 6400), (512, 5120, 6400), (4364, 512, 3200), (1536, 512, 6400)]:
        dyt = torch.randn(N, M, device=D).bfloat16()
        xt = torch.randn(K, M, device=D).bfloat16()
        g = torch.zeros(N, K, device=D)
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        for tile in "012":
            L.set_tuning("GEMM_TILE", int(tile))
            res = []
            for sp in (1, 2, 4, 8):
                us = timeit(lambda: ops.gemm_nt(dyt, xt, out=g, accumulate=True, splits=sp))
                res.append("s%d %6.1fus %5.0fTF" % (sp, us, 2 * M * N * K / us / 1e6))
            print("  wgrad %5d %5d %5d tile %s : %s" % (N, K, M, ["128x128", "128x64", "64x64"][int(tile)], " | ".join(res)))
        L.set_tuning("GEMM_TILE", None)


def conv():
    print("== conv3x3 igemm (B,H,W,Cin,Cout) bf16")
    for B, H, W, Ci, Co in [(32, 161, 800, 64, 64), (32, 80, 400, 64, 128), (32, 80, 400, 128, 128), (32, 80, 400, 128, 64)]:
        x = torch.randn(B, H, W, Ci, device=D).bfloat16()
        wk = torch.randn(Co, 9, Ci, device=D).bfloat16()
        bias = torch.randn(Co, device=D)
        us = timeit(lambda: ops.conv3x3(x, wk, bias, Co, relu=True), iters=10)
        fl = 2 * 9 * Ci * Co * B * H * W
        print("  igemm %s %8.1f us  %7.1f TF/s" % ((B, H, W, Ci, Co), us, fl / us / 1e6))
        msk = torch.randn(B, H, W, Co, device=D).bfloat16()
        us = timeit(lambda: ops.conv3x3(x, wk, None, Co, relu=False, mask_src=msk), iters=10)
        print("  igemm+mask %s %8.1f us  %7.1f TF/s" % ((B, H, W, Ci, Co), us, fl / us / 1e6))


def wgrad():
    print("== conv3x3 wgrad (B,H,W,Cin,Cout) bf16, NHWC native kernel + two-stage dW reduction")
    for B, H, W, Ci, Co in [(32, 161, 800, 64, 64), (32, 80, 400, 64, 128), (32, 80, 400, 128, 128)]:
        x = torch.randn(B, H, W, Ci, device=D).bfloat16()
        dy = torch.randn(B, H, W, Co, device=D).bfloat16()
        dw = torch.zeros(Co, Ci, 3, 3, device=D)
        dbb = torch.zeros(Co, device=D)
        us_n = timeit(lambda: ops.conv3x3_wgrad_nhwc(x, dy, dw, dbb), iters=10)
        print("  wgrad-NHWC %s %8.1f us  %7.1f TF/s" % ((B, H, W, Ci, Co), us_n, 2 * 9 * Ci * Co * B * H * W / us_n / 1e6))


ATTN_CASES = [(32, 8, 200, 200, 64, False, 0.0), (32, 8, 200, 200, 64, False, 0.1), (32, 8, 100, 100, 64, True, 0.1),
              (32, 8, 100, 200, 64, False, 0.1), (32, 8, 800, 800, 64, False, 0.0), (32, 8, 800, 800, 64, False, 0.1),
              (16, 8, 795, 795, 64, False, 0.1)]


def attn(cases=None):
    print("== attention (B,H,Tq,Tk,d) bf16   [p = dropout, len = key_len mask as in the encoder]")
    for B, H, Tq, Tk, d, causal, pd in cases or ATTN_CASES:
        q = torch.randn(B, Tq, H * d, device=D).bfloat16()
        k = torch.randn(B, Tk, H * d, device=D).bfloat16()
        v = torch.randn(B, Tk, H * d, device=D).bfloat16()
        do = torch.randn(B, Tq, H * d, device=D).bfloat16()
        kl = None if causal else torch.full((B,), Tk, device=D, dtype=torch.int32)
        kw = dict(causal=causal, scale=0.125, p=pd, seed=1234, key_len=kl)
        us = timeit(lambda: ops.attn_fwd(q, k, v, H, d, **kw))
        o, lse, _ = ops.attn_fwd(q, k, v, H, d, **kw)
        fl = 4 * B * H * Tq * Tk * d * (0.5 if causal else 1.0)
        usb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d, **kw))
        print("  attn %s p=%.1f fwd %8.1f us %7.1f TF/s (%4.1f%% of 2.5 PF) | bwd %8.1f us %7.1f TF/s (%4.1f%%)" %
              ((B, H, Tq, Tk, d, causal), pd, us, fl / us / 1e6, fl / us / 25e6, usb, 2.5 * fl / usb / 1e6, 2.5 * fl / usb / 25e6))


def tn_ab():
    """Weight-gradient GEMM dW(N,K) = dY^T X over M rows, automatic split: single-stage kernels (TN_PIPE=0) vs the pipelined
    128 x 128 kernel with 3 / 4 LDS stages."""
    print("== wgrad TN, automatic split: TN_PIPE = 0 (single-stage 64x64 / 128x128 kernels) | 3 | 4 stages")
    for N, K, M in [(512, 512, 6400), (1536, 512, 6400), (2048, 512, 6400), (512, 2048, 6400), (512, 5120, 6400), (4364, 512, 3200),
                    (512, 512, 3200), (1024, 512, 6400), (512, 512, 12720), (2048, 512, 12720), (512, 2048, 12720), (1536, 512, 12720)]:
        dy = torch.randn(M, (N + 63) // 64 * 64, device=D).bfloat16()
        x = torch.randn(M, K, device=D).bfloat16()
        g = torch.zeros(N, K, device=D); gb = torch.zeros(N, device=D)
        res = []
        for pipe in (0, 3, 4):
            L.set_tuning("TN_PIPE", pipe)
            us = timeit(lambda: ops.gemm_tn(dy, x, g, colsum_acc=gb, N=N, K=K))
            res.append("pipe%d %6.1fus %5.0fTF" % (pipe, us, 2 * M * N * K / us / 1e6))
        L.set_tuning("TN_PIPE", None)
        print("  tn    %5d %5d %5d : %s" % (N, K, M, " | ".join(res)))


def misc():
    print("== streaming kernels")
    M, Dm = 6400, 512
    y = torch.randn(M, Dm, device=D).bfloat16(); r = torch.randn(M, Dm, device=D).bfloat16()
    g = torch.ones(Dm, device=D); b = torch.zeros(Dm, device=D)
    us = timeit(lambda: ops.add_ln_fwd(y, r, g, b, p=0.1, seed=1))
    print("  add_ln_fwd 6400x512 %7.1f us  %.0f GB/s" % (us, 4 * M * Dm * 2 / us / 1e3))
    out, mean, rstd = ops.add_ln_fwd(y, r, g, b, p=0.1, seed=1)
    dg = torch.zeros(Dm, device=D); db = torch.zeros(Dm, device=D)
    us = timeit(lambda: ops.add_ln_bwd(out, y, mean, rstd, g, None, dg, db, p=0.1, seed=1))
    print("  add_ln_bwd 6400x512 %7.1f us  %.0f GB/s" % (us, 4 * M * Dm * 2 / us / 1e3))
    for Mx in (1600, 25600, 102400):          # latency floor vs streaming rate of the LayerNorm kernels
        yy = torch.randn(Mx, Dm, device=D).bfloat16(); rr = torch.randn(Mx, Dm, device=D).bfloat16()
        us = timeit(lambda: ops.add_ln_fwd(yy, rr, g, b, p=0.1, seed=1))
        print("  add_ln_fwd %dx512 %7.1f us  %.0f GB/s" % (Mx, us, 4 * Mx * Dm * 2 / us / 1e3))
    x = torch.randn(M, 2048, device=D).bfloat16()
    us = timeit(lambda: ops.transpose_padded(x))
    print("  transpose 6400x2048 %7.1f us  %.0f GB/s" % (us, 2 * x.numel() * 2 / us / 1e3))
    x1 = torch.randn(32, 161, 800, 64, device=D).bfloat16()
    us = timeit(lambda: ops.maxpool_fwd(x1), iters=10)
    print("  maxpool_fwd (32,161,800,64) %7.1f us  %.0f GB/s" % (us, 1.25 * x1.numel() * 2 / us / 1e3))
    src = torch.randn(32, 1, 161, 800, device=D); w = torch.randn(64, 1, 3, 3, device=D); bb = torch.randn(64, device=D)
    us = timeit(lambda: ops.conv1_fwd(src, w, bb, torch.bfloat16), iters=10)
    print("  conv1_fwd %7.1f us  %.0f GB/s" % (us, x1.numel() * 2 / us / 1e3))
    dw = torch.zeros(64, 1, 3, 3, device=D); db = torch.zeros(64, device=D)
    us = timeit(lambda: ops.conv1_wgrad(src, x1, dw, db), iters=10)
    print("  conv1_wgrad %7.1f us  %.0f GB/s" % (us, x1.numel() * 2 / us / 1e3))


def decode():
    """Greedy decode, B = 32 utterances x 300 steps over T' = 200 encoder frames (configs[1] model): reference-style full re-run,
    KV-cached eager loop, KV-cached hipGraph replay."""
    import time
    sys.path.insert(0, ROOT)
    from utils import constant
    from utils.functions import init_transformer_model
    flags = ["--num-layers", "4", "--num-heads", "8", "--dim-model", "512", "--dim-key", "64", "--dim-value", "64", "--dim-inner",
             "2048", "--dim-emb", "512", "--feat_extractor", "vgg_cnn", "--tgt-max-len", "301", "--src-max-len", "800",
             "--dropout", "0.1", "--precision", "bf16", "--cuda"]
    args = constant.parse(flags)
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x4E00 + i) for i in range(4361)]
    l2i = {c: i for i, c in enumerate(chars)}
    model = init_transformer_model(args, l2i, {i: c for c, i in l2i.items()}).cuda().eval()
    enc = torch.randn(32, 200, 512, device=D)
    print("== greedy decode, 32 utterances x 300 steps, 4-layer d512 decoder over 200 encoder frames (bf16)")
    modes = (True,) if os.environ.get("MICRO_DECODE_GRAPH_ONLY") == "1" else (True, "graph", "eager", False)
    for mode in modes:
        with torch.no_grad():
            model.decoder.greedy_search(enc, use_cache=mode)
            torch.cuda.synchronize()
            t0 = time.time()
            model.decoder.greedy_search(enc, use_cache=mode)
            torch.cuda.synchronize()
        print("  %-36s %8.1f ms" % ({True: "KV cache + hipGraph, 30-launch step", "graph": "KV cache + hipGraph, op per launch", "eager": "KV cache, eager launches", False: "full re-run (reference)"}[mode],
                                      (time.time() - t0) * 1e3))
    if os.environ.get("MICRO_DECODE_BEAM", "1") == "1":
        print("== beam search (width 4), the same 32 utterances (random weights: every hypothesis runs to the 200-frame limit)")
        for mode, label in ((True, "all utterances in one decoder batch"), ("per_utterance", "loop over utterances")):
            with torch.no_grad():
                torch.cuda.synchronize()
                t0 = time.time()
                model.decoder.beam_search(enc, beam_width=4, nbest=1, c_weight=0.1, use_cache=mode)
                torch.cuda.synchronize()
            print("  %-36s %8.1f ms" % (label, (time.time() - t0) * 1e3))



if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for name, fn in (("gemm", gemm), ("conv", conv), ("wgrad", wgrad), ("attn", attn), ("misc", misc)):
        if which in (name, "all"):
            fn()
    if which == "decode":
        decode()
    if which == "tn":
        tn_ab()
    if which == "attn800":          # the north-star shape only (encoder self-attention, T = 800, bs 32, dropout 0.1)
        attn([ATTN_CASES[5]])
