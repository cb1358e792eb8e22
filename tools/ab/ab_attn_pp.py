"""A/B + blind diagnostics of the long-sequence attention forward (csrc/attention_pp.hip) against the first-generation kernel
(csrc/attention_fast.hip) and a torch fp32 reference.  Development tool:  python tools/ab/ab_attn_pp.py [diag|time|all]

diag: small shapes that isolate one mechanism each (one tile / several tiles / chunk mode / tail mode / ragged key length / dropout
mask identity), error split by where it shows (lse = first contraction + softmax; O = second contraction) and by lane / block.
time: the north-star shape and configs[3]'s shapes under ATTN_PP = 0 | 1, with the tail / priority switches."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
from asr_hip import lib as L  # noqa: E402
from asr_hip import ops  # noqa: E402

D = torch.device("cuda")


def ref(q, k, v, H, d, key_len):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    qh = q.float().view(B, Tq, H, d).permute(0, 2, 1, 3)
    kh = k.float().view(B, Tk, H, d).permute(0, 2, 1, 3)
    vh = v.float().view(B, Tk, H, d).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(d)
    if key_len is not None:
        dead = torch.arange(Tk, device=q.device)[None, :] >= key_len[:, None].long()
        s = s.masked_fill(dead[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)                       # (B,H,Tq)
    o = (torch.softmax(s, dim=-1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, H * d)
    return o, lse


def run(B, H, Tq, Tk, key_len=None, seed=0, p=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = 64
    q = torch.randn(B, Tq, H * d, generator=g).to(D).bfloat16()
    k = torch.randn(B, Tk, H * d, generator=g).to(D).bfloat16()
    v = torch.randn(B, Tk, H * d, generator=g).to(D).bfloat16()
    kl = torch.tensor(key_len, dtype=torch.int32, device=D) if key_len is not None else None
    o32 = torch.zeros(B, Tq, H * d, device=D)
    o, lse, _ = ops.attn_fwd(q, k, v, H, d, key_len=kl, scale=0.125, p=p, seed=77, o32=o32)
    return q, k, v, kl, o, lse, o32


def diag_case(name, B, H, Tq, Tk, key_len=None):
    q, k, v, kl, o, lse, o32 = run(B, H, Tq, Tk, key_len)
    ro, rl = ref(q, k, v, H, 64, kl)
    eo = (o32 - ro).abs()                       # (B,Tq,H*64)
    el = (lse.view(B, H, Tq) - rl).abs()
    el = torch.where(torch.isfinite(el), el, torch.zeros_like(el))
    eb = (o.float() - ro).abs().max().item()
    print("%-34s O32 err %.3e (bf16 O %.3e)  lse err %.3e   |O| max %.2f" % (name, eo.max().item(), eb, el.max().item(), ro.abs().max().item()))
    if eo.max().item() > 2e-2 or el.max().item() > 2e-2:
        # where: by 32-query block, by query lane (q % 32), by 8-wide d group of a head, by head
        eq = eo.amax(dim=(0, 2))
        nb = (Tq + 31) // 32
        print("   O err by 32-query block :", ["%.1e" % eq[i * 32:(i + 1) * 32].max().item() for i in range(nb)][:28])
        lane = torch.zeros(32)
        for i in range(Tq):
            lane[i % 32] = max(lane[i % 32], eq[i].item())
        print("   O err by query lane     :", ["%.0e" % x for x in lane.tolist()])
        ed = eo.view(B, Tq, H, 8, 8).amax(dim=(0, 1, 2, 4))
        print("   O err by 8-wide d group :", ["%.1e" % x for x in ed.tolist()])
        print("   O err by head           :", ["%.1e" % x for x in eo.view(B, Tq, H, 64).amax(dim=(0, 1, 3)).tolist()])
        lq = el.amax(dim=(0, 1))
        print("   lse err by 32-query blk :", ["%.1e" % lq[i * 32:(i + 1) * 32].max().item() for i in range(nb)][:28])
        print("   first bad rows (b,q): O", (eo.amax(dim=2) > 2e-2).nonzero()[:6].tolist(), " lse", (el.amax(dim=1) > 2e-2).nonzero()[:6].tolist())


def diag():
    print("== diagnostics, ATTN_PP = 1, ATTN_PP_MIN = 1 (every shape goes through attention_pp.hip)")
    L.set_tuning("ATTN_PP", 1)
    L.set_tuning("ATTN_PP_MIN", 1)
    diag_case("tail  Tq=32  Tk=64  (1 tile)", 1, 1, 32, 64)
    diag_case("tail  Tq=32  Tk=256 (4 tiles)", 1, 2, 32, 256)
    diag_case("tail  Tq=32  Tk=832 (13 tiles)", 2, 2, 32, 832)
    diag_case("chunk Tq=128 Tk=64  (1 tile)", 1, 1, 128, 64)
    diag_case("chunk Tq=128 Tk=128 (2 tiles)", 1, 1, 128, 128)
    diag_case("chunk Tq=128 Tk=192 (3 tiles)", 1, 2, 128, 192)
    diag_case("chunk Tq=128 Tk=448 (7 tiles)", 1, 2, 128, 448)
    diag_case("chunk Tq=256 Tk=64  (1 tile)", 1, 1, 256, 64)
    diag_case("chunk Tq=256 Tk=128 (2 tiles)", 1, 2, 256, 128)
    diag_case("chunk Tq=256 Tk=256 (4 tiles)", 1, 2, 256, 256)
    diag_case("chunk Tq=256 Tk=800 ragged tile", 2, 8, 256, 800)
    diag_case("tail  Tq=64  Tk=320 (2 blocks)", 1, 2, 64, 320)
    diag_case("tail  Tq=40  Tk=320", 1, 2, 40, 320)
    diag_case("mixed Tq=800 Tk=800", 2, 8, 800, 800, key_len=[800, 613])
    diag_case("mixed Tq=795 Tk=795", 2, 8, 795, 795, key_len=[795, 402])
    diag_case("tails Tq=100 Tk=795", 2, 8, 100, 795, key_len=[700, 795])
    diag_case("key_len 0 and 1", 2, 2, 160, 200, key_len=[0, 1])
    # spiked scores: the deferred reference must move (and move everything at the old reference exactly once)
    g = torch.Generator().manual_seed(5)
    B, H, Tq, Tk, d = 1, 2, 160, 512, 64
    q = torch.randn(B, Tq, H * d, generator=g)
    k = torch.randn(B, Tk, H * d, generator=g)
    v = torch.randn(B, Tk, H * d, generator=g)
    for t in (70, 200, 333, 500):               # a key aligned with every query of head 0, later tiles: raw score >> the rest
        k[0, t, :d] = q[0, t % Tq, :d] * (2.0 + t / 250.0)
    k[0, 10, d:] = -30 * q[0, 3, d:]            # a hugely NEGATIVE outlier must not matter
    q, k, v = q.to(D).bfloat16(), k.to(D).bfloat16(), v.to(D).bfloat16()
    o32 = torch.zeros(B, Tq, H * d, device=D)
    o, lse, _ = ops.attn_fwd(q, k, v, H, d, scale=0.125, o32=o32)
    ro, rl = ref(q, k, v, H, d, None)
    print("%-34s O32 err %.3e  lse err %.3e" % ("spiked keys (reference moves)", (o32 - ro).abs().max().item(), (lse.view(B, H, Tq) - rl).abs().max().item()))
    # dropout: same mask as the first-generation kernel (same function of (seed, row, key)) and as the probability dump
    q, k, v, kl, o1, lse1, o32a = run(2, 4, 288, 320, p=0.25, seed=3)
    L.set_tuning("ATTN_PP", 0)
    _, _, _, _, o0, lse0, o32b = run(2, 4, 288, 320, p=0.25, seed=3)
    L.set_tuning("ATTN_PP", 1)
    print("%-34s |O32(pp) - O32(v1)| %.3e   lse diff %.3e   (same dropout mask <=> small)" %
          ("dropout p=0.25 vs first generation", (o32a - o32b).abs().max().item(), (lse1 - lse0).abs().max().item()))
    L.set_tuning("ATTN_PP_MIN", None)


def timeit(fn, n=30):
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


def time_all():
    print("== forward time (us), bf16 d=64; flop = 4 B H Tq Tk d; %% of the 2.5 PF dense bf16 peak")
    shapes = [(32, 8, 800, 800, 0.0), (32, 8, 800, 800, 0.1), (16, 8, 795, 795, 0.0), (16, 8, 795, 795, 0.1), (16, 8, 100, 795, 0.1),
              (32, 8, 200, 200, 0.0), (32, 8, 200, 200, 0.1), (32, 8, 100, 200, 0.1), (32, 8, 512, 512, 0.0), (32, 8, 1024, 1024, 0.0),
              (8, 8, 2048, 2048, 0.0)]
    variants = [("v1", dict(ATTN_PP=0)), ("pp", dict(ATTN_PP=1)), ("pp no tails", dict(ATTN_PP=1, ATTN_PP_TAIL=0)),
                ("pp prio1", dict(ATTN_PP=1, ATTN_PP_PRIO=1))]
    for B, H, Tq, Tk, p in shapes:
        q = torch.randn(B, Tq, H * 64, device=D).bfloat16()
        k = torch.randn(B, Tk, H * 64, device=D).bfloat16()
        v = torch.randn(B, Tk, H * 64, device=D).bfloat16()
        kl = torch.full((B,), Tk, device=D, dtype=torch.int32)
        fl = 4.0 * B * H * Tq * Tk * 64
        row = []
        for name, tv in variants:
            for kk in ("ATTN_PP", "ATTN_PP_TAIL", "ATTN_PP_PRIO"):
                L.set_tuning(kk, tv.get(kk))
            L.set_tuning("ATTN_PP_MIN", 1)
            us = timeit(lambda: ops.attn_fwd(q, k, v, H, 64, key_len=kl, scale=0.125, p=p, seed=5))
            row.append("%s %6.1f us %4.1f%%" % (name, us, fl / us / 25e6))
        print("  (%d,%d,%d,%d) p=%.1f : %s" % (B, H, Tq, Tk, p, " | ".join(row)))
    for kk in ("ATTN_PP", "ATTN_PP_TAIL", "ATTN_PP_PRIO", "ATTN_PP_MIN"):
        L.set_tuning(kk, None)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("diag", "all"):
        diag()
    if which in ("time", "all"):
        time_all()
