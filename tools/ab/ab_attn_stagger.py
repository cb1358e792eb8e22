"""A/B of the long-sequence attention forward's launch-structure switches: ATTN_PP_STAGGER (x 64 cycles start delay for every other
workgroup of a CU), ATTN_PP_STAGGER_SEL (which block-id bit picks them), ATTN_PP_PRIO.  (Round 3 also measured, through switches that
have since been removed: S(t+1) issued inside the softmax stream of tile t -- 71.0 vs 71.8 us, worse with dropout -- and 8-wave
workgroups, plain 79.4 us and as a two-barrier ping-pong 101 us against 72.5: profiles/r03_attention_pp_*.txt.)
usage: python tools/ab/ab_attn_stagger.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end2end-asr-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "ab"))
from asr_hip import lib as L  # noqa: E402
from asr_hip import ops  # noqa: E402
import ab_attn_pp as AB  # noqa: E402

D = torch.device("cuda")
KEYS = ("ATTN_PP", "ATTN_PP_STAGGER", "ATTN_PP_STAGGER_SEL", "ATTN_PP_PRIO")


def setk(**kw):
    for k in KEYS:
        L.set_tuning(k, kw.get(k))


def main():
    variants = [("v1", dict(ATTN_PP=0))]
    variants += [("pp stag0", dict(ATTN_PP=1, ATTN_PP_STAGGER=0))]
    for sel in (0, 1):
        for st in (4, 8, 12, 16, 24):
            variants.append(("stag%d/b%d" % (st, 3 if sel else 8), dict(ATTN_PP=1, ATTN_PP_STAGGER=st, ATTN_PP_STAGGER_SEL=sel)))
    variants += [("default", dict(ATTN_PP=1)), ("default prio1", dict(ATTN_PP=1, ATTN_PP_PRIO=1))]
    print("== forward time (us) and % of the 2.5 PF dense bf16 peak")
    for B, H, Tq, Tk, p in [(32, 8, 800, 800, 0.0), (32, 8, 800, 800, 0.1), (16, 8, 795, 795, 0.1), (8, 8, 2048, 2048, 0.0)]:
        q = torch.randn(B, Tq, H * 64, device=D).bfloat16()
        k = torch.randn(B, Tk, H * 64, device=D).bfloat16()
        v = torch.randn(B, Tk, H * 64, device=D).bfloat16()
        kl = torch.full((B,), Tk, device=D, dtype=torch.int32)
        fl = 4.0 * B * H * Tq * Tk * 64
        row = []
        for name, tv in variants:
            setk(**tv)
            us = AB.timeit(lambda: ops.attn_fwd(q, k, v, H, 64, key_len=kl, scale=0.125, p=p, seed=5))
            row.append("%s %5.1f %4.1f%%" % (name, us, fl / us / 25e6))
        print("  (%d,%d,%d,%d) p=%.1f : %s" % (B, H, Tq, Tk, p, " | ".join(row)))
    setk()


if __name__ == "__main__":
    main()
