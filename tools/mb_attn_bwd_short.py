"""Eager back-to-back timing of the attention backward at the benchmark model's three shapes: the two-half launch (ATTN_BWD_FUSED = 0)
against the one-pass workgroup per (batch, head) (1).  delta is handed in (asr_gemm_nn_rowdot computes it in the step)."""
import sys, torch
sys.path.insert(0, "end2end-asr-pytorch_amd")
from asr_hip import lib as L, ops
B, H, d = 32, 8, 64
D = "cuda"
g = torch.Generator().manual_seed(1)
for name, Tq, Tk, causal, pad in (("encoder self 200 x 200", 200, 200, False, False), ("decoder cross 100 x 200", 100, 200, False, False),
                                  ("decoder self 100 x 100 causal + pad", 100, 100, True, True)):
    q, do = (torch.randn(B, Tq, H * d, generator=g).to(D).bfloat16() for _ in range(2))
    k, v = (torch.randn(B, Tk, H * d, generator=g).to(D).bfloat16() for _ in range(2))
    kl = None if pad else torch.randint(Tk // 2, Tk + 1, (B,), generator=g).to(torch.int32).to(D)
    kp = None
    if pad:
        kp = torch.zeros(B, Tk, dtype=torch.uint8)
        for b in range(B):
            kp[b, Tk - 1 - b:] = 1
        kp = kp.to(D)
    o32 = torch.empty(B, Tq, H * d, device=D)
    o, lse, _ = ops.attn_fwd(q, k, v, H, d, key_len=kl, key_pad=kp, causal=causal, scale=0.125, p=0.1, seed=3, o32=o32)
    delta = torch.zeros(B, H, Tq, device=D)
    res = {}
    for fused in (0, 1, 0, 1):
        L.set_tuning("ATTN_BWD_FUSED", fused)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.attn_bwd(q, k, v, o, do, lse, H, d, key_len=kl, key_pad=kp, causal=causal, scale=0.125, p=0.1, seed=3, o32=o32, delta=delta)
        e0.record()
        for _ in range(50):
            ops.attn_bwd(q, k, v, o, do, lse, H, d, key_len=kl, key_pad=kp, causal=causal, scale=0.125, p=0.1, seed=3, o32=o32, delta=delta)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(fused, []).append(round(e0.elapsed_time(e1) * 1e3 / 50, 1))
    print("%-38s two halves %s us, one pass %s us" % (name, res[0], res[1]))
L.set_tuning("ATTN_BWD_FUSED", None)
