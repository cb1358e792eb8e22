#!/usr/bin/env python3
"""Where does a gradient error enter?  Development tool (GPU box): runs a BASELINE-shape case (tests/golden/<case>.npz) in
the product model and in the fp64 oracle with autograd hooks on every sub-layer output on both sides, and prints the
relative L2 error of d(loss)/d(sub-layer output) in backward order, then the same for the vgg_cnn stack's internal
tensors (the product's VGGFn.backward replayed step by step against fp64 torch).

usage: python tools/diag_fp32.py [case] [fp32|bf16]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "end2end-asr-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.nn.functional as F

import big_cases as BC
from oracle import asr_oracle as O


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_b2"
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    z = BC.load(os.path.join(ROOT, "tests", "golden"), name)
    args, model, l2i, i2l = BC.build_product(z, precision, True)
    src, src_len, tgt = BC.batch(z)
    model = model.cuda().train()
    from utils.functions import init_optimizer
    from utils.metrics import calculate_metrics
    opt = init_optimizer(args, model, "noam")
    ours = {}

    def hook_module(tag, m):
        def fwd_hook(mod, inp, out):
            t = out[0] if isinstance(out, tuple) else out
            if t.requires_grad:
                t.register_hook(lambda g, tag=tag: ours.__setitem__(tag, g.detach().float().cpu()))
        m.register_forward_hook(fwd_hook)

    core = model.module if hasattr(model, "module") else model
    for i, l in enumerate(core.encoder.layers):
        hook_module("encoder.layers.%d.self_attn." % i, l.self_attn)
        hook_module("encoder.layers.%d.pos_ffn." % i, l.pos_ffn)
    for i, l in enumerate(core.decoder.layers):
        hook_module("decoder.layers.%d.self_attn." % i, l.self_attn)
        hook_module("decoder.layers.%d.encoder_attn." % i, l.encoder_attn)
        hook_module("decoder.layers.%d.pos_ffn." % i, l.pos_ffn)
    orig_feat = core._features

    def feat_hook(x):
        f = orig_feat(x)
        if f.requires_grad:
            f.register_hook(lambda g: ours.__setitem__("feats", g.detach().float().cpu()))
        return f
    core._features = feat_hook
    opt.zero_grad()
    pred, gold, hyp, _ = model(src.cuda(), src_len, tgt.cuda())
    loss, _ = calculate_metrics(pred, gold, smoothing=float(z["smoothing"]), loss_type="ce")
    loss.backward()
    torch.cuda.synchronize()

    # ---- oracle fp64 with the same hooks
    truth = {}
    mha0, ffn0, cfe0 = O.multi_head_attention, O.pos_ffn, O.conv_front_end

    def mha(w, p, *a, **k):
        out = mha0(w, p, *a, **k)
        t = out[0] if isinstance(out, tuple) else out
        if t.requires_grad:
            t.register_hook(lambda g, p=p: truth.__setitem__(p, g.detach()))
        return out

    def ffn(w, p, x):
        out = ffn0(w, p, x)
        if out.requires_grad:
            out.register_hook(lambda g, p=p: truth.__setitem__(p, g.detach()))
        return out

    def cfe(*a, **k):
        out = cfe0(*a, **k)
        if out.requires_grad:
            out.register_hook(lambda g: truth.__setitem__("feats", g.detach()))
        return out
    O.multi_head_attention, O.pos_ffn, O.conv_front_end = mha, ffn, cfe
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    w = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    w64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in w.items()}
    r64 = O.train_step(w64, BC.oracle_cfg(z), src.double(), src_len, tgt, float(z["smoothing"]), bn_state={})
    O.multi_head_attention, O.pos_ffn, O.conv_front_end = mha0, ffn0, cfe0
    print("== %s %s: d(loss)/d(sub-layer output), relative L2 error vs fp64 oracle (backward order)" % (name, precision))
    order = []
    nd, ne = len(core.decoder.layers), len(core.encoder.layers)
    for i in range(nd - 1, -1, -1):
        order += ["decoder.layers.%d.pos_ffn." % i, "decoder.layers.%d.encoder_attn." % i, "decoder.layers.%d.self_attn." % i]
    for i in range(ne - 1, -1, -1):
        order += ["encoder.layers.%d.pos_ffn." % i, "encoder.layers.%d.self_attn." % i]
    order += ["feats"]
    for k in order:
        if k in ours and k in truth:
            t = truth[k].reshape(ours[k].shape) if truth[k].numel() == ours[k].numel() else truth[k]
            print("  %-40s %.3e   |g| %.3e" % (k, rel(ours[k], t), float(t.norm())))
        else:
            print("  %-40s missing (ours %s, truth %s)" % (k, k in ours, k in truth))
    g = {k: q.grad.detach().float().cpu() for k, q in model.named_parameters()}
    for k in ("decoder.trg_embedding.weight", "decoder.layers.0.pos_ffn.conv_1.bias", "decoder.layers.0.pos_ffn.conv_2.bias",
              "conv.0.weight", "conv.0.bias", "conv.2.weight"):
        if k in g:
            a, b = g[k].double().reshape(-1), r64["grads"][k].double().reshape(-1)
            d = (a - b).abs()
            top = torch.topk(d, min(5, d.numel()))
            print("  param %-45s rel %.3e ; worst elements: %s" % (k, rel(a, b), [(int(i), float(a[i]), float(b[i])) for i in top.indices]))

    if getattr(core, "feat_extractor", "") == "vgg_cnn":
        conv_chain(core, src, ours.get("feats"), precision)


def conv_chain(core, src, dfeat, precision):
    """VGGFn forward + backward replayed op by op (asr_hip.ops) against fp64 torch."""
    from asr_hip import ops
    from asr_hip import params as P
    c = core.conv
    x = src.cuda().contiguous().float()
    cd = ops.compute_dtype()
    w0, b0, w2, b2, w5, b5, w7, b7 = c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[5].weight, c[5].bias, c[7].weight, c[7].bias
    y1 = ops.conv1_fwd(x, w0.data, b0.data, cd)
    wk2, wd2 = P.conv_shadow(w2)
    y2, p1 = ops.conv3x3_relu_pool(y1, wk2, b2.data, w2.shape[0])
    wk5, wd5 = P.conv_shadow(w5)
    y3 = ops.conv3x3(p1, wk5, b5.data, w5.shape[0], relu=True)
    wk7, wd7 = P.conv_shadow(w7)
    y4 = ops.conv3x3(y3, wk7, b7.data, w7.shape[0], relu=True)
    out = ops.maxpool_fwd(y4, tcf=True)
    # fp64 reference
    xd = src.double()
    W = [t.detach().double().cpu() for t in (w0, b0, w2, b2, w5, b5, w7, b7)]
    t1 = F.relu(F.conv2d(xd, W[0], W[1], padding=1)); t1.requires_grad_()
    t2 = F.relu(F.conv2d(t1, W[2], W[3], padding=1)); t2.retain_grad()
    q1 = F.max_pool2d(t2, 2, 2); q1.retain_grad()
    t3 = F.relu(F.conv2d(q1, W[4], W[5], padding=1)); t3.retain_grad()
    t4 = F.relu(F.conv2d(t3, W[6], W[7], padding=1)); t4.retain_grad()
    q2 = F.max_pool2d(t4, 2, 2)
    B, C, Fq, T = q2.shape
    ref_out = q2.reshape(B, C * Fq, T).transpose(1, 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    print("== vgg_cnn forward, relative L2 error vs fp64 torch")
    for tag, a, b in (("y1", y1, nhwc(t1)), ("y2", y2, nhwc(t2)), ("p1", p1, nhwc(q1)), ("y3", y3, nhwc(t3)), ("y4", y4, nhwc(t4)),
                      ("out", out, ref_out)):
        print("  %-6s %.3e" % (tag, rel(a, b)))
    g = torch.Generator().manual_seed(3)
    dout = dfeat if dfeat is not None else torch.randn(out.shape, generator=g)
    dout = dout.to(out.dtype)
    ref_out.backward(dout.double().cpu())
    dy4 = ops.maxpool_bwd(y4, dout.cuda().contiguous(), tcf=True)
    dy3 = ops.conv3x3(dy4, wd7, None, w7.shape[1], relu=False, mask_src=y3)
    dp1 = ops.conv3x3(dy3, wd5, None, w5.shape[1], relu=False)
    dy2 = ops.maxpool_bwd(y2, dp1)
    dy1 = ops.conv3x3(dy2, wd2, None, w2.shape[1], relu=False, mask_src=y1)
    # reference gradients w.r.t. the PRE-activation of each conv = grad of the ReLU output masked by (output > 0)
    pre = lambda t: nhwc(t.grad * (t > 0))
    print("== vgg_cnn backward (gradient w.r.t. each conv's pre-activation / pool input), relative L2 error vs fp64 torch")
    for tag, a, b in (("dy4", dy4, pre(t4)), ("dy3", dy3, pre(t3)), ("dp1", dp1, nhwc(q1.grad)), ("dy2", dy2, pre(t2)),
                      ("dy1", dy1, pre(t1))):
        d = (a.detach().double().cpu() - b).abs()
        print("  %-6s %.3e   max abs err %.3e (max |ref| %.3e) ; elements off by > 1e-3 max: %d of %d" %
              (tag, rel(a, b), float(d.max()), float(b.abs().max()), int((d > 1e-3 * float(b.abs().max())).sum()), d.numel()))
    d4 = (dy4.detach().double().cpu() - pre(t4)).abs()
    bad = (d4 > 1e-3 * float(pre(t4).abs().max())).nonzero()
    print("== mismatching dy4 elements (b, f, t, c): %d ; first ones %s" % (bad.shape[0], bad[:12].tolist()))
    for bb in range(dy4.shape[0]):
        tt = bad[bad[:, 0] == bb][:, 2]
        print("   batch %d: %d elements, t range %s..%s" % (bb, tt.numel(), int(tt.min()) if tt.numel() else None, int(tt.max()) if tt.numel() else None))
    for bi in bad[:6].tolist():
        b_, f_, t_, c_ = bi
        f0, t0 = f_ // 2 * 2, t_ // 2 * 2
        print("   window of", bi, "ours", y4[b_, f0:f0 + 2, t0:t0 + 2, c_].flatten().tolist(), "fp64", t4[b_, c_, f0:f0 + 2, t0:t0 + 2].flatten().tolist())
    # ---- the same backward in fp64 but with the PRODUCT's discrete decisions (its ReLU masks and pooling arg-maxes):
    #      what is left is the arithmetic error of the backward kernels alone
    nchw = lambda t: t.detach().double().cpu().permute(0, 3, 1, 2).contiguous()
    Y1, Y2, Y3, Y4 = nchw(y1), nchw(y2), nchw(y3), nchw(y4)

    def unpool(g, y):
        _, idx = F.max_pool2d(y, 2, 2, return_indices=True)
        return F.max_unpool2d(g, idx, 2, 2, output_size=y.shape[-2:]) * (y > 0)

    def dgrad(g, w):
        return torch.nn.grad.conv2d_input(g.shape[:1] + (w.shape[1],) + g.shape[2:], w, g, padding=1)
    Bq, Tq, CF = dout.shape
    g_out = dout.double().cpu().transpose(1, 2).reshape(Bq, Y4.shape[1], CF // Y4.shape[1], Tq)
    r4 = unpool(g_out, Y4)
    r3 = dgrad(r4, W[6]) * (Y3 > 0)
    rp1 = dgrad(r3, W[4])
    r2 = unpool(rp1, Y2)
    r1 = dgrad(r2, W[2]) * (Y1 > 0)
    print("== vgg_cnn backward against fp64 WITH THE PRODUCT'S OWN ReLU masks / pooling arg-maxes (arithmetic error only)")
    for tag, a, b in (("dy4", dy4, r4), ("dy3", dy3, r3), ("dp1", dp1, rp1), ("dy2", dy2, r2), ("dy1", dy1, r1)):
        print("  %-6s %.3e" % (tag, rel(a, nhwc(b))))
    Pq = nchw(p1)
    for tag, xin, g_, w_, b_ in (("conv.7", Y3, r4, w7, b7), ("conv.5", Pq, r3, w5, b5), ("conv.2", Y1, r2, w2, b2)):
        dw = torch.zeros_like(w_.data); db = torch.zeros_like(b_.data)
        ours_x = {"conv.7": y3, "conv.5": p1, "conv.2": y1}[tag]
        ours_g = {"conv.7": dy4, "conv.5": dy3, "conv.2": dy2}[tag]
        ops.conv3x3_wgrad_nhwc(ours_x, ours_g, dw, db)
        ref_w = torch.nn.grad.conv2d_weight(xin, w_.shape, g_, padding=1)
        print("  %-6s dW %.3e  db %.3e   (ours on OUR dy vs fp64 on the fp64 dy)" % (tag, rel(dw, ref_w), rel(db, g_.sum((0, 2, 3)))))
    dw = torch.zeros_like(w0.data); db = torch.zeros_like(b0.data)
    ops.conv1_wgrad(x, dy1, dw, db)
    ref_w = torch.nn.grad.conv2d_weight(xd, w0.shape, r1, padding=1)
    print("  conv.0 dW %.3e  db %.3e" % (rel(dw, ref_w), rel(db, r1.sum((0, 2, 3)))))


if __name__ == "__main__":
    main()
