"""Shared helpers for the BASELINE-shape parity cases (tests/golden/{cfg0,cfg1_b2,cfg3_shape,cfg1_b32,cfg3_b16}.npz).

The fixtures were produced by EXECUTING the reference (oracle/gen_golden.py, BIG cases) and hold summaries only; the
weights are rebuilt here from the reference's seed: the product's constructors consume torch's CPU RNG exactly like
the reference's (same modules in the same order), and the 1-D parameters get the same seeded perturbation in sorted
name order.  `wsum` in the fixture (sum and sum of squares over all weights) confirms that the rebuilt weights are
the ones the reference ran with.
"""
import os
import zlib

import numpy as np
import torch

BIG_CASES = ["cfg0", "cfg1_b2", "cfg3_shape", "cfg1_b32", "cfg3_b16"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def labels(V):
    from utils import constant
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x4E00 + i) for i in range(V - 3)]
    l2i = {c: i for i, c in enumerate(chars)}
    return l2i, {i: c for c, i in l2i.items()}


def perturb_1d(model):
    g = torch.Generator().manual_seed(4321)
    named = dict(model.named_parameters())
    with torch.no_grad():
        for n in sorted(named):
            if named[n].dim() == 1:
                named[n].add_(0.1 * torch.randn(named[n].shape, generator=g))


def batch(z):
    B, T, V = int(z["B"]), int(z["T"]), int(z["V"])
    g = torch.Generator().manual_seed(1234)
    src = torch.randn(B, 1, 161, T, generator=g)
    src_len = torch.from_numpy(z["src_len"]).to(torch.int32)
    for b in range(B):
        src[b, :, :, int(src_len[b]):] = 0.0
    tgt_len = [int(x) for x in z["tgt_len"]]
    tgt = torch.zeros(B, max(tgt_len), dtype=torch.int64)
    for b, L in enumerate(tgt_len):
        tgt[b, :L] = torch.randint(3, V, (L,), generator=g)
    assert torch.equal(tgt, torch.from_numpy(z["tgt"])), "synthetic targets differ from the fixture's"
    assert abs(float(src.double().sum()) - float(z["src_sum"])) < 1e-6 * src.numel() ** 0.5
    return src, src_len, tgt


def build_product(z, precision, cuda):
    """The product model (CPU construction = the reference's init stream), its args and label maps."""
    from utils import constant
    from utils.functions import init_transformer_model
    flags = str(z["flags"]).split()
    args = constant.parse(flags + ["--precision", precision] + (["--cuda"] if cuda else []))
    l2i, i2l = labels(int(z["V"]))
    torch.manual_seed(123456)
    if int(z["enc_layers"]) > 0:
        from asr_hip import ops
        from models.asr.transformer import Decoder, Encoder, Transformer
        args.dim_input = 32 * 21
        ops.set_compute_dtype(torch.float32 if precision == "fp32" else torch.bfloat16)
        enc = Encoder(int(z["enc_layers"]), args.num_heads, args.dim_model, args.dim_key, args.dim_value, args.dim_input,
                      args.dim_inner, dropout=args.dropout, src_max_length=args.src_max_len)
        dec = Decoder(i2l, len(l2i), len(l2i), int(z["dec_layers"]), args.num_heads, args.dim_emb, args.dim_model,
                      args.dim_inner, args.dim_key, args.dim_value, dropout=args.dropout, trg_max_length=args.tgt_max_len,
                      emb_trg_sharing=False)
        model = Transformer(enc, dec, feat_extractor="emb_cnn")
    else:
        model = init_transformer_model(args, l2i, i2l)
    perturb_1d(model)
    s1 = s2 = 0.0
    sd = model.state_dict()
    for k in sorted(sd):
        if k.endswith("num_batches_tracked"):
            continue
        a = sd[k].detach().double().numpy()
        s1 += float(a.sum())
        s2 += float((a * a).sum())
    ref = z["wsum"]
    # bit-identical in the build container; another host CPU may take a different vectorised sampling path -> tolerance
    assert abs(s1 - ref[0]) <= 1e-3 + 1e-6 * abs(ref[0]) and abs(s2 - ref[1]) <= 1e-6 * abs(ref[1]), (s1, s2, ref)
    return args, model, l2i, i2l


def oracle_cfg(z):
    from oracle import asr_oracle as O
    cfg = O.Cfg.from_flags(str(z["flags"]))
    if int(z["enc_layers"]) > 0:
        cfg.num_layers, cfg.num_dec_layers = int(z["enc_layers"]), int(z["dec_layers"])
    return cfg


def _sign(name, n):
    seed = zlib.crc32(name.encode()) & 0x7FFFFFFF
    return torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(seed)).double() * 2 - 1


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def summary_errors(z, pred, loss, grads):
    """Compare (pred (B,Td,V), loss, {name: grad}) with the reference summaries.  Returns a dict of error figures:
    pred_sub / pred_lse max abs error, loss abs error, per-parameter relative norm error, projection error in units of the
    gradient norm, relative L2 error on the stored samples; plus the arg-max agreement over rows with margin > `m`."""
    p = pred.detach().float().cpu()
    out = {"pred_sub": float((p[:, :, torch.from_numpy(z["pred_idx"])] - torch.from_numpy(z["pred_sub"])).abs().max()),
           "pred_lse": float((torch.logsumexp(p, dim=2) - torch.from_numpy(z["pred_lse"])).abs().max()),
           "loss": abs(float(loss) - float(z["loss"])), "gn": {}, "gp": {}, "gs": {}}
    for k, g in grads.items():
        f = g.detach().reshape(-1).double().cpu()
        ref_n = float(z["gn/" + k])
        out["gn"][k] = abs(float(f.norm()) - ref_n) / (ref_n + 1e-30)
        # projection on a +-1 vector: the error is ~ ||dg|| (random signs), so measure it in units of ||g||
        out["gp"][k] = abs(float((f * _sign(k, f.numel())).sum()) - float(z["gp/" + k])) / (ref_n + 1e-30)
        if "g0/" + k in z.files:
            out["gs"][k] = rel_l2(f.numpy(), z["g0/" + k])
        else:
            stride = f.numel() // 1024
            out["gs"][k] = rel_l2(f[::stride][:1024].numpy(), z["gs/" + k])
    return out


def argmax_agreement(z, hyp, margin):
    h = np.asarray(hyp.cpu() if hasattr(hyp, "cpu") else hyp).astype(np.int64)
    sure = z["margin"] > margin
    return int((h[sure] != z["hyp"][sure]).sum()), int(sure.sum())


def noise_driven(k, emb):
    """Parameters whose exact gradient is zero: key biases (softmax shift invariance), conv biases feeding BatchNorm."""
    return k.endswith("key_linear.bias") or (emb and k in ("conv.0.bias", "conv.3.bias"))
