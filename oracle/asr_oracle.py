"""CPU oracle for the Transformer-ASR training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module.
The product path (end2end-asr-pytorch_amd/) never imports it and has no CPU fallback.

What it is: a self-contained fp32 restatement (plain torch-CPU tensor algebra, no nn.Module, no
reference imports) of the reference hot path, written from SURVEY.md Appendix B with every function
citing the reference file:line it follows.  It is PINNED: tests/test_oracle_golden.py checks it against
tests/golden/*.npz, which oracle/gen_golden.py produced by executing the unmodified reference
(/root/reference) on CPU in the build container.  Parity status: pinned (forward, loss, every
parameter gradient, two Noam/Adam steps, eval-mode forward) for vgg_cnn, emb_cnn and no-CNN configs.

Weights are addressed by the reference's state_dict key names.
"""
import math

import torch
import torch.nn.functional as F

PAD, SOS, EOS = 0, 1, 2          # utils/constant.py:102-104


class Cfg:
    """Hyper-parameters that the reference reads from its global argparse Namespace."""

    def __init__(self, num_layers, num_heads, dim_model, dim_key, dim_value, dim_inner, feat_extractor,
                 tgt_max_len, emb_trg_sharing=False, num_dec_layers=None):
        self.num_layers = num_layers
        self.num_dec_layers = num_layers if num_dec_layers is None else num_dec_layers
        self.num_heads = num_heads
        self.dim_model = dim_model
        self.dim_key = dim_key
        self.dim_value = dim_value
        self.dim_inner = dim_inner
        self.feat_extractor = feat_extractor
        self.tgt_max_len = tgt_max_len
        self.emb_trg_sharing = emb_trg_sharing

    @staticmethod
    def from_flags(flags):
        """Parse the subset of utils/constant.py:6-94 flags that shape the model."""
        f = flags.split() if isinstance(flags, str) else list(flags)
        d = dict(num_layers=3, num_heads=5, dim_model=512, dim_key=64, dim_value=64, dim_inner=1024,
                 feat_extractor="vgg_cnn", tgt_max_len=1000, emb_trg_sharing=False)
        names = {"--num-layers": ("num_layers", int), "--num-heads": ("num_heads", int),
                 "--dim-model": ("dim_model", int), "--dim-key": ("dim_key", int), "--dim-value": ("dim_value", int),
                 "--dim-inner": ("dim_inner", int), "--feat_extractor": ("feat_extractor", str),
                 "--tgt-max-len": ("tgt_max_len", int)}
        i = 0
        while i < len(f):
            if f[i] == "--emb_trg_sharing":
                d["emb_trg_sharing"] = True
                i += 1
            elif f[i] in names:
                k, ty = names[f[i]]
                # `--feat_extractor ""` loses its empty value when flags were joined with spaces
                if i + 1 >= len(f) or f[i + 1].startswith("--"):
                    d[k] = ty()
                    i += 1
                else:
                    d[k] = ty(f[i + 1])
                    i += 2
            else:
                i += 1
        return Cfg(**d)


# ------------------------------------------------------------------------------------------------ front end
def _relu(x, keep=None):
    """ReLU; with `keep` (bool, same shape) the SELECTION is imposed instead of derived: x * keep.  Used by parity tests to
    evaluate the fp64 gradient under the discrete decisions (ReLU masks, pooling arg-maxes) that the implementation under
    test took in its own forward pass -- at an element within rounding distance of 0 (or of a tie) either decision is a
    correct fp32 result, but the gradients of the two branches differ by that element's whole contribution."""
    return F.relu(x) if keep is None else x * keep.to(x.dtype)


def _max_pool(x, idx=None):
    """MaxPool2d(2, stride=2); with `idx` (flat indices into H*W per (b,c), as F.max_pool2d(return_indices=True)) the
    selection is imposed: out = x.flatten(2).gather(idx)."""
    if idx is None:
        return F.max_pool2d(x, 2, stride=2)
    B, C, H, W = x.shape
    return x.flatten(2).gather(2, idx.flatten(2)).view(B, C, H // 2, W // 2)


def conv_front_end(w, x, feat_extractor, training, bn_state=None, decisions=None):
    """models/asr/transformer.py:32-53 (module definition) and :70-76 (application + reshape).

    x: (B,1,F,T) -> (B,T',C*F') with feature index c*F'+f.
    bn_state: dict updated in place with BatchNorm running stats when training (emb_cnn only).
    decisions: optional dict {"relu0","relu2","pool4","relu5","relu7","pool9"} (vgg_cnn; see _relu / _max_pool).
    """
    dz = decisions or {}
    if feat_extractor == "vgg_cnn":                                   # :41-53
        x = _relu(F.conv2d(x, w["conv.0.weight"], w["conv.0.bias"], padding=1), dz.get("relu0"))
        x = _relu(F.conv2d(x, w["conv.2.weight"], w["conv.2.bias"], padding=1), dz.get("relu2"))
        x = _max_pool(x, dz.get("pool4"))
        x = _relu(F.conv2d(x, w["conv.5.weight"], w["conv.5.bias"], padding=1), dz.get("relu5"))
        x = _relu(F.conv2d(x, w["conv.7.weight"], w["conv.7.bias"], padding=1), dz.get("relu7"))
        x = _max_pool(x, dz.get("pool9"))
    elif feat_extractor == "emb_cnn":                                 # :32-40
        st = bn_state if bn_state is not None else {}
        x = F.conv2d(x, w["conv.0.weight"], w["conv.0.bias"], stride=(2, 2), padding=(0, 10))
        x = _batch_norm(x, w, "conv.1", training, st)
        x = F.hardtanh(x, 0.0, 20.0)
        x = F.conv2d(x, w["conv.3.weight"], w["conv.3.bias"], stride=(2, 1))
        x = _batch_norm(x, w, "conv.4", training, st)
        x = F.hardtanh(x, 0.0, 20.0)
    B, C, Fp, Tp = x.shape if x.dim() == 4 else (x.shape[0], 1, x.shape[1], x.shape[2])
    x = x.reshape(B, C * Fp, Tp).transpose(1, 2).contiguous()         # :74-76
    return x


def _batch_norm(x, w, prefix, training, st):
    """nn.BatchNorm2d(32) defaults: eps 1e-5, momentum 0.1, unbiased running var (transformer.py:35,38)."""
    rm = st.setdefault(prefix + ".running_mean", w[prefix + ".running_mean"].clone())
    rv = st.setdefault(prefix + ".running_var", w[prefix + ".running_var"].clone())
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() / x.shape[1]
        with torch.no_grad():
            rm.mul_(0.9).add_(0.1 * mean)
            rv.mul_(0.9).add_(0.1 * var * n / (n - 1))
            key = prefix + ".num_batches_tracked"
            st[key] = st.get(key, w[key].clone()) + 1
    else:
        mean, var = rm, rv
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
    return xh * w[prefix + ".weight"][None, :, None, None] + w[prefix + ".bias"][None, :, None, None]


# ------------------------------------------------------------------------------------------------ layers
def layer_norm(x, g, b):
    """nn.LayerNorm(D): biased variance, eps 1e-5 (common_layers.py:110,133,163; transformer.py:149)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-5) * g + b


def proj(w, name, x):
    """x W^T + b for the projection `name`; the Low-Rank Transformer variant (arXiv:1910.13923 -- cited by the reference
    README, no code in the reference tree: PARITY UNPINNED for this branch) stores it as a linear encoder-decoder unit
    <name>.u.weight (r, in) / <name>.v.weight (out, r) / <name>.v.bias: y = V (U x) + b."""
    if name + ".u.weight" in w:
        return (x @ w[name + ".u.weight"].t()) @ w[name + ".v.weight"].t() + w[name + ".v.bias"]
    W = w[name + ".weight"]
    if W.dim() == 3:                     # Conv1d(k=1) storage of the feed-forward weights
        W = W[:, :, 0]
    return x @ W.t() + w[name + ".bias"]


def multi_head_attention(w, p, q_in, kv_in, mask, H, dk, dv, return_attn=False):
    """models/common_layers.py:170-225.  mask: (B,Tq,Tk) bool, True = masked, or None."""
    B, Tq, _ = q_in.shape
    Tk = kv_in.shape[1]
    q = proj(w, p + "query_linear", q_in).view(B, Tq, H, dk)         # :181
    k = proj(w, p + "key_linear", kv_in).view(B, Tk, H, dk)           # :182
    v = proj(w, p + "value_linear", kv_in).view(B, Tk, H, dv)         # :183
    q = q.permute(2, 0, 1, 3)                                        # (H,B,Tq,dk)  :185
    k = k.permute(2, 0, 1, 3)
    v = v.permute(2, 0, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / (dk ** 0.5)           # :215-216, temperature = dk^0.5 (:162)
    if mask is not None:
        s = s.masked_fill(mask[None], float("-inf"))                 # :219 (mask.repeat(H,1,1) :190)
    a = torch.softmax(s, dim=-1)                                     # :221
    o = torch.matmul(a, v)                                           # :223  (H,B,Tq,dv)
    o = o.permute(1, 2, 0, 3).reshape(B, Tq, H * dv)                 # :194-195
    o = proj(w, p + "output_linear", o)                              # :197
    out = layer_norm(o + q_in, w[p + "layer_norm.weight"], w[p + "layer_norm.bias"])                 # :198
    if return_attn:
        return out, a.reshape(H * B, Tq, Tk)
    return out


def pos_ffn(w, p, x, decisions=None):
    """models/common_layers.py:135-142 (Conv1d k=1 == Linear on the last dim).  decisions[p + "relu"]: see _relu."""
    n1, n2 = ("linear_1", "linear_2") if (p + "linear_1.u.weight") in w else ("conv_1", "conv_2")
    h = _relu(proj(w, p + n1, x), (decisions or {}).get(p + "relu"))
    y = proj(w, p + n2, h)
    return layer_norm(y + x, w[p + "layer_norm.weight"], w[p + "layer_norm.bias"])


def length_masks(lengths, T):
    """common_layers.py:28-44 with input_lengths: keep[b,t] = t < len[b]  (RAW lengths vs this axis)."""
    t = torch.arange(T)[None, :]
    keep = (t < lengths.to(torch.int64)[:, None])
    return keep


def encoder_forward(w, cfg, x, src_len, decisions=None):
    """models/asr/transformer.py:157-180 + EncoderLayer :195-203."""
    B, Te, _ = x.shape
    keep = length_masks(src_len, Te)                                  # :168
    m_e = keep.to(x.dtype).unsqueeze(-1)
    attn_mask = (~keep)[:, None, :].expand(B, Te, Te)                 # :170  common_layers.py:57-64
    e = layer_norm(x @ w["encoder.input_linear.weight"].t() + w["encoder.input_linear.bias"],
                   w["encoder.layer_norm_input.weight"], w["encoder.layer_norm_input.bias"])
    e = e + w["encoder.positional_encoding.pe"][:, :Te]               # :172-173
    for l in range(cfg.num_layers):
        p = "encoder.layers.%d." % l
        e = multi_head_attention(w, p + "self_attn.", e, e, attn_mask, cfg.num_heads, cfg.dim_key, cfg.dim_value)
        e = e * m_e                                                   # :198
        e = pos_ffn(w, p + "pos_ffn.", e, decisions)
        e = e * m_e                                                   # :201
    return e


def decoder_preprocess(tgt, Td):
    """models/asr/transformer.py:254-266 + pad_list (common_layers.py:14-22): strip PAD anywhere, frame, pad to Td."""
    B = tgt.shape[0]
    seq_in = torch.full((B, Td), EOS, dtype=torch.int64)
    seq_out = torch.full((B, Td), PAD, dtype=torch.int64)
    for b in range(B):
        y = tgt[b][tgt[b] != PAD]
        n = y.numel()
        if n + 1 > Td:
            raise RuntimeError("target longer than --tgt-max-len (pad_list, common_layers.py:21)")
        seq_in[b, 0] = SOS
        seq_in[b, 1:n + 1] = y
        seq_out[b, :n] = y
        seq_out[b, n] = EOS
    return seq_in, seq_out


def decoder_forward(w, cfg, tgt, enc_out, src_len, decisions=None):
    """models/asr/transformer.py:268-305 + DecoderLayer :533-545."""
    B, Te, D = enc_out.shape
    Td = cfg.tgt_max_len
    seq_in, seq_out = decoder_preprocess(tgt, Td)
    m_d = (seq_in != EOS).to(enc_out.dtype).unsqueeze(-1)             # :282
    causal = torch.triu(torch.ones(Td, Td, dtype=torch.bool), diagonal=1)[None]           # :283
    keypad = (seq_in == EOS)[:, None, :].expand(B, Td, Td)            # :284-285
    self_mask = keypad | causal                                       # :286
    enc_keep = length_masks(src_len, Te)
    cross_mask = (~enc_keep)[:, None, :].expand(B, Td, Te)            # :289-290
    scale = (D ** -0.5) if cfg.emb_trg_sharing else 1.0               # :248-252
    emb_w = w["decoder.trg_embedding.weight"]
    d = emb_w[seq_in] * scale + w["decoder.positional_encoding.pe"][:, :Td]               # :292-293
    for l in range(cfg.num_dec_layers):
        p = "decoder.layers.%d." % l
        d = multi_head_attention(w, p + "self_attn.", d, d, self_mask, cfg.num_heads, cfg.dim_key, cfg.dim_value)
        d = d * m_d
        d = multi_head_attention(w, p + "encoder_attn.", d, enc_out, cross_mask, cfg.num_heads, cfg.dim_key,
                                 cfg.dim_value)
        d = d * m_d
        d = pos_ffn(w, p + "pos_ffn.", d, decisions)
        d = d * m_d
    out_w = emb_w if cfg.emb_trg_sharing else w["decoder.output_linear.weight"]
    logits = d @ out_w.t()                                            # :302 (no bias)
    return logits, seq_out


def transformer_forward(w, cfg, src, src_len, tgt, training=True, bn_state=None, decisions=None):
    """models/asr/transformer.py:59-85.  Returns (pred, gold, hyp_seq).  decisions: see _relu (parity tests only)."""
    if cfg.feat_extractor in ("vgg_cnn", "emb_cnn"):
        x = conv_front_end(w, src, cfg.feat_extractor, training, bn_state, (decisions or {}).get("conv"))
    else:
        B, C, Fq, T = src.shape
        x = src.reshape(B, C * Fq, T).transpose(1, 2).contiguous()
    enc = encoder_forward(w, cfg, x, src_len, decisions)
    pred, gold = decoder_forward(w, cfg, tgt, enc, src_len, decisions)
    hyp = pred.argmax(dim=2)      # torch.topk(pred,1) (:80) -- lowest index on exact ties, as argmax
    return pred, gold, hyp


# ------------------------------------------------------------------------------------------------ loss
def smoothed_ce(pred, gold, smoothing):
    """utils/metrics.py:102-132 and :86-94.  Returns (loss, num_correct, num_word)."""
    V = pred.shape[-1]
    logits = pred.reshape(-1, V)
    g = gold.reshape(-1)
    nonpad = g != PAD
    num_word = int(nonpad.sum())
    lp = torch.log_softmax(logits, dim=1)
    if smoothing > 0.0:
        onehot = torch.zeros_like(logits).scatter(1, (g * nonpad).view(-1, 1), 1.0)       # :121-122
        q = onehot * (1 - smoothing) + (1 - onehot) * smoothing / V                         # :123 (eps/V)
        row = -(q * lp).sum(dim=1)
        loss = row[nonpad].sum() / num_word                                                 # :127-130
    else:
        loss = F.nll_loss(lp, g, ignore_index=PAD, reduction="mean")                        # :132
    ncorrect = int((logits.argmax(1) == g)[nonpad].sum())                                   # :89-94
    return loss, ncorrect, num_word


# ------------------------------------------------------------------------------------------------ optimiser
def ctc_loss(pred, gold, input_lengths, target_lengths):
    """utils/metrics.py:133-154: log-softmax over the vocabulary, then torch's ctc_loss with blank = 0, reduction="mean"
    (per-utterance negative log-likelihood / target length, averaged over the batch), zero_infinity=False."""
    log_probs = F.log_softmax(pred.transpose(0, 1), dim=2)            # T x B x C  (:134, :153)
    return F.ctc_loss(log_probs, gold, input_lengths, target_lengths, reduction="mean")        # :154


def noam_rate(step, model_size, factor, warmup, min_lr):
    """utils/optimizer.py:27-32."""
    return max(min_lr, factor * (model_size ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5))))


class NoamAdam:
    """utils/optimizer.py:15-22 over torch.optim.Adam(betas=(0.9,0.98), eps=1e-9) (utils/functions.py:107)."""

    def __init__(self, params, model_size, factor=1.0, warmup=4000, min_lr=1e-5):
        self.params = params                      # dict name -> tensor (updated in place)
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0
        self.model_size, self.factor, self.warmup, self.min_lr = model_size, factor, warmup, min_lr
        self.rate = 0.0

    def step(self, grads):
        self.t += 1
        lr = self.rate = noam_rate(self.t, self.model_size, self.factor, self.warmup, self.min_lr)
        b1, b2, eps = 0.9, 0.98, 1e-9
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        with torch.no_grad():
            for k, p in self.params.items():
                g = grads[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = self.v[k].sqrt() / math.sqrt(bc2) + eps
                p.addcdiv_(self.m[k], denom, value=-lr / bc1)


# ------------------------------------------------------------------------------------------------ whole step
PARAM_SKIP = ("running_mean", "running_var", "num_batches_tracked", "positional_encoding.pe")


def trainable_names(w, cfg):
    names = [k for k in w if not k.endswith(PARAM_SKIP)]
    if cfg.emb_trg_sharing:
        names = [k for k in names if k != "decoder.output_linear.weight"]
    return names


def train_step(w, cfg, src, src_len, tgt, smoothing, opt=None, bn_state=None, decisions=None):
    """trainer/asr/trainer.py:56-111 minus the string/CER bookkeeping.  w: dict of fp32 tensors.
    Returns dict(loss, pred, gold, hyp, num_correct, grads)."""
    names = trainable_names(w, cfg)
    leaves = {k: w[k].detach().clone().requires_grad_(True) for k in names}
    wl = dict(w)
    wl.update(leaves)
    pred, gold, hyp = transformer_forward(wl, cfg, src, src_len, tgt, True, bn_state, decisions)
    loss, ncorrect, num_word = smoothed_ce(pred, gold, smoothing)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(w[k])) for k, g in zip(names, grads)}
    if opt is not None:
        opt.step(grads)
    return dict(loss=float(loss.detach()), pred=pred.detach(), gold=gold, hyp=hyp, num_correct=ncorrect,
                num_word=num_word, grads=grads)


def edit_distance(a, b):
    """Levenshtein distance (python-Levenshtein `distance`, used by utils/metrics.py:56,76)."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


# ------------------------------------------------------------------------------------------------ decoding
def decode_logits(w, cfg, ys, enc_out):
    """One pass of the decode-time decoder over the prefix ys (B,t): models/asr/transformer.py:336-350 (greedy) and
    :437-455 (beam) -- non_pad_mask of ones, causal self-attention mask only, dec_enc_attn_mask=None, dropout in eval
    mode.  Returns logits (B,t,V)."""
    B, t = ys.shape
    D = enc_out.shape[2]
    causal = torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)[None].expand(B, t, t)        # common_layers.py:66-74
    scale = (D ** -0.5) if cfg.emb_trg_sharing else 1.0
    emb_w = w["decoder.trg_embedding.weight"]
    d = emb_w[ys] * scale + w["decoder.positional_encoding.pe"][:, :t]
    for l in range(cfg.num_dec_layers):
        p = "decoder.layers.%d." % l
        d = multi_head_attention(w, p + "self_attn.", d, d, causal, cfg.num_heads, cfg.dim_key, cfg.dim_value)
        d = multi_head_attention(w, p + "encoder_attn.", d, enc_out, None, cfg.num_heads, cfg.dim_key, cfg.dim_value)
        d = pos_ffn(w, p + "pos_ffn.", d)
    out_w = emb_w if cfg.emb_trg_sharing else w["decoder.output_linear.weight"]
    return d @ out_w.t()


def greedy_search(w, cfg, enc_out, id2label, steps=300):
    """models/asr/transformer.py:316-394 without LM rescoring: 300 arg-max steps for the whole batch, then every row is
    cut at its first EOS (:385-393).  Returns the list of strings."""
    B = enc_out.shape[0]
    ys = torch.full((B, 1), SOS, dtype=torch.int64)
    toks = []
    with torch.no_grad():
        for _ in range(steps):
            nxt = decode_logits(w, cfg, ys, enc_out)[:, -1].argmax(dim=1)          # torch.max(prob[:, -1], dim=1) :373
            toks.append(nxt)
            ys = torch.cat([ys, nxt[:, None]], dim=1)
    sents = []
    for row in torch.stack(toks, dim=1).tolist():
        st = ""
        for t in row:
            if t == EOS:
                break
            st += id2label[t]
        sents.append(st)
    return sents


def beam_search(w, cfg, enc_out, id2label, beam_width, nbest=1, c_weight=1.0, pad_char="¶", sos_char="§",
                eos_char="¤", steps=300):
    """models/asr/transformer.py:396-517 without the LM branch.  Per utterance: every live hypothesis is extended by its
    beam_width best tokens, the candidate list is re-sorted and cut INSIDE the hypothesis loop (:460), hypotheses ending in
    EOS are retired with final_score = score + sqrt(#words) * c_weight (:487-490), EOS is forced at step T_enc-1 (:465-467).
    Returns (ids, strings) of the nbest retired hypotheses per utterance; strings keep the EOS char (post_process_hyp)."""
    ids_out, strs_out = [], []
    max_len = enc_out.shape[1]
    with torch.no_grad():
        for b in range(enc_out.shape[0]):
            enc = enc_out[b:b + 1]
            hyps = [{"score": 0.0, "yseq": [SOS]}]
            ended = []
            for i in range(steps):
                kept = []
                for hyp in hyps:
                    ys = torch.tensor([hyp["yseq"]], dtype=torch.int64)
                    lp = torch.log_softmax(decode_logits(w, cfg, ys, enc)[:, -1], dim=1)
                    best, idx = torch.topk(lp, beam_width, dim=1)
                    for j in range(beam_width):
                        kept.append({"score": hyp["score"] + float(best[0, j]), "yseq": hyp["yseq"] + [int(idx[0, j])]})
                    kept = sorted(kept, key=lambda h: h["score"], reverse=True)[:beam_width]
                hyps = kept
                if i == max_len - 1:
                    for hyp in hyps:
                        hyp["yseq"] = hyp["yseq"] + [EOS]
                alive = []
                for hyp in hyps:
                    if hyp["yseq"][-1] == EOS:
                        s = "".join(id2label[t] for t in hyp["yseq"])
                        for ch in (pad_char, sos_char, eos_char):
                            s = s.replace(ch, "")
                        s = s.replace("  ", " ")
                        hyp["final_score"] = hyp["score"] + math.sqrt(len(s.split())) * c_weight
                        ended.append(hyp)
                    else:
                        alive.append(hyp)
                hyps = alive
                if not hyps:
                    break
            for hyp in sorted(ended, key=lambda h: h["final_score"], reverse=True)[:min(len(ended), nbest)]:
                ids_out.append(hyp["yseq"])
                strs_out.append("".join(id2label[t] for t in hyp["yseq"][1:]))
    return ids_out, strs_out


def eval_error_counts(strs_hyps, strs_gold, pad_char="¶", sos_char="§", eos_char="¤"):
    """test.py:42-58: (total_cer, total_char, total_wer, total_word) over a batch of hypothesis / gold strings."""
    tc = tch = tw = twd = 0
    for h, g in zip(strs_hyps, strs_gold):
        for ch in (eos_char, sos_char, pad_char):
            h, g = h.replace(ch, ""), g.replace(ch, "")
        tw += edit_distance(h.split(), g.split())                     # utils/metrics.py:58-76
        tc += edit_distance(h.strip(), g.strip())                     # utils/metrics.py:48-56
        twd += len(g.split(" "))
        tch += len(g)
    return tc, tch, tw, twd
