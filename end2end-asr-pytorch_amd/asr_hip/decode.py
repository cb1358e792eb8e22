"""KV-cached incremental decoding for models.asr.transformer.Decoder.

The reference (models/asr/transformer.py:316-517) re-runs the WHOLE decoder over the growing prefix for every one of its
300 greedy steps (and once per live hypothesis per step in beam search).  Decoding is causal, so position t of that
re-run depends only on positions <= t; this module computes exactly that row: per layer the self-attention keys/values
of the prefix are cached, the cross-attention keys/values of the encoder output are projected once, and a step pushes
ONE token per sequence through the same HIP kernels (asr_gemm_nt / asr_attn_fwd / asr_add_ln_fwd) in the same operand
order -- row t of the full recomputation and the cached step are the same arithmetic.

Decode-time masks are the reference's: causal only for self attention, no encoder-length mask for cross attention
(dec_enc_attn_mask=None, transformer.py:336-350); dropout off.
"""
import torch

from . import functions as F_
from . import ops
from utils import constant


class DecoderKVCache:
    def __init__(self, decoder, encoder_padded_outputs, max_len, batch=None):
        """encoder_padded_outputs (Be,Te,D).  batch: number of decoder rows (hypotheses); with Be == 1 the cross-attention
        keys/values are shared by every row (stride-0 batch axis)."""
        self.dec = decoder
        cd = ops.compute_dtype()
        enc = encoder_padded_outputs
        if enc.dtype != cd:
            enc = enc.to(cd)
        enc = enc.contiguous()
        Be, Te, D = enc.shape
        self.B = Be if batch is None else batch
        assert Be in (1, self.B)
        self.H, self.dk = decoder.num_heads, decoder.dim_key
        HD = self.H * self.dk
        self.max_len = max_len
        self.t = 0
        dev = enc.device
        self.cross, self.self_k, self.self_v = [], [], []
        enc2 = enc.view(Be * Te, D)
        for layer in decoder.layers:
            a = layer.encoder_attn
            k = F_._linear_fwd(enc2, a.key_linear.weight, a.key_linear.bias).view(Be, Te, HD)
            v = F_._linear_fwd(enc2, a.value_linear.weight, a.value_linear.bias).view(Be, Te, HD)
            if Be != self.B:
                k, v = k.expand(self.B, Te, HD), v.expand(self.B, Te, HD)
            self.cross.append((k, v))
            self.self_k.append(torch.empty((self.B, max_len, HD), device=dev, dtype=cd))
            self.self_v.append(torch.empty((self.B, max_len, HD), device=dev, dtype=cd))

    # ------------------------------------------------------------------------------------------------ hypotheses
    def select(self, rows, cross=True):
        """Keep / duplicate / reorder decoder rows (beam search: row i of the new state continues old row rows[i]).
        cross=False: the cross-attention keys / values stay where they are (the caller guarantees that rows[i] and i belong to
        the same utterance, and len(rows) is the current batch)."""
        idx = torch.as_tensor(rows, device=self.self_k[0].device, dtype=torch.int64)
        t = self.t
        for i in range(len(self.self_k)):
            nk = torch.empty((len(rows),) + tuple(self.self_k[i].shape[1:]), device=idx.device, dtype=self.self_k[i].dtype)
            nv = torch.empty_like(nk)
            nk[:, :t] = self.self_k[i][:, :t].index_select(0, idx)
            nv[:, :t] = self.self_v[i][:, :t].index_select(0, idx)
            self.self_k[i], self.self_v[i] = nk, nv
            if not cross:
                assert len(rows) == self.B
                continue
            k, v = self.cross[i]
            if k.stride(0) == 0:
                self.cross[i] = (k[:1].expand(len(rows), -1, -1), v[:1].expand(len(rows), -1, -1))
            else:
                self.cross[i] = (k.index_select(0, idx), v.index_select(0, idx))
        self.B = len(rows)

    # ------------------------------------------------------------------------------------------------ one token
    def _attend(self, attn_mod, x, q_in, k, v):
        """MultiHeadAttention on one query row per sequence: out = LN(W_o . attn(q, K, V) + b_o + x)."""
        B = x.shape[0]
        HD = self.H * self.dk
        o, _, _ = ops.attn_fwd(q_in.view(B, 1, HD), k, v, self.H, self.dk, scale=1.0 / (self.dk ** 0.5))
        y = F_._linear_fwd(o.view(B, HD), attn_mod.output_linear.weight, attn_mod.output_linear.bias)
        out, _, _ = ops.add_ln_fwd(y, x, attn_mod.layer_norm.weight.data, attn_mod.layer_norm.bias.data)
        return out

    @torch.no_grad()
    def step(self, tokens):
        """tokens (B,) int64: the input token at position self.t -> logits (B,V) fp32 for position self.t."""
        dec = self.dec
        t = self.t
        if t >= self.max_len:
            raise RuntimeError("decode position %d exceeds the cache length %d" % (t, self.max_len))
        B = self.B
        HD = self.H * self.dk
        pe = dec.positional_encoding.pe[0]
        x = ops.embed_fwd(tokens.view(B, 1).contiguous(), dec.trg_embedding.weight.data, pe[t:t + 1].contiguous(),
                          dec.x_logit_scale, 0.0, 0, ops.compute_dtype()).view(B, -1)
        for i, layer in enumerate(dec.layers):
            sa = layer.self_attn
            q = F_._linear_fwd(x, sa.query_linear.weight, sa.query_linear.bias)
            # this position's key / value rows go straight into the cache (strided GEMM output)
            kt, vt = self.self_k[i][:, t], self.self_v[i][:, t]
            W, Wv = F_.P.linear_weight(sa.key_linear.weight), F_.P.linear_weight(sa.value_linear.weight)
            xp = F_._pad_cols(x) if W.shape[1] != x.shape[1] else x
            ops.gemm_nt(xp, W, bias=sa.key_linear.bias.data, out=kt)
            ops.gemm_nt(xp, Wv, bias=sa.value_linear.bias.data, out=vt)
            x = self._attend(sa, x, q, self.self_k[i][:, :t + 1], self.self_v[i][:, :t + 1])
            ca = layer.encoder_attn
            q = F_._linear_fwd(x, ca.query_linear.weight, ca.query_linear.bias)
            x = self._attend(ca, x, q, self.cross[i][0], self.cross[i][1])
            ff = layer.pos_ffn
            w1, w2 = (ff.conv_1, ff.conv_2) if hasattr(ff, "conv_1") else (ff.linear_1, ff.linear_2)
            h = F_._linear_fwd(x, w1.weight, w1.bias, relu=True)
            y = F_._linear_fwd(h, w2.weight, w2.bias)
            x, _, _ = ops.add_ln_fwd(y, x, ff.layer_norm.weight.data, ff.layer_norm.bias.data)
        self.t = t + 1
        return F_._linear_fwd(x, dec.output_linear.weight, None, out_dtype=torch.float32)


class GraphedGreedyDecoder:
    """The greedy loop as ONE captured hipGraph replayed per token.  Everything that changes between steps lives on the
    device: the position (state[0]), the token buffer (argmax written in place), the caches.  Fixed shapes: the
    self-attention always addresses the whole (B, max_len, H*d) cache and masks positions > t with a device-side key_len
    (asr_decode_prepare), this position's key / value rows are appended by asr_kv_append at the device-side row.  A step is
    ~15 launches per layer; replayed they cost no host time (the eager cached loop is bound by ~60 Python-issued launches per
    token)."""

    def __init__(self, decoder, encoder_padded_outputs, max_len):
        self.dec = decoder
        self.cache = DecoderKVCache(decoder, encoder_padded_outputs, max_len)
        c = self.cache
        dev = encoder_padded_outputs.device
        self.B, self.max_len = c.B, max_len
        self.state = torch.zeros(2, dtype=torch.int64, device=dev)
        self.key_len = torch.zeros(c.B, dtype=torch.int32, device=dev)
        pe = decoder.positional_encoding.pe[0]
        self.pe = pe[:max_len].float().contiguous()
        self.pe_cur = torch.zeros((1, pe.shape[1]), dtype=torch.float32, device=dev)
        self.tok = torch.full((c.B,), constant.SOS_TOKEN, dtype=torch.int64, device=dev)
        self.done = torch.zeros(c.B, dtype=torch.bool, device=dev)
        self.out = torch.zeros((max_len, c.B), dtype=torch.int64, device=dev)
        for i in range(len(c.self_k)):               # positions > t are masked, but must hold finite numbers
            c.self_k[i].zero_()
            c.self_v[i].zero_()
        self.graph = None

    def _attend(self, attn_mod, x, q_in, k, v, key_len):
        c = self.cache
        B = x.shape[0]
        HD = c.H * c.dk
        o, _, _ = ops.attn_fwd(q_in.view(B, 1, HD), k, v, c.H, c.dk, key_len=key_len, scale=1.0 / (c.dk ** 0.5))
        y = F_._linear_fwd(o.view(B, HD), attn_mod.output_linear.weight, attn_mod.output_linear.bias)
        out, _, _ = ops.add_ln_fwd(y, x, attn_mod.layer_norm.weight.data, attn_mod.layer_norm.bias.data)
        return out

    def _step(self):
        dec, c = self.dec, self.cache
        B = self.B
        ops.decode_prepare(self.pe, self.pe_cur, self.key_len, self.state)
        x = ops.embed_fwd(self.tok.view(B, 1), dec.trg_embedding.weight.data, self.pe_cur, dec.x_logit_scale, 0.0, 0,
                          ops.compute_dtype()).view(B, -1)
        for i, layer in enumerate(dec.layers):
            sa = layer.self_attn
            q = F_._linear_fwd(x, sa.query_linear.weight, sa.query_linear.bias)
            kt = F_._linear_fwd(x, sa.key_linear.weight, sa.key_linear.bias)
            vt = F_._linear_fwd(x, sa.value_linear.weight, sa.value_linear.bias)
            ops.kv_append(kt, vt, c.self_k[i], c.self_v[i], self.state)
            x = self._attend(sa, x, q, c.self_k[i], c.self_v[i], self.key_len)
            ca = layer.encoder_attn
            q = F_._linear_fwd(x, ca.query_linear.weight, ca.query_linear.bias)
            x = self._attend(ca, x, q, c.cross[i][0], c.cross[i][1], None)
            ff = layer.pos_ffn
            w1, w2 = (ff.conv_1, ff.conv_2) if hasattr(ff, "conv_1") else (ff.linear_1, ff.linear_2)
            h = F_._linear_fwd(x, w1.weight, w1.bias, relu=True)
            y = F_._linear_fwd(h, w2.weight, w2.bias)
            x, _, _ = ops.add_ln_fwd(y, x, ff.layer_norm.weight.data, ff.layer_norm.bias.data)
        logits = F_._linear_fwd(x, dec.output_linear.weight, None, out_dtype=torch.float32)
        ops.argmax_rows(logits, out=self.tok)
        self.done |= self.tok.eq(constant.EOS_TOKEN)
        ops.decode_advance(self.state)

    @torch.no_grad()
    def run(self, steps, check_every=32):
        """-> token ids (B, n <= steps).  Two eager steps (warm-up: shadows, workspaces), capture, then replays."""
        n = 0
        for _ in range(min(2, steps)):
            self._step()
            self.out[n].copy_(self.tok)
            n += 1
        if n < steps:
            torch.cuda.synchronize()
            if self.graph is None:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self._step()
            while n < steps:
                self.graph.replay()
                self.out[n].copy_(self.tok)
                n += 1
                if n % check_every == 0 and bool(self.done.all()):
                    break
        return self.out[:n].t().contiguous()


def _weights_key(decoder):
    ws = list(decoder.parameters())
    return (F_.P._state["generation"], tuple(p._version for p in ws), tuple(p.data_ptr() for p in ws))


def _dec_pack(decoder):
    """bf16 weights of the decode-step GEMMs, Q/K/V of the self attention concatenated; cached on the decoder until a weight
    changes (optimiser step / load_state_dict)."""
    key = _weights_key(decoder)
    hit = getattr(decoder, "_asr_dec_pack", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    bf = torch.bfloat16

    def w2(lin):
        w = lin.weight.data
        return w.reshape(w.shape[0], -1).to(bf).contiguous()

    def fr(w):                                  # (flat fragment-major weight, (N, K)): 1 KB per wave-wide load (include/asr_hip.h)
        return ops.frag_pack(w), tuple(w.shape)

    def b1(lin):
        return lin.bias.data.float().contiguous() if lin.bias is not None else None

    layers = []
    for layer in decoder.layers:
        sa, ca, ff = layer.self_attn, layer.encoder_attn, layer.pos_ffn
        w1, w2_ = (ff.conv_1, ff.conv_2) if hasattr(ff, "conv_1") else (ff.linear_1, ff.linear_2)
        layers.append(dict(
            wqkv=fr(torch.cat([w2(sa.query_linear), w2(sa.key_linear), w2(sa.value_linear)], 0)),
            wq_c_rm=w2(ca.query_linear),
            bqkv=torch.cat([b1(sa.query_linear), b1(sa.key_linear), b1(sa.value_linear)], 0).contiguous(),
            wo_s=fr(w2(sa.output_linear)), bo_s=b1(sa.output_linear),
            ln_s=(sa.layer_norm.weight.data.float().contiguous(), sa.layer_norm.bias.data.float().contiguous(), sa.layer_norm.eps),
            wq_c=fr(w2(ca.query_linear)), bq_c=b1(ca.query_linear), wo_c=fr(w2(ca.output_linear)), bo_c=b1(ca.output_linear),
            ln_c=(ca.layer_norm.weight.data.float().contiguous(), ca.layer_norm.bias.data.float().contiguous(), ca.layer_norm.eps),
            w1=fr(w2(w1)), b1=b1(w1), w2=fr(w2(w2_)), b2=b1(w2_),
            ln_f=(ff.layer_norm.weight.data.float().contiguous(), ff.layer_norm.bias.data.float().contiguous(), ff.layer_norm.eps)))
    pack = dict(layers=layers, wout=fr(w2(decoder.output_linear)), table=decoder.trg_embedding.weight.data.float().contiguous())
    decoder._asr_dec_pack = (key, pack)
    return pack


def fused_decode_supported(decoder, encoder_padded_outputs, max_len):
    """The 34-launch step (csrc/decode.hip): bf16, d_model <= 512 and a multiple of 64, dk = dv = 64, inner dimension a multiple
    of 64, plain (full-rank) projections.  Any number of sequences (greedy_search_graphed decodes them 32 at a time: the step's GEMMs
    are one 32-row MFMA tile) and of encoder frames / positions (asr_dec_attn walks the keys in passes of 512; the one-launch
    projection + attention form, asr_dec_attn_fused, is used up to 512 keys)."""
    if ops.compute_dtype() != torch.bfloat16 or not hasattr(decoder.layers[0].self_attn, "query_linear"):
        return False
    sa = decoder.layers[0].self_attn
    if not isinstance(getattr(sa.query_linear, "weight", None), torch.Tensor):
        return False
    D = decoder.dim_model
    B, Te, _ = encoder_padded_outputs.shape
    ff = decoder.layers[0].pos_ffn
    w1 = ff.conv_1 if hasattr(ff, "conv_1") else ff.linear_1
    return (B >= 1 and D % 64 == 0 and D <= 512 and decoder.dim_key == 64 and getattr(decoder, "dim_value", 64) == 64 and
            w1.weight.shape[0] % 64 == 0 and (decoder.num_heads * decoder.dim_key) % 64 == 0)


class FusedGreedyDecoder:
    """Greedy loop on the decode-step kernels of csrc/decode.hip: per layer 8 launches (Q/K/V GEMM with the previous LayerNorm or
    the embedding as its prologue, self attention that appends its own key / value rows, output GEMM, cross-attention query GEMM
    with LayerNorm prologue, cross attention, output GEMM, two feed-forward GEMMs), then the vocabulary GEMM and asr_dec_finish:
    34 launches per token for the 4-layer model instead of 62 (30 with the cross-attention query projection inside the attention
    launch, asr_dec_attn_fused), captured once (TOKENS_PER_GRAPH steps per graph: the host's
    replay gap is paid once per graph) and replayed; the object -- buffers and graph -- is kept on the decoder and reused by
    later calls with the same shapes (greedy_search_graphed), only the cross-attention keys / values are recomputed."""

    TOKENS_PER_GRAPH = 8
    # projections inside the attention launches (asr_dec_attn_fused).  Measured at B = 32, t = 300 (profiles/r02_decode_trace.txt):
    # cross attention 11.0 us fused vs 6.7 + 8.6 us as two launches -- on.  (The self attention fused the same way measured 19.9 us
    # against 7.0 + 10.8: every (sequence, head) workgroup re-reads the head's 192 KB of Q/K/V rows, 48 MB through L2 per layer against
    # 1.5 MB in the GEMM; that variant is not kept.)
    FUSE_CROSS = True

    def __init__(self, decoder, encoder_padded_outputs, max_len):
        self.dec = decoder
        self.key = _weights_key(decoder)
        self.cache = DecoderKVCache(decoder, encoder_padded_outputs, max_len)
        c = self.cache
        dev = encoder_padded_outputs.device
        bf = torch.bfloat16
        self.B, self.max_len = c.B, max_len
        B, D, HD = c.B, decoder.dim_model, c.H * c.dk
        if B > 32:
            raise ValueError("the fused step holds at most 32 sequences (greedy_search_graphed splits larger batches)")
        # the one-launch projection + attention kernels keep every key of a (sequence, head) in registers: up to 512 of them
        self.fuse_cross = self.FUSE_CROSS and encoder_padded_outputs.shape[1] <= 512
        self.pack = _dec_pack(decoder)
        self.state = torch.zeros(2, dtype=torch.int64, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pe = decoder.positional_encoding.pe[0][:max_len].float().contiguous()
        self.tok = torch.full((B,), constant.SOS_TOKEN, dtype=torch.int64, device=dev)
        self.done = torch.zeros(B, dtype=torch.bool, device=dev)
        self.out = torch.zeros((max_len, B), dtype=torch.int64, device=dev)
        dff = self.pack["layers"][0]["w1"][1][0]
        V = self.pack["wout"][1][0]
        z = lambda *shape: torch.zeros(shape, dtype=bf, device=dev)
        self.xs = [z(B, D), z(B, D), z(B, D)]          # sub-layer inputs (residuals): x0 -> self attention, x1 -> cross, x2 -> ffn
        self.qkv, self.y, self.qc = z(B, 3 * HD), z(B, D), z(B, HD)
        self.o, self.h = z(32 * HD), z(32 * dff)             # fragment-major: what the next GEMM's wave-wide loads want
        self.y2, self.y3 = z(B, D), z(B, D)
        self.logits = torch.zeros((B, (V + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :V]
        self.graph = None

    def _step(self):
        dec, c, P_ = self.dec, self.cache, self.pack
        HD = c.H * c.dk
        scale = 1.0 / (c.dk ** 0.5)
        x0, x1, x2 = self.xs
        prev = None                                                   # (Y, R, gamma, beta, eps) of the pending LayerNorm
        B = self.B

        def gemm(w, bias, out, **kw):
            return ops.dec_gemm(w[0], bias, out, w_frag=w[1], **kw)

        for i, Lw in enumerate(P_["layers"]):
            if prev is None:
                gemm(Lw["wqkv"], Lw["bqkv"], self.qkv, x_out=x0, embed=(self.tok, P_["table"], self.pe, dec.x_logit_scale, self.state))
            else:
                gemm(Lw["wqkv"], Lw["bqkv"], self.qkv, ln=prev, x_out=x0)
            ops.dec_attn(self.qkv[:, :HD], c.self_k[i], c.self_v[i], self.o, c.H, c.dk, scale,
                         k_new=self.qkv[:, HD:2 * HD], v_new=self.qkv[:, 2 * HD:], state=self.state, out_frag=True)
            gemm(Lw["wo_s"], Lw["bo_s"], self.y, x=self.o, x_frag=True)
            if self.fuse_cross:
                ops.dec_attn_fused(Lw["wq_c_rm"], Lw["bq_c"], c.cross[i][0], c.cross[i][1], self.o, c.H, c.dk, scale,
                                   ln=(self.y, x0) + Lw["ln_s"], x_out=x1, out_frag=True)
            else:
                gemm(Lw["wq_c"], Lw["bq_c"], self.qc, ln=(self.y, x0) + Lw["ln_s"], x_out=x1)
                ops.dec_attn(self.qc, c.cross[i][0], c.cross[i][1], self.o, c.H, c.dk, scale, out_frag=True)
            gemm(Lw["wo_c"], Lw["bo_c"], self.y2, x=self.o, x_frag=True)
            gemm(Lw["w1"], Lw["b1"], self.h, ln=(self.y2, x1) + Lw["ln_c"], x_out=x2, relu=True, out_frag=True, B=B)
            gemm(Lw["w2"], Lw["b2"], self.y3, x=self.h, x_frag=True)
            prev = (self.y3, x2) + Lw["ln_f"]
        gemm(P_["wout"], None, self.logits, ln=prev)
        ops.dec_finish(self.logits, self.tok, self.done, self.out, constant.EOS_TOKEN, self.state, self.ticket)

    @torch.no_grad()
    def reset(self, encoder_padded_outputs):
        """Start a new batch of the same shape: position 0, SOS tokens, fresh cross-attention keys / values (in place: the
        captured graph keeps its addresses)."""
        c = self.cache
        cd = ops.compute_dtype()
        enc = encoder_padded_outputs.to(cd).contiguous()
        Be, Te, D = enc.shape
        assert (Be, Te) == tuple(c.cross[0][0].shape[:2])
        enc2 = enc.view(Be * Te, D)
        for i, layer in enumerate(self.dec.layers):
            a = layer.encoder_attn
            c.cross[i][0].copy_(F_._linear_fwd(enc2, a.key_linear.weight, a.key_linear.bias).view(Be, Te, -1))
            c.cross[i][1].copy_(F_._linear_fwd(enc2, a.value_linear.weight, a.value_linear.bias).view(Be, Te, -1))
        self.state.zero_()
        self.ticket.zero_()
        self.tok.fill_(constant.SOS_TOKEN)
        self.done.zero_()
        self.out.zero_()

    @torch.no_grad()
    def step_logits(self, tokens):
        """Teacher-forced step (tests): feed `tokens` (B,) at the current position -> fp32 logits (B, V) (a view, overwritten by
        the next step)."""
        self.tok.copy_(tokens)
        self._step()
        return self.logits

    @torch.no_grad()
    def run(self, steps, check_every=32):
        """-> token ids (B, n <= steps).  First use: two eager steps (warm-up) and the capture; then one replay per
        TOKENS_PER_GRAPH tokens, the remainder eagerly."""
        assert steps <= self.max_len
        n, G = 0, self.TOKENS_PER_GRAPH
        if self.graph is None:
            for _ in range(min(2, steps)):
                self._step()
                n += 1
            if steps - n >= G:
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    for _ in range(G):
                        self._step()
        nxt = check_every
        while n < steps:
            if self.graph is not None and steps - n >= G:
                self.graph.replay()
                n += G
            else:
                self._step()
                n += 1
            if n >= nxt:
                nxt += check_every
                if bool(self.done.all()):
                    break
        return self.out[:n].t().contiguous()


@torch.no_grad()
def greedy_search_graphed(decoder, encoder_padded_outputs, steps=300, fused=None):
    """Same tokens as greedy_search below, one hipGraph replay per token.  fused: None = the 34-launch step when the shapes
    allow it (bf16), False = the kernel-per-op step."""
    if fused is None:
        fused = fused_decode_supported(decoder, encoder_padded_outputs, steps)
    if fused:
        B, Te, _ = encoder_padded_outputs.shape
        if B > 32:                                            # 32 sequences per pass of the fused step; rows are independent
            parts = [greedy_search_graphed(decoder, encoder_padded_outputs[i:i + 32], steps=steps, fused=True) for i in range(0, B, 32)]
            n = max(t.shape[1] for t in parts)
            return torch.cat([torch.nn.functional.pad(t, (0, n - t.shape[1]), value=constant.EOS_TOKEN) for t in parts], dim=0)
        slot = (B, Te, steps, str(encoder_padded_outputs.device))
        held = getattr(decoder, "_asr_fused_decoders", None)
        if held is None:
            held = decoder._asr_fused_decoders = {}
        obj = held.get(slot)
        if obj is not None and obj.key == _weights_key(decoder):
            obj.reset(encoder_padded_outputs)
        else:
            if len(held) >= 2:                                # a full 32-row shape and a remainder shape: the buffers of a 32 x 300 decode are ~80 MB
                held.clear()
            obj = held[slot] = FusedGreedyDecoder(decoder, encoder_padded_outputs, max_len=steps)
        return obj.run(steps)
    return GraphedGreedyDecoder(decoder, encoder_padded_outputs, max_len=steps).run(steps)


@torch.no_grad()
def greedy_search(decoder, encoder_padded_outputs, steps=300, check_every=16):
    """Token ids (B, n<=steps) of the reference's greedy loop (transformer.py:316-394): argmax fed back for `steps`
    positions.  The loop stops early once every sequence has produced EOS -- what follows an EOS is never read."""
    B = encoder_padded_outputs.size(0)
    dev = encoder_padded_outputs.device
    cache = DecoderKVCache(decoder, encoder_padded_outputs, max_len=steps)
    tok = torch.full((B,), constant.SOS_TOKEN, dtype=torch.int64, device=dev)
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    out = []
    for i in range(steps):
        logits = cache.step(tok)
        tok = ops.argmax_rows(logits)
        out.append(tok)
        done |= tok.eq(constant.EOS_TOKEN)
        if (i + 1) % check_every == 0 and bool(done.all()):
            break
    return torch.stack(out, dim=1)
