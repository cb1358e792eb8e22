"""End-to-end Transformer ASR model -- same classes, constructor signatures, forward contracts and state_dict keys as
the reference (reference: models/asr/transformer.py), executed by libasr_hip.so.

Reference behaviours that are reproduced on purpose: `input_lengths` are PRE-CNN frame counts
compared against the POST-CNN time axis; every parameter with dim > 1 is re-initialised with xavier_uniform_ last;
decoder rows whose input token is EOS are zeroed; targets are always padded to --tgt-max-len; the encoder owns a Dropout
it never applies.
"""

import numpy as np
import torch
import torch.nn as nn

from asr_hip import functions as F_
from asr_hip import ops
from models.common_layers import (LowRankMultiHeadAttention, LowRankPositionwiseFeedForward, MultiHeadAttention,
                                  PositionalEncoding, PositionwiseFeedForwardWithConv)
from utils import constant


def _lengths_to_device(input_lengths, device):
    t = torch.as_tensor(input_lengths)
    return t.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()


class Transformer(nn.Module):
    """Transformer(encoder, decoder, feat_extractor='vgg_cnn')   (reference: transformer.py:16-57)"""

    def __init__(self, encoder, decoder, feat_extractor='vgg_cnn'):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.id2label = decoder.id2label
        self.feat_extractor = feat_extractor
        if feat_extractor == 'emb_cnn':
            self.conv = nn.Sequential(
                nn.Conv2d(1, 32, kernel_size=(41, 11), stride=(2, 2), padding=(0, 10)), nn.BatchNorm2d(32),
                nn.Hardtanh(0, 20, inplace=True),
                nn.Conv2d(32, 32, kernel_size=(21, 11), stride=(2, 1)), nn.BatchNorm2d(32),
                nn.Hardtanh(0, 20, inplace=True))
        elif feat_extractor == 'vgg_cnn':
            # indices 0,2,5,7 hold the parameters (state_dict keys conv.{0,2,5,7}.*); ReLU / pooling are fused in-kernel
            self.conv = nn.Sequential(
                nn.Conv2d(1, 64, 3, stride=1, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.ReLU(),
                nn.MaxPool2d(2, stride=2),
                nn.Conv2d(64, 128, 3, stride=1, padding=1), nn.ReLU(), nn.Conv2d(128, 128, 3, stride=1, padding=1),
                nn.ReLU(), nn.MaxPool2d(2, stride=2))
        for p in self.parameters():            # reference: transformer.py:55-57 (overrides every earlier init)
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    # -------------------------------------------------------------------------------------------- front end
    def _features(self, padded_input):
        """(B,1,F,T) -> (B,T',C*F') with feature index c*F'+f   (reference: transformer.py:70-76)"""
        if self.feat_extractor == 'vgg_cnn':
            c = self.conv
            return F_.VGGFn.apply(padded_input, c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[5].weight, c[5].bias,
                                  c[7].weight, c[7].bias)
        if self.feat_extractor == 'emb_cnn':
            c = self.conv
            return F_.EmbCNNFn.apply(padded_input, c[0].weight, c[0].bias, c[1].weight, c[1].bias, c[3].weight, c[3].bias,
                                     c[4].weight, c[4].bias, c[1], c[4], self.training)
        b, c, f, t = padded_input.shape
        return padded_input.reshape(b, c * f, t).transpose(1, 2).contiguous().to(ops.compute_dtype())

    def forward(self, padded_input, input_lengths, padded_target, verbose=False):
        """-> (pred (B,Td,V) fp32, gold (B,Td), hyp_seq (B,Td), gold_seq)   (reference: transformer.py:59-85)"""
        feats = self._features(padded_input)
        enc_out, _ = self.encoder(feats, input_lengths)
        pred, gold, *_ = self.decoder(padded_target, enc_out, input_lengths)
        hyp_seq = ops.argmax_rows(pred.detach().reshape(-1, pred.shape[-1])).view(pred.shape[0], pred.shape[1])
        return pred, gold, hyp_seq, gold

    def evaluate(self, padded_input, input_lengths, padded_target, beam_search=False, beam_width=0, beam_nbest=0, lm=None,
                 lm_rescoring=False, lm_weight=0.1, c_weight=1, verbose=False):
        """-> (_, strs_hyps, strs_gold)   (reference: transformer.py:87-124)"""
        feats = self._features(padded_input)
        enc_out, _ = self.encoder(feats, input_lengths)
        _, gold, *_ = self.decoder(padded_target, enc_out, input_lengths)
        gold_cpu = gold.cpu().tolist()
        strs_gold = ["".join(self.id2label[int(x)] for x in row) for row in gold_cpu]
        if beam_search:
            _, strs_hyps = self.decoder.beam_search(enc_out, beam_width=beam_width, nbest=1, lm=lm, lm_rescoring=lm_rescoring,
                                                    lm_weight=lm_weight, c_weight=c_weight)
            if len(strs_hyps) != padded_input.shape[0]:
                strs_hyps = self.decoder.greedy_search(enc_out)
        else:
            strs_hyps = self.decoder.greedy_search(enc_out)
        if verbose:
            print("GOLD", strs_gold)
            print("HYP", strs_hyps)
        return _, strs_hyps, strs_gold


class Encoder(nn.Module):
    """Encoder(num_layers, num_heads, dim_model, dim_key, dim_value, dim_input, dim_inner, dropout=0.1,
    src_max_length=2500)   (reference: transformer.py:126-180)"""

    def __init__(self, num_layers, num_heads, dim_model, dim_key, dim_value, dim_input, dim_inner, dropout=0.1,
                 src_max_length=2500, rank=0):
        super().__init__()
        self.dim_input, self.num_layers, self.num_heads = dim_input, num_layers, num_heads
        self.dim_model, self.dim_key, self.dim_value, self.dim_inner = dim_model, dim_key, dim_value, dim_inner
        self.src_max_length = src_max_length
        self.dropout = nn.Dropout(dropout)          # never applied (as in the reference, transformer.py:145)
        self.dropout_rate = dropout
        self.input_linear = nn.Linear(dim_input, dim_model)
        self.layer_norm_input = nn.LayerNorm(dim_model)
        self.positional_encoding = PositionalEncoding(dim_model, src_max_length)
        self.layers = nn.ModuleList([EncoderLayer(num_heads, dim_model, dim_inner, dim_key, dim_value, dropout=dropout, rank=rank)
                                     for _ in range(num_layers)])

    def forward(self, padded_input, input_lengths, need_attn=False):
        """padded_input (B,T,D_in), input_lengths (B) -> (output (B,T,D), [self_attn per layer])"""
        B, T, _ = padded_input.shape
        dev = padded_input.device
        lens = _lengths_to_device(input_lengths, dev)
        # row_keep[b,t] = t < len[b]   (reference: common_layers.py:33-38 via transformer.py:168)
        row_keep = ops.length_mask(lens, T)
        x = F_.EncInFn.apply(padded_input, self.input_linear.weight, self.input_linear.bias, self.layer_norm_input.weight,
                             self.layer_norm_input.bias, self.positional_encoding.pe[0])
        attns = []
        for layer in self.layers:
            x, a = layer(x, row_keep=row_keep, key_len=lens, need_attn=need_attn)
            attns.append(a)
        return x, attns


class EncoderLayer(nn.Module):
    """EncoderLayer(num_heads, dim_model, dim_inner, dim_key, dim_value, dropout=0.1)   (reference: transformer.py:183-203)"""

    def __init__(self, num_heads, dim_model, dim_inner, dim_key, dim_value, dropout=0.1, rank=0):
        super().__init__()
        if rank > 0:          # Low-Rank Transformer (BASELINE configs[4]): every projection is V (out,r) . U (r,in)
            self.self_attn = LowRankMultiHeadAttention(num_heads, dim_model, dim_key, dim_value, rank, dropout=dropout)
            self.pos_ffn = LowRankPositionwiseFeedForward(dim_model, dim_inner, rank, dropout=dropout)
            return
        self.self_attn = MultiHeadAttention(num_heads, dim_model, dim_key, dim_value, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForwardWithConv(dim_model, dim_inner, dropout=dropout)

    def forward(self, enc_input, non_pad_mask=None, self_attn_mask=None, row_keep=None, key_len=None, need_attn=False):
        if row_keep is None and non_pad_mask is not None:      # reference-style call with materialised masks
            row_keep = non_pad_mask.reshape(-1).ne(0).to(torch.uint8)
        out, attn = self.self_attn(enc_input, enc_input, enc_input, mask=self_attn_mask, key_len=key_len, row_keep=row_keep,
                                   need_attn=need_attn)
        out = self.pos_ffn(out, row_keep=row_keep)
        return out, attn


class Decoder(nn.Module):
    """Decoder(id2label, num_src_vocab, num_trg_vocab, num_layers, num_heads, dim_emb, dim_model, dim_inner, dim_key,
    dim_value, dropout=0.1, trg_max_length=1000, emb_trg_sharing=False)   (reference: transformer.py:206-305)"""

    def __init__(self, id2label, num_src_vocab, num_trg_vocab, num_layers, num_heads, dim_emb, dim_model, dim_inner, dim_key,
                 dim_value, dropout=0.1, trg_max_length=1000, emb_trg_sharing=False, rank=0):
        super().__init__()
        self.sos_id, self.eos_id = constant.SOS_TOKEN, constant.EOS_TOKEN
        self.id2label = id2label
        self.num_src_vocab, self.num_trg_vocab = num_src_vocab, num_trg_vocab
        self.num_layers, self.num_heads = num_layers, num_heads
        self.dim_emb, self.dim_model, self.dim_inner = dim_emb, dim_model, dim_inner
        self.dim_key, self.dim_value = dim_key, dim_value
        self.dropout_rate, self.emb_trg_sharing, self.trg_max_length = dropout, emb_trg_sharing, trg_max_length
        self.trg_embedding = nn.Embedding(num_trg_vocab, dim_emb, padding_idx=constant.PAD_TOKEN)
        self.positional_encoding = PositionalEncoding(dim_model, max_length=trg_max_length)
        self.dropout = nn.Dropout(dropout)
        self.layers = nn.ModuleList([DecoderLayer(dim_model, dim_inner, num_heads, dim_key, dim_value, dropout=dropout, rank=rank)
                                     for _ in range(num_layers)])
        self.output_linear = nn.Linear(dim_model, num_trg_vocab, bias=False)
        nn.init.xavier_normal_(self.output_linear.weight)
        # hint for the flat-parameter layout (utils/optimizer.py:_slot_order): every layer's cross-attention K / V projection reads the
        # SAME encoder output (reference: transformer.py:296-299, 533-537), so adjacent weights make them one GEMM (forward below)
        if rank == 0:
            for li, layer in enumerate(self.layers):
                ea = layer.encoder_attn
                ea.key_linear.weight._asr_cross_kv = ("w", 2 * li)
                ea.value_linear.weight._asr_cross_kv = ("w", 2 * li + 1)
                ea.key_linear.bias._asr_cross_kv = ("b", 2 * li)
                ea.value_linear.bias._asr_cross_kv = ("b", 2 * li + 1)
        if emb_trg_sharing:
            self.output_linear.weight = self.trg_embedding.weight
            self.x_logit_scale = dim_model ** -0.5
        else:
            self.x_logit_scale = 1.0

    def preprocess(self, padded_input):
        """(B,L) -> seq_in_pad, seq_out_pad (B,Td)   (reference: transformer.py:254-266); Td = --tgt-max-len."""
        Td = constant.args.tgt_max_len
        seq_in, seq_out, key_pad, row_keep, overflow = ops.decoder_preprocess(padded_input, Td)
        if padded_input.shape[1] + 1 > Td and int(overflow.item()) != 0:
            raise RuntimeError("a target needs more than --tgt-max-len=%d positions (reference pad_list would fail, "
                               "common_layers.py:21)" % Td)
        self._masks = (key_pad, row_keep)
        return seq_in, seq_out

    def forward(self, padded_input, encoder_padded_outputs, encoder_input_lengths, need_attn=False):
        """-> (pred (B,Td,V) fp32, gold (B,Td), [self_attn], [enc_attn])   (reference: transformer.py:268-305)"""
        seq_in, seq_out = self.preprocess(padded_input)
        key_pad, row_keep = self._masks
        row_keep = row_keep.reshape(-1)
        dev = seq_in.device
        enc_len = _lengths_to_device(encoder_input_lengths, dev)
        tied = self.emb_trg_sharing
        p = self.dropout.p if self.training else 0.0
        x = F_.EmbedFn.apply(seq_in, self.trg_embedding.weight, self.positional_encoding.pe[0], self.x_logit_scale, p,
                             constant.PAD_TOKEN, True)
        self_attns, enc_attns = [], []
        # the cross-attention K | V projections of every layer as ONE GEMM on the encoder output (and one data-gradient GEMM, one weight-gradient
        # problem in backward) where the flat parameter layout has their weights adjacent; else one gradient buffer for the encoder output
        # that the layers' cross-attention backward GEMMs accumulate into
        box = None
        kv_pre = F_.cross_kv_all(encoder_padded_outputs, self.layers) if len(self.layers) > 1 else None
        enc_views = [encoder_padded_outputs] * len(self.layers)
        if kv_pre is None and torch.is_grad_enabled() and encoder_padded_outputs.requires_grad and len(self.layers) > 1:
            box = {}
            enc_views = F_.FanOutFn.apply(encoder_padded_outputs, len(self.layers), box)
        for li, layer in enumerate(self.layers):
            x, sa, ea = layer(x, enc_views[li], row_keep=row_keep, self_key_pad=key_pad, enc_key_len=enc_len,
                              need_attn=need_attn, kv_grad_box=box, kv_pre=None if kv_pre is None else (kv_pre[0][li], kv_pre[1], li))
            self_attns.append(sa)
            enc_attns.append(ea)
        # with --emb_trg_sharing the embedding backward (which runs last) reports the shared weight as ready
        pred = F_.linear(x, self.output_linear.weight, None, True, not tied)
        return pred, seq_out, self_attns, enc_attns

    def post_process_hyp(self, hyp):
        return "".join(self.id2label[int(x)] for x in hyp['yseq'][1:])

    # ---- decoding: KV-cached by default, the reference's full re-run kept for equivalence tests ----
    def _step_logits(self, ys, encoder_padded_outputs):
        """Teacher-forced decoder pass over the prefix `ys` (B,t) with the reference's decode-time masks: causal only,
        no encoder-length mask (reference: transformer.py:336-350, dec_enc_attn_mask=None)."""
        p = self.dropout.p if self.training else 0.0
        x = F_.EmbedFn.apply(ys, self.trg_embedding.weight, self.positional_encoding.pe[0], self.x_logit_scale, p,
                             constant.PAD_TOKEN, True)
        for layer in self.layers:
            x, _, _ = layer(x, encoder_padded_outputs, causal_only=True)
        return F_.linear(x, self.output_linear.weight, None, True, False)

    def _kv_cache_supported(self):
        """The incremental decoders (asr_hip/decode.py) read the full-rank projection weights of every layer; the Low-Rank
        Transformer (--rank > 0: LowRankLinear holds .u / .v, no .weight) decodes by re-running the layer modules over the
        prefix instead, like the reference's own loop.  So does a model with dim_key != dim_value (the caches hold H * dim_key columns for
        keys and values alike)."""
        return (self.dim_key == self.dim_value and
                all(isinstance(l.self_attn, MultiHeadAttention) and isinstance(l.encoder_attn, MultiHeadAttention) for l in self.layers))

    @torch.no_grad()
    def greedy_search(self, encoder_padded_outputs, beam_width=2, lm_rescoring=False, lm=None, lm_weight=0.1, c_weight=1,
                      use_cache=True):
        """1-best strings of the reference's 300-step greedy loop (transformer.py:316-394).  Needs --tgt-max-len >= 301.
        use_cache=True decodes incrementally with per-layer key/value caches, one captured hipGraph replayed per token
        (asr_hip/decode.py; in bf16 the step is the 30-launch one of csrc/decode.hip when the shapes allow it); "graph" the same
        with the kernel-per-op step; "eager" that step without the graph; False re-runs the full decoder over the prefix at
        every step like the reference -- "graph" / "eager" / False give the same tokens (tests/test_gpu_decode.py), the fused
        step the same within the bf16 tolerance (tests/test_gpu_decode_fused.py)."""
        if lm_rescoring:
            raise NotImplementedError("LM rescoring is outside the accelerated path (SURVEY.md section 2, row 12)")
        if not self._kv_cache_supported():
            use_cache = False
        if use_cache == "eager":                      # cached, launches issued from Python per token
            from asr_hip.decode import greedy_search as cached_greedy
            toks = cached_greedy(self, encoder_padded_outputs, steps=300).cpu().tolist()
        elif use_cache:                               # cached + one hipGraph replay per token (device-side position)
            from asr_hip.decode import greedy_search_graphed
            fused = False if use_cache == "graph" else None      # "graph": the kernel-per-op step; True: 30-launch step in bf16
            toks = greedy_search_graphed(self, encoder_padded_outputs, steps=300, fused=fused).cpu().tolist()
        else:
            B = encoder_padded_outputs.size(0)
            ys = torch.full((B, 1), constant.SOS_TOKEN, dtype=torch.int64, device=encoder_padded_outputs.device)
            steps = []
            for _ in range(300):
                logits = self._step_logits(ys, encoder_padded_outputs)
                nxt = ops.argmax_rows(logits[:, -1].contiguous())
                steps.append(nxt)
                ys = torch.cat([ys, nxt.unsqueeze(1)], dim=1)
            toks = torch.stack(steps, dim=1).cpu().tolist()       # one D2H copy instead of per-token .item()
        sents = []
        for row in toks:
            st = ''
            for t in row:
                if t == constant.EOS_TOKEN:
                    break
                st += self.id2label[t]
            sents.append(st)
        return sents

    @torch.no_grad()
    def beam_search(self, encoder_padded_outputs, beam_width=2, nbest=5, lm_rescoring=False, lm=None, lm_weight=0.1,
                    c_weight=1, prob_weight=1.0, use_cache=True):
        """Per-utterance beam search with the reference's scoring (transformer.py:396-517, LM branch excluded).  With
        use_cache the live hypotheses of an utterance are one batch of the KV-cached decoder (one step = one token per
        hypothesis); the candidate bookkeeping on the host is the reference's, including its in-loop re-sort (:460).
        With more than one utterance the cached search runs for all of them at once (_beam_search_batched);
        use_cache="per_utterance" keeps the utterance loop."""
        if lm_rescoring:
            raise NotImplementedError("LM rescoring is outside the accelerated path (SURVEY.md section 2, row 12)")
        from asr_hip.decode import DecoderKVCache
        if not self._kv_cache_supported():
            use_cache = False
        if use_cache and use_cache != "per_utterance" and encoder_padded_outputs.size(0) > 1:
            return self._beam_search_batched(encoder_padded_outputs, beam_width, nbest, c_weight)
        ids_out, strs_out = [], []
        max_len = encoder_padded_outputs.size(1)
        dev = encoder_padded_outputs.device
        for b in range(encoder_padded_outputs.size(0)):
            enc = encoder_padded_outputs[b:b + 1]
            hyps = [{'score': 0.0, 'yseq': [constant.SOS_TOKEN]}]
            ended = []
            cache = DecoderKVCache(self, enc, max_len=300, batch=1) if use_cache else None
            for i in range(300):
                if use_cache:
                    last = torch.tensor([h['yseq'][-1] for h in hyps], dtype=torch.int64, device=dev)
                    best_all, idx_all = ops.logsoftmax_topk(cache.step(last).float().contiguous(), beam_width)
                    best_all, idx_all = best_all.tolist(), idx_all.tolist()
                cand = []
                for hi, hyp in enumerate(hyps):
                    if use_cache:
                        best, idx = best_all[hi], idx_all[hi]
                    else:
                        ys = torch.tensor([hyp['yseq']], dtype=torch.int64, device=dev)
                        logits = self._step_logits(ys, enc)[:, -1]
                        best, idx = ops.logsoftmax_topk(logits.float().contiguous(), beam_width)
                        best, idx = best[0].tolist(), idx[0].tolist()
                    for j in range(beam_width):
                        cand.append({'score': hyp['score'] + best[j], 'yseq': hyp['yseq'] + [idx[j]], 'parent': hi})
                    # the reference re-sorts the running candidate list inside the hypothesis loop (:460)
                    cand = sorted(cand, key=lambda h: h['score'], reverse=True)[:beam_width]
                hyps = cand
                if i == max_len - 1:
                    for hyp in hyps:
                        hyp['yseq'] = hyp['yseq'] + [constant.EOS_TOKEN]
                alive = []
                for hyp in hyps:
                    if hyp['yseq'][-1] == constant.EOS_TOKEN:
                        ended.append(self._finish_hyp(hyp, c_weight))
                    else:
                        alive.append(hyp)
                hyps = alive
                if not hyps:
                    break
                if use_cache:
                    cache.select([h['parent'] for h in hyps])
            for hyp in sorted(ended, key=lambda h: h['final_score'], reverse=True)[:min(len(ended), nbest)]:
                ids_out.append(hyp['yseq'])
                strs_out.append(self.post_process_hyp(hyp))
        return ids_out, strs_out


    def _finish_hyp(self, hyp, c_weight):
        import math
        s = "".join(self.id2label[t] for t in hyp['yseq'])
        for ch in (constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR):
            s = s.replace(ch, "")
        s = s.replace("  ", " ")
        hyp['final_score'] = hyp['score'] + math.sqrt(len(s.split())) * c_weight
        return hyp

    def _beam_search_batched(self, encoder_padded_outputs, beam_width, nbest, c_weight):
        """The same search for ALL utterances of the batch at once: utterance b owns decoder rows b * W .. b * W + W - 1 of ONE
        KV-cached decoder batch (its live hypotheses in the first rows, the others idle), so a step is one decoder step, one
        log-softmax / top-W launch and one device -> host copy for the whole batch instead of one of each per utterance.  The
        candidate bookkeeping per utterance is the reference's (transformer.py:437-497: the in-loop re-sort, forced EOS at the
        last encoder frame, sqrt(words) * c_weight on finished hypotheses), so the strings are those of the per-utterance loop."""
        from asr_hip.decode import DecoderKVCache
        W = beam_width
        B, max_len = encoder_padded_outputs.size(0), encoder_padded_outputs.size(1)
        dev = encoder_padded_outputs.device
        cache = DecoderKVCache(self, encoder_padded_outputs.repeat_interleave(W, dim=0), max_len=300)
        hyps = [[{'score': 0.0, 'yseq': [constant.SOS_TOKEN]}] for _ in range(B)]
        ended = [[] for _ in range(B)]
        for i in range(300):
            last = [constant.SOS_TOKEN] * (B * W)
            for b in range(B):
                for hi, h in enumerate(hyps[b]):
                    last[b * W + hi] = h['yseq'][-1]
            logits = cache.step(torch.tensor(last, dtype=torch.int64, device=dev))
            best_all, idx_all = ops.logsoftmax_topk(logits.float().contiguous(), W)
            best_all, idx_all = best_all.tolist(), idx_all.tolist()
            rows = list(range(B * W))
            moved = False
            for b in range(B):
                if not hyps[b]:
                    continue
                cand = []
                for hi, hyp in enumerate(hyps[b]):
                    best, idx = best_all[b * W + hi], idx_all[b * W + hi]
                    for j in range(W):
                        cand.append({'score': hyp['score'] + best[j], 'yseq': hyp['yseq'] + [idx[j]], 'parent': hi})
                    cand = sorted(cand, key=lambda h: h['score'], reverse=True)[:W]      # the reference's in-loop re-sort (:460)
                if i == max_len - 1:
                    for hyp in cand:
                        hyp['yseq'] = hyp['yseq'] + [constant.EOS_TOKEN]
                alive = []
                for hyp in cand:
                    if hyp['yseq'][-1] == constant.EOS_TOKEN:
                        ended[b].append(self._finish_hyp(hyp, c_weight))
                    else:
                        alive.append(hyp)
                hyps[b] = alive
                for j, hyp in enumerate(alive):
                    moved |= hyp['parent'] != j
                    rows[b * W + j] = b * W + hyp['parent']
            if not any(hyps):
                break
            if moved:
                cache.select(rows, cross=False)           # parents stay inside their utterance: the cross keys / values do not move
        ids_out, strs_out = [], []
        for b in range(B):
            for hyp in sorted(ended[b], key=lambda h: h['final_score'], reverse=True)[:min(len(ended[b]), nbest)]:
                ids_out.append(hyp['yseq'])
                strs_out.append(self.post_process_hyp(hyp))
        return ids_out, strs_out


class DecoderLayer(nn.Module):
    """DecoderLayer(dim_model, dim_inner, num_heads, dim_key, dim_value, dropout=0.1)   (reference: transformer.py:519-545)"""

    def __init__(self, dim_model, dim_inner, num_heads, dim_key, dim_value, dropout=0.1, rank=0):
        super().__init__()
        if rank > 0:
            self.self_attn = LowRankMultiHeadAttention(num_heads, dim_model, dim_key, dim_value, rank, dropout=dropout)
            self.encoder_attn = LowRankMultiHeadAttention(num_heads, dim_model, dim_key, dim_value, rank, dropout=dropout)
            self.pos_ffn = LowRankPositionwiseFeedForward(dim_model, dim_inner, rank, dropout=dropout)
            return
        self.self_attn = MultiHeadAttention(num_heads, dim_model, dim_key, dim_value, dropout=dropout)
        self.encoder_attn = MultiHeadAttention(num_heads, dim_model, dim_key, dim_value, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForwardWithConv(dim_model, dim_inner, dropout=dropout)

    def forward(self, decoder_input, encoder_output, non_pad_mask=None, self_attn_mask=None, dec_enc_attn_mask=None,
                row_keep=None, self_key_pad=None, enc_key_len=None, causal_only=False, need_attn=False, kv_grad_box=None, kv_pre=None):
        if causal_only:
            x, sa = self.self_attn(decoder_input, decoder_input, decoder_input, causal=True, need_attn=need_attn)
            x, ea = self.encoder_attn(x, encoder_output, encoder_output, need_attn=need_attn)
            return self.pos_ffn(x), sa, ea
        if row_keep is None and non_pad_mask is not None:      # reference-style call with materialised masks
            row_keep = non_pad_mask.reshape(-1).ne(0).to(torch.uint8)
        generic = self_key_pad is None and self_attn_mask is not None
        x, sa = self.self_attn(decoder_input, decoder_input, decoder_input, mask=self_attn_mask if generic else None,
                               key_pad=self_key_pad, causal=not generic, row_keep=row_keep, need_attn=need_attn)
        if kv_pre is not None:
            x, ea = self.encoder_attn(x, encoder_output, encoder_output, mask=dec_enc_attn_mask, key_len=enc_key_len,
                                      row_keep=row_keep, need_attn=need_attn, kv_pre=kv_pre)
        else:
            x, ea = self.encoder_attn(x, encoder_output, encoder_output, mask=dec_enc_attn_mask, key_len=enc_key_len,
                                      row_keep=row_keep, need_attn=need_attn, kv_grad_box=kv_grad_box)
        x = self.pos_ffn(x, row_keep=row_keep)
        return x, sa, ea
