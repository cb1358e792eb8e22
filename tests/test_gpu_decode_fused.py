"""The 34-launch greedy step (csrc/decode.hip: asr_dec_gemm / asr_dec_attn / asr_dec_finish) against plain fp32 torch
restatements of the same operators on the same bf16 operands, and against the kernel-per-op cached step
(asr_hip/decode.py: DecoderKVCache, itself pinned to the reference's strings in tests/test_gpu_decode.py).
Tolerances: bf16 outputs within 2 bf16 ulp of the fp32 result (fp32 accumulation, different summation order); logits of a
whole 2-layer step within 3e-2 of the logit range."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _ulp_close(a, ref, ulps=2.0):
    a, ref = a.float(), ref.float()
    tol = ulps * 2.0 ** -8 * ref.abs().clamp_min(1e-2)
    bad = (a - ref).abs() > tol
    assert not bad.any(), ((a - ref).abs().max().item(), int(bad.sum()))


@pytest.mark.parametrize("B,N,K,relu,f32out", [(32, 512, 512, False, False), (5, 96, 2048, True, False), (32, 4364, 512, False, True),
                                                (1, 100, 64, False, False), (17, 1536, 256, False, False)])
def test_dec_gemm_plain(B, N, K, relu, f32out):
    from asr_hip import ops
    g = torch.Generator().manual_seed(B * 131 + N)
    x = torch.randn(B, K, generator=g).to(bf).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).cuda()
    bias = None if f32out else torch.randn(N, generator=g).cuda()
    Np = (N + 3) // 4 * 4
    out = torch.full((B, Np), 7.0, dtype=torch.float32 if f32out else bf, device="cuda")[:, :N]
    ops.dec_gemm(W, bias, out, x=x, relu=relu)
    ref = x.float() @ W.float().t()
    if bias is not None:
        ref = ref + bias
    if relu:
        ref = ref.relu()
    if f32out:
        assert (out - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-5
    else:
        _ulp_close(out, ref)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,N,K", [(32, 512, 2048), (9, 2048, 512), (5, 4364, 512)])
def test_dec_gemm_fragment_major_operands(B, N, K):
    """Packed weight, packed input rows and packed output are the same numbers in the order of include/asr_hip.h."""
    from asr_hip import ops
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(B, K, generator=g).to(bf).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).cuda()
    bias = torch.randn(N, generator=g).cuda()
    plain = torch.zeros(B, N, dtype=bf, device="cuda")
    ops.dec_gemm(W, bias, plain, x=x, relu=True)
    Wf = ops.frag_pack(W)
    assert torch.equal(ops.frag_unpack(Wf, N, K), W)
    xf = ops.frag_pack(x)                                  # rows >= B: zeros
    a = torch.zeros(B, N, dtype=bf, device="cuda")
    ops.dec_gemm(Wf, bias, a, x=xf, relu=True, w_frag=(N, K), x_frag=True)
    assert torch.equal(a, plain)
    if N % 16 == 0:
        of = torch.zeros(32 * N, dtype=bf, device="cuda")
        ops.dec_gemm(Wf, bias, of, x=x, relu=True, w_frag=(N, K), out_frag=True)
        assert torch.equal(ops.frag_unpack(of, 32, N)[:B], plain)


@pytest.mark.parametrize("B,K", [(32, 512), (7, 256), (3, 64)])
def test_dec_gemm_layernorm_prologue_equals_add_ln_then_gemm(B, K):
    from asr_hip import ops
    g = torch.Generator().manual_seed(K + B)
    N = 160
    Y = torch.randn(B, K, generator=g).to(bf).cuda()
    R = (2 * torch.randn(B, K, generator=g)).to(bf).cuda()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.3 * torch.randn(K, generator=g)).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = torch.zeros(B, N, dtype=bf, device="cuda")
    x_out = torch.zeros(B, K, dtype=bf, device="cuda")
    ops.dec_gemm(W, bias, out, ln=(Y, R, gamma, beta, 1e-5), x_out=x_out, relu=True)
    # the stand-alone kernel the prologue replaces
    x_k, _, _ = ops.add_ln_fwd(Y.clone(), R, gamma, beta)
    assert torch.equal(x_out, x_k)
    z = (Y.float() + R.float()).to(bf).float()
    x_ref = torch.nn.functional.layer_norm(z, (K,), gamma, beta, 1e-5)
    _ulp_close(x_out, x_ref, ulps=1.01)
    ref = (x_out.float() @ W.float().t() + bias).relu()
    _ulp_close(out, ref)


def test_dec_gemm_embedding_prologue():
    from asr_hip import ops
    g = torch.Generator().manual_seed(11)
    B, K, N, Vt, T = 9, 512, 1536, 50, 20
    table = torch.randn(Vt, K, generator=g).cuda()
    pe = torch.randn(T, K, generator=g).cuda()
    tok = torch.randint(0, Vt, (B,), generator=g).cuda()
    state = torch.tensor([13, 0], dtype=torch.int64, device="cuda")
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = torch.zeros(B, N, dtype=bf, device="cuda")
    x_out = torch.zeros(B, K, dtype=bf, device="cuda")
    ops.dec_gemm(W, bias, out, embed=(tok, table, pe, 0.5, state), x_out=x_out)
    x_ref = (table[tok] * 0.5 + pe[13]).to(bf)
    assert torch.equal(x_out, x_ref)
    _ulp_close(out, x_ref.float() @ W.float().t() + bias)


def _attn_ref(q, K, V, H, scale):
    B, HD = q.shape
    d = HD // H
    qh = q.float().view(B, H, 1, d)
    Kh = K.float().view(B, -1, H, d).transpose(1, 2)
    Vh = V.float().view(B, -1, H, d).transpose(1, 2)
    p = torch.softmax((qh @ Kh.transpose(2, 3)) * scale, dim=-1)
    return (p @ Vh).reshape(B, HD)


@pytest.mark.parametrize("t", [0, 17, 63, 64, 299, 511, 512, 700, 1100])
def test_dec_attn_self_appends_and_attends(t):
    """positions >= 512: the kernel walks the cache in passes of 512 keys with a running maximum"""
    from asr_hip import ops
    g = torch.Generator().manual_seed(t)
    B, H, d, max_len = 5, 8, 64, (300 if t < 300 else 1200)
    HD = H * d
    kc = torch.randn(B, max_len, HD, generator=g).to(bf).cuda()
    vc = torch.randn(B, max_len, HD, generator=g).to(bf).cuda()
    kc0, vc0 = kc.clone(), vc.clone()
    qkv = torch.randn(B, 3 * HD, generator=g).to(bf).cuda()
    state = torch.tensor([t, 0], dtype=torch.int64, device="cuda")
    out = torch.zeros(B, HD, dtype=bf, device="cuda")
    ops.dec_attn(qkv[:, :HD], kc, vc, out, H, d, d ** -0.5, k_new=qkv[:, HD:2 * HD], v_new=qkv[:, 2 * HD:], state=state)
    assert torch.equal(kc[:, t], qkv[:, HD:2 * HD]) and torch.equal(vc[:, t], qkv[:, 2 * HD:])
    keep = torch.ones(max_len, dtype=torch.bool, device="cuda")
    keep[t] = False
    assert torch.equal(kc[:, keep], kc0[:, keep]) and torch.equal(vc[:, keep], vc0[:, keep])
    ref = _attn_ref(qkv[:, :HD], kc[:, :t + 1], vc[:, :t + 1], H, d ** -0.5)
    assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("rows,shared", [(200, False), (37, True), (512, False), (513, False), (795, False), (1300, True)])
def test_dec_attn_cross(rows, shared):
    from asr_hip import ops
    g = torch.Generator().manual_seed(rows)
    B, H, d = 6, 8, 64
    HD = H * d
    Bk = 1 if shared else B
    K = torch.randn(Bk, rows, HD, generator=g).to(bf).cuda()
    V = torch.randn(Bk, rows, HD, generator=g).to(bf).cuda()
    if shared:
        K, V = K.expand(B, rows, HD), V.expand(B, rows, HD)
    q = torch.randn(B, HD, generator=g).to(bf).cuda()
    out = torch.zeros(B, HD, dtype=bf, device="cuda")
    ops.dec_attn(q, K, V, out, H, d, d ** -0.5)
    ref = _attn_ref(q, K, V, H, d ** -0.5)
    assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    of = torch.zeros(32 * HD, dtype=bf, device="cuda")
    ops.dec_attn(q, K, V, of, H, d, d ** -0.5, out_frag=True)
    assert torch.equal(ops.frag_unpack(of, 32, HD)[:B], out)


@pytest.mark.parametrize("mode,t,D", [("ln", 17, 512), ("ln", 0, 512), ("embed", 5, 512), ("ln", 299, 256), ("cross", -1, 512), ("cross", -1, 128)])
def test_dec_attn_fused_equals_prologue_gemm_then_attention(mode, t, D):
    """asr_dec_attn_fused against asr_dec_gemm (LayerNorm / embedding prologue) followed by asr_dec_attn: the same input row, the
    same cache rows, the attention output within 2 bf16 ulp (the projections are fp32 dot products in another order)."""
    from asr_hip import ops
    g = torch.Generator().manual_seed(t + D)
    B, H, d, rows = 6, 8, 64, 300
    HD = H * d
    self_attn = mode != "cross"
    NP = 3 if self_attn else 1
    W = (torch.randn(NP * HD, D, generator=g) * D ** -0.5).to(bf).cuda()
    bias = torch.randn(NP * HD, generator=g).cuda()
    kc = torch.randn(B, rows, HD, generator=g).to(bf).cuda()
    vc = torch.randn(B, rows, HD, generator=g).to(bf).cuda()
    if self_attn:                                    # rows >= t are uninitialised memory in a real decode: poison them
        kc[:, t:], vc[:, t:] = float("nan"), float("nan")
    kc2, vc2 = kc.clone(), vc.clone()
    state = torch.tensor([max(t, 0), 0], dtype=torch.int64, device="cuda")
    src = {}
    if mode == "embed":
        table, pe = torch.randn(50, D, generator=g).cuda(), torch.randn(rows, D, generator=g).cuda()
        tok = torch.randint(0, 50, (B,), generator=g).cuda()
        src_f, src_g = dict(embed=(tok, table, pe, 0.7)), dict(embed=(tok, table, pe, 0.7, state))
    else:
        Y, R = torch.randn(B, D, generator=g).to(bf).cuda(), (2 * torch.randn(B, D, generator=g)).to(bf).cuda()
        gamma, beta = (1 + 0.2 * torch.randn(D, generator=g)).cuda(), (0.3 * torch.randn(D, generator=g)).cuda()
        src_f = src_g = dict(ln=(Y, R, gamma, beta, 1e-5))
    # two launches
    qkv = torch.zeros(B, NP * HD, dtype=bf, device="cuda")
    x_ref = torch.zeros(B, D, dtype=bf, device="cuda")
    ops.dec_gemm(W, bias, qkv, x_out=x_ref, **src_g)
    o_ref = torch.zeros(B, HD, dtype=bf, device="cuda")
    if self_attn:
        ops.dec_attn(qkv[:, :HD], kc, vc, o_ref, H, d, d ** -0.5, k_new=qkv[:, HD:2 * HD], v_new=qkv[:, 2 * HD:], state=state)
    else:
        ops.dec_attn(qkv, kc, vc, o_ref, H, d, d ** -0.5)
    # one launch
    o = torch.zeros(B, HD, dtype=bf, device="cuda")
    x = torch.zeros(B, D, dtype=bf, device="cuda")
    ops.dec_attn_fused(W, bias, kc2, vc2, o, H, d, d ** -0.5, x_out=x, state=state if self_attn else None, self_attention=self_attn,
                       **src_f)
    assert torch.equal(x, x_ref)
    if self_attn:
        _ulp_close(kc2[:, t], kc[:, t])
        _ulp_close(vc2[:, t], vc[:, t])
        assert torch.equal(kc2[:, :t], kc[:, :t]) and torch.equal(vc2[:, :t], vc[:, :t]) and torch.isnan(kc2[:, t + 1:].float()).all()
    assert torch.isfinite(o.float()).all() and torch.isfinite(o_ref.float()).all()
    assert (o.float() - o_ref.float()).abs().max().item() <= 2e-2 * o_ref.float().abs().max().item()
    of = torch.zeros(32 * HD, dtype=bf, device="cuda")
    ops.dec_attn_fused(W, bias, kc2, vc2, of, H, d, d ** -0.5, state=state if self_attn else None, self_attention=self_attn, out_frag=True,
                       **src_f)
    assert torch.equal(ops.frag_unpack(of, 32, HD)[:B], o)


def test_dec_finish_argmax_done_output_and_position():
    from asr_hip import ops
    B, V, max_len = 6, 4364, 10
    logits = torch.randn(B, V).cuda()
    logits[1, 7] = logits[1, 4000] = 50.0            # tie: lowest index
    logits[2, 2] = 60.0                              # EOS
    tok = torch.zeros(B, dtype=torch.int64, device="cuda")
    done = torch.zeros(B, dtype=torch.bool, device="cuda")
    done[4] = True                                   # stays set
    out = torch.zeros(max_len, B, dtype=torch.int64, device="cuda")
    state = torch.tensor([3, 0], dtype=torch.int64, device="cuda")
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.dec_finish(logits, tok, done, out, 2, state, ticket)
    ref = logits.argmax(1)
    ref[1] = 7
    assert torch.equal(tok, ref) and torch.equal(out[3], ref) and int(out.sum()) == int(ref.sum())
    assert done.tolist() == [False, False, True, False, True, False]
    assert state.tolist() == [4, 0] and ticket.item() == 0
    ops.dec_finish(logits, tok, done, out, 2, state, ticket)
    assert state.tolist() == [5, 0] and torch.equal(out[4], ref)


def _model(layers=2, inner=256, V=40):
    from utils import constant
    from utils.functions import init_transformer_model
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x61 + i) for i in range(V - 3)]
    l2i = {c: i for i, c in enumerate(chars)}
    args = constant.parse(["--num-layers", str(layers), "--num-heads", "8", "--dim-model", "512", "--dim-key", "64",
                           "--dim-value", "64", "--dim-inner", str(inner), "--dim-emb", "512", "--feat_extractor", "vgg_cnn",
                           "--tgt-max-len", "301", "--src-max-len", "64", "--dropout", "0.1", "--precision", "bf16", "--cuda"])
    torch.manual_seed(7)
    model = init_transformer_model(args, l2i, {i: c for c, i in l2i.items()}).cuda().eval()
    with torch.no_grad():
        model.decoder.output_linear.weight[2] += 0.02
    return model


@pytest.mark.parametrize("fuse_cross", [True, False])
def test_fused_step_logits_match_kernel_per_op_step(fuse_cross, monkeypatch):
    """Teacher forcing: the same tokens through both steps -> logits within 3e-2 of the logit range at every position."""
    from asr_hip.decode import DecoderKVCache, FusedGreedyDecoder, fused_decode_supported
    monkeypatch.setattr(FusedGreedyDecoder, "FUSE_CROSS", fuse_cross)
    model = _model()
    dec = model.decoder
    g = torch.Generator().manual_seed(3)
    B, Te, T = 7, 37, 24
    enc = torch.randn(B, Te, 512, generator=g).cuda()
    ys = torch.randint(3, 40, (B, T), generator=g).cuda()
    ys[:, 0] = 1
    assert fused_decode_supported(dec, enc, T)
    slow = DecoderKVCache(dec, enc, max_len=T)
    fast = FusedGreedyDecoder(dec, enc, max_len=T)
    worst = 0.0
    for t in range(T):
        a = slow.step(ys[:, t].contiguous()).float()
        b = fast.step_logits(ys[:, t]).float()
        worst = max(worst, ((a - b).abs().max() / a.abs().max()).item())
    assert worst <= 3e-2, worst
    assert fast.state[0].item() == T
    # the self-attention caches hold the same rows (bf16 values of the same projections)
    for i in range(len(slow.self_k)):
        d = (slow.self_k[i][:, :T].float() - fast.cache.self_k[i][:, :T].float()).abs().max().item()
        assert d <= 3e-2 * slow.self_k[i][:, :T].float().abs().max().item()


def test_fused_greedy_graph_replay_equals_eager_and_follows_the_per_op_argmax():
    from asr_hip.decode import DecoderKVCache, FusedGreedyDecoder
    model = _model(layers=2, inner=2048, V=200)
    dec = model.decoder
    g = torch.Generator().manual_seed(5)
    B, steps = 4, 40
    enc = torch.randn(B, 50, 512, generator=g).cuda()
    B, steps = 4, 45
    toks = FusedGreedyDecoder(dec, enc, max_len=steps).run(steps, check_every=1000)          # 2 eager steps + 5 replays of 8 + 3 eager
    eager = FusedGreedyDecoder(dec, enc, max_len=steps)
    for _ in range(steps):
        eager._step()
    assert torch.equal(toks, eager.out.t()) and toks.shape == (B, steps)
    # every emitted token is the per-op step's arg max for the same prefix, or within the bf16 tolerance of it
    slow = DecoderKVCache(dec, enc, max_len=steps)
    prev = torch.full((B,), 1, dtype=torch.int64, device="cuda")
    for t in range(steps):
        lg = slow.step(prev).float()
        top = lg.max(1).values
        got = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        assert ((top - got) <= 3e-2 * lg.abs().max()).all(), t
        prev = toks[:, t].contiguous()
    # the public entry point picks this path for bf16 at these shapes, and a second call with the same shapes reuses the
    # captured graph with fresh encoder keys / values
    from asr_hip.decode import greedy_search_graphed
    a1 = greedy_search_graphed(dec, enc, steps=steps)
    assert a1.shape[1] <= steps and torch.equal(a1, toks[:, :a1.shape[1]])      # (stops once every row has emitted EOS)
    enc2 = torch.randn(B, 50, 512, generator=g).cuda()
    held = dec._asr_fused_decoders[(B, 50, steps, str(enc.device))]
    a2 = greedy_search_graphed(dec, enc2, steps=steps)
    assert dec._asr_fused_decoders[(B, 50, steps, str(enc.device))] is held
    fresh = FusedGreedyDecoder(dec, enc2, max_len=steps)
    for _ in range(steps):
        fresh._step()
    assert torch.equal(a2, fresh.out.t()[:, :a2.shape[1]]) and not torch.equal(a2[:, :8], a1[:, :8])
    strs = dec.greedy_search(enc[:2], use_cache=True)
    assert len(strs) == 2


def test_greedy_beyond_32_sequences_and_512_encoder_frames():
    """configs[3]'s decode shape class (795 encoder frames) and more than 32 utterances stay on the fused step: the batch is decoded
    32 sequences at a time, cross attention walks the 600 keys in two passes.  Rows are independent, so every row equals the row of
    a decode of its own 32-chunk, and follows the per-op step's arg max within the bf16 tolerance."""
    from asr_hip.decode import DecoderKVCache, FusedGreedyDecoder, fused_decode_supported, greedy_search_graphed
    model = _model(layers=2, inner=256, V=60)
    dec = model.decoder
    g = torch.Generator().manual_seed(9)
    B, Te, steps = 40, 600, 24
    enc = torch.randn(B, Te, 512, generator=g).cuda()
    assert fused_decode_supported(dec, enc, steps)
    toks = greedy_search_graphed(dec, enc, steps=steps)
    assert toks.shape[0] == B
    for lo in (0, 32):
        part = FusedGreedyDecoder(dec, enc[lo:lo + 32], max_len=steps)
        assert not part.fuse_cross                                   # more than 512 keys: query GEMM + multi-pass attention
        for _ in range(steps):
            part._step()
        want = part.out.t()
        n = min(toks.shape[1], steps)
        assert torch.equal(toks[lo:lo + 32, :n], want[:, :n])
    slow = DecoderKVCache(dec, enc, max_len=steps)
    prev = torch.full((B,), 1, dtype=torch.int64, device="cuda")
    for t in range(min(toks.shape[1], steps)):
        lg = slow.step(prev).float()
        got = lg.gather(1, toks[:, t:t + 1]).squeeze(1)
        assert ((lg.max(1).values - got) <= 3e-2 * lg.abs().max()).all(), t
        prev = toks[:, t].contiguous()
