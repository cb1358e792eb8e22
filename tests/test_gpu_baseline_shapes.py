"""Model-level parity on the MI355X AT THE SHAPES THE BENCHMARK RUNS: the product model against

  (1) the CPU oracle run on the same box with the same weights and inputs -- every logit, the loss, the arg-max rows,
      EVERY parameter gradient, the loss after one Noam/Adam step;
  (2) the summaries the executed reference left in tests/golden/{cfg0,cfg1_b2,cfg3_shape}.npz.

  cfg0        BASELINE configs[0] exactly: 2-layer d256 h4 dk=64 vgg_cnn, B=4, T=800 -> T'=200, Td=100, V=4364
  cfg1_b2     configs[1] (the benched 4-layer d512 h8 dk=64 model) at B=2
  cfg3_shape  configs[3]-shaped: emb_cnn, T=1600 -> T'=795 (ragged), d512 h8 dk=64, V=32, 2 encoder / 1 decoder layers
  cfg1_b32    configs[1] AS BENCHED: batch 32, ragged lengths (round 3: the kernel variants that are chosen only at this size -- the
              1700-tile data-gradient path, tn128p weight gradients, 128-row one-launch tiles -- against the oracle, not only in op tests)

  cfg3_b16    configs[3] AS BENCHED (round 4): 12 encoder / 6 decoder layers, emb_cnn, B=16, T=1600 ragged (five utterances shorter than
              T'=795 on the encoder axis): M = 12 720 rows -- per-slice grouped weight gradients, 256 x 256 blocks, the emb_cnn packet
              contractions at full height, the long-sequence attention kernels -- against the fp64 oracle, not only in op tests

dk=64 + bf16 runs attention_fast.hip, the 128x64 / tn128 GEMM tiles and the V=4364 -> 4416 padded vocabulary GEMM
that the tiny goldens never reach.  Ragged lengths: source rows below T' and targets from 5 to 99 tokens.

The truth is the oracle in FLOAT64 (same weights, same inputs, run on the GPU box's host cores).  Every fixture also holds,
per parameter, what the executed reference's OWN arithmetic is worth against its fp64 self at this shape
(oracle/gen_golden.py run_big): e32/<name> = relative L2 error of the reference's fp32 gradient, ebf/<name> = the same for
the reference under torch.autocast(cpu, bfloat16) -- PyTorch's own mixed precision.  The product's bounds are stated as
multiples of those floors:

  fp32 mode: logits atol 5e-5*max(1,max|logit|); loss 2e-5; per-tensor gradient relative L2 error <= max(2e-4, 4*e32);
             loss after the step 1e-4; arg-max exact on every row whose reference margin exceeds 1e-3.
             The gradient truth is evaluated UNDER THE PRODUCT'S OWN DISCRETE SELECTIONS (its ReLU masks and max-pool
             arg-maxes, tapped with asr_hip.functions.capture_selections and imposed on the fp64 oracle): at a pre-activation
             within fp32 rounding of 0, or a pooling window whose two largest values are within rounding of each other
             (exactly tied over the zero-padded frames), either selection is a correct fp32 result, but ONE flipped ReLU
             among the 4e5 hidden units of a layer moves that layer's gradient by 1e-3 relative, and the fp64 oracle's own
             selections over exactly tied windows depend on last-bit differences of its BLAS (measured: tools/diag_fp32.py,
             DESIGN.md section 2).  That the imposed selections are legitimate is checked by the oracle's FORWARD under them:
             its logits must equal the free-running fp64 logits to 1e-6.  The error against the free-running fp64 gradient
             is reported next to it (parity json: grad_rel_l2_free).
             Also asserted: the error against the FREE-RUNNING fp64 gradient (the oracle's own selections) <= 5e-3 per tensor.
  bf16 mode: logits atol 4e-2*max|logit|; loss 2e-2; per-tensor gradient relative L2 error <= max(REL_BF16, 2*ebf) (see BF16_FLOOR_FACTOR);
             arg-max: checked on EVERY row -- a row may differ from the oracle's arg-max only if the oracle's top-2 margin on that
             row is <= 2 x the measured max logit error of this run (north_star: "token-index argmax bit-exact"; a tie within the
             arithmetic's own error is the only admissible difference); the number of such rows is reported.
Measured values (every tensor) are written to gpurun_out/parity_r06.json and quoted in DESIGN.md section 2.
"""
import json
import os

import numpy as np
import pytest
import torch

import big_cases as BC

pytestmark = pytest.mark.gpu

REL_BF16 = 5e-2        # floor of the per-tensor bound ||g - g_64|| / ||g_64|| in bf16 mode (the bound is max(this, BF16_FLOOR_FACTOR * ebf[name]))
# ebf[name] is ONE sample of PyTorch-autocast's error on that tensor and the product's error is another sample of the same kind of noise.
# The factor is DERIVED, once (round 6, VERDICT r5 #6b), from a measurement of how far two such samples lie apart on the reference itself
# (oracle/bf16_floor_study.py -> profiles/r06_bf16_floor_study.json: the reference under torch.autocast(cpu, bf16) against its fp64 self,
# three summation orders, cfg0 / cfg1_b2 / cfg3_shape, 984 tensor x pair ratios): median 1.02, 90 % below 1.17, the largest ratio on an
# ordinary weight tensor 1.52 (an encoder key projection; the input LayerNorm's two parameters move 3 - 30 x when the intra-op thread
# count changes -- another reduction kernel inside PyTorch -- and are the whole tail above 1.75).  So the SAME implementation, re-ordered,
# already needs 1.52; another implementation of the same arithmetic gets the next quarter step of headroom on top: 2.0.  It is not to be
# moved when a test fails: a tensor above 2 x its floor is a defect to be explained.  (History: rounds 3 - 4 used 1.5 and round 5 met a
# decoder query bias at 1.55 x -- inside what the reference does to itself.)  The product's own ratios are histogrammed in the parity json.
BF16_FLOOR_FACTOR = 2.0
_oracle_cache = {}
_report = {}


def _oracle(name, z, model, src, src_len, tgt):
    from oracle import asr_oracle as O
    if name not in _oracle_cache:
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        w = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        cfg = BC.oracle_cfg(z)
        names = O.trainable_names(w, cfg)
        opt = O.NoamAdam({k: w[k] for k in names}, model_size=int(z["dim_input"]))
        bn = {}
        w64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in w.items()}
        r64 = O.train_step(w64, cfg, src.double(), src_len, tgt, float(z["smoothing"]), bn_state={})      # the truth
        r1 = O.train_step(w, cfg, src, src_len, tgt, float(z["smoothing"]), opt=opt, bn_state=bn)
        r2 = O.train_step(w, cfg, src, src_len, tgt, float(z["smoothing"]), bn_state=bn)
        r1["grads64"] = {k: v.double() for k, v in r64["grads"].items()}
        r1["pred64"] = r64["pred"].double()
        _oracle_cache[name] = (r1, r2["loss"], opt.rate)
    return _oracle_cache[name]


def _oracle_under_selections(z, model, taps, src, src_len, tgt, ref):
    """fp64 oracle gradients with the product's ReLU masks / pooling arg-maxes imposed (see the module docstring).
    Returns (grads, max |logit shift| of the oracle's forward caused by imposing them)."""
    import torch.nn.functional as F
    from oracle import asr_oracle as O
    core = model.module if hasattr(model, "module") else model
    prefixes = (["encoder.layers.%d.pos_ffn." % i for i in range(len(core.encoder.layers))] +
                ["decoder.layers.%d.pos_ffn." % i for i in range(len(core.decoder.layers))])
    ffn = [t for kind, t in taps if kind == "ffn"]
    assert len(ffn) == len(prefixes), (len(ffn), len(prefixes))
    dec = {}
    B = src.shape[0]
    for p, h in zip(prefixes, ffn):
        dec[p + "relu"] = (h.detach().cpu() > 0).view(B, -1, h.shape[-1])
    vgg = [t for kind, t in taps if kind == "vgg"]
    if vgg:
        nchw = lambda t: t.detach().cpu().permute(0, 3, 1, 2).contiguous()
        y1, y2, y3, y4 = (nchw(t) for t in vgg[0])
        dec["conv"] = {"relu0": y1 > 0, "relu2": y2 > 0, "relu5": y3 > 0, "relu7": y4 > 0,
                       "pool4": F.max_pool2d(y2.double(), 2, 2, return_indices=True)[1],
                       "pool9": F.max_pool2d(y4.double(), 2, 2, return_indices=True)[1]}
    w = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    w64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in w.items()}
    r = O.train_step(w64, BC.oracle_cfg(z), src.double(), src_len, tgt, float(z["smoothing"]), bn_state={}, decisions=dec)
    shift = float((r["pred"].double() - ref["pred64"]).abs().max())
    return {k: v.double() for k, v in r["grads"].items()}, shift


def _dump():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    # the product's rel / floor ratios over every (case, tensor) of the bf16 runs so far, as a histogram (same bin edges as the
    # reference-against-itself study in profiles/r06_bf16_floor_study.json)
    ratios = [r / max(fl, 1e-30) for key, e in _report.items() if key.endswith("/bf16") and "grad_rel_l2" in e for r, fl in e["grad_rel_l2"].values()]
    if ratios:
        edges = [0.0, 0.5, 0.75, 1.0, 1.05, 1.1, 1.2, 1.3, 1.5, 1.75, 2.0, 2.5, 3.0, 1e9]
        ra = np.array(ratios)
        _report["bf16_rel_over_floor"] = {
            "n": int(ra.size), "max": float(ra.max()), "p50": float(np.quantile(ra, 0.5)), "p90": float(np.quantile(ra, 0.9)), "p99": float(np.quantile(ra, 0.99)),
            "histogram": {"[%.2f, %s)" % (edges[i], ("%.2f" % edges[i + 1]) if edges[i + 1] < 1e8 else "inf"): int(((ra >= edges[i]) & (ra < edges[i + 1])).sum())
                          for i in range(len(edges) - 1)},
            "factor": BF16_FLOOR_FACTOR, "floor_absolute": REL_BF16,
            "note": "rel = ||g - g64|| / ||g64|| of the product's bf16 gradient per parameter tensor, floor = the same for the reference under "
                    "torch.autocast(cpu, bf16) (tests/golden/<case>.npz ebf/*); the per-tensor bound is max(floor_absolute, factor x floor)"}
    with open(os.path.join(out, "parity_r06.json"), "w") as f:
        json.dump(_report, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", BC.BIG_CASES)
def test_product_matches_oracle_and_reference_at_baseline_shape(golden_dir, name, precision):
    from utils.functions import init_optimizer
    from utils.metrics import calculate_metrics
    z = BC.load(golden_dir, name)
    args, model, l2i, i2l = BC.build_product(z, precision, True)
    src, src_len, tgt = BC.batch(z)
    model = model.cuda().train()
    ref, ref_loss2, ref_lr = _oracle(name, z, model, src, src_len, tgt)
    opt = init_optimizer(args, model, "noam")
    sm = float(z["smoothing"])
    srcd, tgtd = src.cuda(), tgt.cuda()

    def step():
        opt.zero_grad()
        pred, gold, hyp, _ = model(srcd, src_len, tgtd)
        loss, ncorrect = calculate_metrics(pred, gold, smoothing=sm, loss_type="ce")
        loss.backward()
        return pred, gold, hyp, loss, ncorrect

    from asr_hip import functions as F_
    F_.capture_selections = [] if precision == "fp32" else None
    pred, gold, hyp, loss, ncorrect = step()
    torch.cuda.synchronize()
    taps, F_.capture_selections = F_.capture_selections, None
    p = pred.detach().float().cpu()
    assert torch.isfinite(p).all()
    amax = float(ref["pred"].abs().max())
    assert abs(amax - float(z["pred_absmax"])) < 1e-4
    perr = float((p - ref["pred"]).abs().max())
    lerr = abs(loss.item() - ref["loss"])
    assert torch.equal(gold.cpu(), ref["gold"]) and np.array_equal(gold.cpu().numpy(), z["gold"])
    emb = name.startswith("cfg3")
    grads = {k: q.grad.detach().float().cpu() for k, q in model.named_parameters()}
    rel_free = {k: BC.rel_l2(grads[k].numpy(), ref["grads64"][k].numpy()) for k in grads if not BC.noise_driven(k, emb)}
    truth, sel_logit_dev = ref["grads64"], 0.0
    if precision == "fp32":
        truth, sel_logit_dev = _oracle_under_selections(z, model, taps, src, src_len, tgt, ref)
    rel = {k: BC.rel_l2(grads[k].numpy(), truth[k].numpy()) for k in grads if not BC.noise_driven(k, emb)}
    floor = {k: float(z[("e32/" if precision == "fp32" else "ebf/") + k]) for k in rel}
    bound = {k: max(2e-4, 4 * floor[k]) if precision == "fp32" else max(REL_BF16, BF16_FLOOR_FACTOR * floor[k]) for k in rel}
    worst = max(rel, key=lambda k: rel[k] / bound[k])
    perr64 = float((p.double() - ref["pred64"]).abs().max())
    summ = BC.summary_errors(z, pred, loss.item(), grads)
    margin = 1e-3 if precision == "fp32" else 8e-2 * amax
    miss, nsure = BC.argmax_agreement(z, hyp, margin)
    rm = ref["pred"].topk(2, dim=2).values
    sure = (rm[..., 0] - rm[..., 1]) > margin
    miss_o = int((hyp.cpu()[sure] != ref["hyp"][sure]).sum())
    # every row: a difference from the oracle's arg-max is admissible only inside the run's own logit error
    diff_rows = hyp.cpu() != ref["hyp"]
    n_diff = int(diff_rows.sum())
    worst_diff_margin = float((rm[..., 0] - rm[..., 1])[diff_rows].max()) if n_diff else 0.0
    opt.step()
    _, _, _, loss2, _ = step()
    l2err = abs(loss2.item() - ref_loss2)
    _report["%s/%s" % (name, precision)] = {
        "logit_max_abs_err": perr, "logit_abs_max": amax, "loss_err": lerr, "loss2_err": l2err, "loss2_err_vs_reference": abs(loss2.item() - float(z["loss2"])),
        "grad_rel_l2_worst": rel[worst], "grad_rel_l2_worst_name": worst, "grad_rel_l2_worst_bound": bound[worst],
        "grad_rel_l2_median": float(np.median(list(rel.values()))), "grad_rel_l2_max": max(rel.values()),
        "reference_floor_median": float(np.median(list(floor.values()))), "reference_floor_max": max(floor.values()),
        "logit_max_abs_err_vs_fp64": perr64,
        "reference_logit_err": float(z["pred_err_f32" if precision == "fp32" else "pred_err_autocast_bf16"]),
        "grad_rel_l2": {k: [rel[k], floor[k]] for k in sorted(rel, key=lambda k: -rel[k] / bound[k])},
        "grad_rel_l2_free_max": max(rel_free.values()), "grad_rel_l2_free_median": float(np.median(list(rel_free.values()))),
        "grad_rel_l2_free": {k: rel_free[k] for k in sorted(rel_free, key=lambda k: -rel_free[k])[:12]},
        "oracle_logit_shift_under_product_selections": sel_logit_dev,
        "argmax_rows_checked": nsure, "argmax_mismatch_vs_reference": miss, "argmax_mismatch_vs_oracle": miss_o,
        "argmax_rows_total": int(diff_rows.numel()), "argmax_rows_differing": n_diff, "argmax_worst_margin_of_a_differing_row": worst_diff_margin,
        "ref_summary_pred_sub": summ["pred_sub"], "ref_summary_gs_worst": max(v for k, v in summ["gs"].items() if not BC.noise_driven(k, emb)),
        "num_correct": int(ncorrect), "lr1": opt._rate}
    _dump()
    assert abs(opt._rate - ref_lr) < 1e-12 and abs(opt._rate - float(z["lr1"])) < 1e-12
    assert int(ncorrect) == int(z["num_correct"]) or precision == "bf16"
    if precision == "fp32":
        assert sel_logit_dev <= 1e-6 * max(1.0, amax), sel_logit_dev
        assert perr <= 5e-5 * max(1.0, amax), perr
        assert lerr < 2e-5 and l2err < 1e-4, (lerr, l2err)
        assert rel[worst] <= bound[worst], (worst, rel[worst], bound[worst])
        assert max(rel_free.values()) <= 5e-3, max(rel_free.items(), key=lambda kv: kv[1])
        assert summ["pred_sub"] <= 1e-4 and abs(loss2.item() - float(z["loss2"])) < 1e-4
    else:
        assert perr <= 4e-2 * amax, (perr, amax)
        assert lerr < 2e-2 and l2err < 3e-2, (lerr, l2err)
        assert rel[worst] <= bound[worst], (worst, rel[worst], bound[worst])
    assert miss == 0 and miss_o == 0 and nsure > (20 if precision == "fp32" else 5), (miss, miss_o, nsure)
    assert worst_diff_margin <= 2 * perr + 1e-12, (n_diff, worst_diff_margin, perr)


def test_graph_replay_equals_eager_at_dk64_bf16(golden_dir):
    """The benched launch mode (whole-step hipGraph) against the eager launch sequence on configs[0] in bf16, dropout 0:
    same loss for four consecutive steps and the same weights afterwards."""
    from asr_hip.graph import GraphedTrainStep
    from utils.functions import init_optimizer
    from utils.metrics import calculate_loss
    z = BC.load(golden_dir, "cfg0")
    src, src_len, tgt = BC.batch(z)
    srcd, tgtd = src.cuda(), tgt.cuda()
    sm = float(z["smoothing"])
    args, m1, _, _ = BC.build_product(z, "bf16", True)
    m1 = m1.cuda().train()
    o1 = init_optimizer(args, m1, "noam")
    losses = []
    for _ in range(4):
        o1.zero_grad()
        pred, gold, _, _ = m1(srcd, src_len, tgtd)
        loss = calculate_loss(pred, gold, smoothing=sm)
        loss.backward()
        o1.step()
        losses.append(loss.item())
    args, m2, _, _ = BC.build_product(z, "bf16", True)
    m2 = m2.cuda().train()
    o2 = init_optimizer(args, m2, "noam")
    gs = GraphedTrainStep(m2, o2, sm, srcd, src_len, tgtd, warmup_steps=1)
    assert abs(gs.loss.item() - losses[1]) < 2e-3
    for k in (2, 3):
        l, _ = gs(srcd, src_len, tgtd)
        assert abs(l.item() - losses[k]) < 2e-3, (k, l.item(), losses[k])
    worst = 0.0
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if k.endswith("key_linear.bias") or k.endswith(".pe"):
            continue
        worst = max(worst, float((a - b).abs().max()))
    # 4 Adam steps at lr <= 4 * lr1: fp32 atomics order inside the split reductions is the only difference
    assert worst < 4 * 4 * float(z["lr1"]) + 1e-7, worst


def test_benched_conv_path_agrees_with_the_tapped_launch_chain(golden_dir):
    """VERDICT r4 #6b.  The fp32-tight evidence above runs the conv front end with the activation tap ON (the parity tests need the
    stored activations), i.e. on the launch chain conv1_fwd -> conv.2 + pool codes -> conv.5 -> conv.7 -> pooling kernel; the benched
    step runs the kernels that never store those tensors (conv_level0.hip, conv.7's pooled epilogue in conv_ws.hip).  This ties the two:
    configs[1] at its own batch, bf16, dropout 0, same weights, same batch -- logits, loss and EVERY parameter gradient of the two
    product runs must agree to bf16 rounding (the two chains round the same fp32 sums at the same places, in another summation order;
    a ReLU / pooling decision that flips inside that rounding moves a conv gradient by ~1e-3)."""
    from utils.functions import init_optimizer
    from utils.metrics import calculate_metrics
    from asr_hip import functions as F_
    z = BC.load(golden_dir, "cfg1_b32")
    src, src_len, tgt = BC.batch(z)
    srcd, tgtd = src.cuda(), tgt.cuda()
    sm = float(z["smoothing"])
    runs = {}
    for tap in (True, False):
        args, model, _, _ = BC.build_product(z, "bf16", True)
        model = model.cuda().train()
        opt = init_optimizer(args, model, "noam")
        F_.capture_selections = [] if tap else None
        try:
            opt.zero_grad()
            pred, gold, hyp, _ = model(srcd, src_len, tgtd)
            loss, _ = calculate_metrics(pred, gold, smoothing=sm, loss_type="ce")
            loss.backward()
            torch.cuda.synchronize()
        finally:
            taps, F_.capture_selections = F_.capture_selections, None
        if tap:
            assert any(kind == "vgg" for kind, _ in taps), "the tapped run is expected to go through the stored-activation chain"
        runs[tap] = (pred.detach().float().cpu(), float(loss.item()), {k: q.grad.detach().float().cpu() for k, q in model.named_parameters()})
    (p1, l1, g1), (p0, l0, g0) = runs[True], runs[False]
    amax = float(p1.abs().max())
    perr = float((p1 - p0).abs().max())
    rel = {k: BC.rel_l2(g0[k].numpy(), g1[k].numpy()) for k in g1 if not BC.noise_driven(k, False)}
    worst = max(rel, key=lambda k: rel[k])
    _report["cfg1_b32/bf16/tap_off_vs_tap_on"] = {"logit_max_abs_diff": perr, "logit_abs_max": amax, "loss_diff": abs(l1 - l0),
                                                   "grad_rel_l2_worst": rel[worst], "grad_rel_l2_worst_name": worst,
                                                   "grad_rel_l2_median": float(np.median(list(rel.values()))),
                                                   "grad_rel_l2_conv": {k: rel[k] for k in rel if k.startswith("conv.")},
                                                   "grad_rel_l2": {k: rel[k] for k in sorted(rel, key=lambda k: -rel[k])[:12]}}
    _dump()
    # Bound per tensor = twice the bound each run has against the fp64 truth (triangle inequality over the two runs).  Measured (profiles/r05_parity_baseline_shapes.json): logits
    # 1.7e-2 of 1.82, loss 4e-5, gradients median 2.5e-2, worst 8.9e-2 on a decoder self-attention query weight whose autocast floor is
    # 0.1 -- a last-bit difference in the conv features is amplified by every bf16 rounding behind it, exactly like a change of seed.
    floor = {k: float(z["ebf/" + k]) for k in rel}
    # either run is within max(REL_BF16, BF16_FLOOR_FACTOR x floor) of the truth, so two runs are within twice that of each other; the
    # median says how close they really are (measured 2.5e-2)
    bound = {k: 2 * max(REL_BF16, BF16_FLOOR_FACTOR * floor[k]) for k in rel}
    worst_b = max(rel, key=lambda k: rel[k] / bound[k])
    assert perr <= 1.5e-2 * amax, (perr, amax)
    assert abs(l1 - l0) < 2e-3, (l1, l0)
    assert rel[worst_b] <= bound[worst_b], (worst_b, rel[worst_b], bound[worst_b])
    assert float(np.median(list(rel.values()))) <= 4e-2
    # (the front end's own gradients differ by 2.1 - 3.5e-2: they inherit the difference of the gradient that ARRIVES from the encoder;
    #  the chains themselves are tied bit for bit on exact-integer inputs by tests/test_gpu_level0.py and tools/conv_ws_test.cpp)
