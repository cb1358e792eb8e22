#!/usr/bin/env python3
"""Throughput of the Transformer-ASR TRAINING STEP on MI355X (BASELINE.json metric: input spectrogram frames/s).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): 4-layer d_model=512 heads=8 dim-inner=2048 vgg_cnn Transformer,
synthetic spectrogram batch (B=32 per GPU, 1, 161, T_src=800) fp32, targets (B, 99) int64 padded to T_tgt=100,
V=4364, label smoothing 0.1, dropout 0.1, bf16 kernels with fp32 accumulation; a step = zero_grad + forward +
label-smoothed CE + backward (+ gradient all-reduce over RCCL when N > 1) + Noam/Adam update.  Inputs are resident in
HBM before the timed region.  One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "end2end-asr-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

V = 4364
T_SRC, T_TGT, F_BINS = 800, 100, 161
MODEL_FLAGS = ["--num-layers", "4", "--num-heads", "8", "--dim-model", "512", "--dim-key", "64", "--dim-value", "64",
               "--dim-inner", "2048", "--dim-emb", "512", "--feat_extractor", "vgg_cnn", "--tgt-max-len", str(T_TGT),
               "--src-max-len", str(T_SRC), "--label-smoothing", "0.1"]
MFLOP_PER_FRAME = 129.9          # fwd+bwd algorithmic FLOPs (2*MAC over conv/mm/bmm) per input frame, SURVEY.md 8(d)
# configs[3] (not the headline; `--workload librispeech`): enc 12 / dec 6 layers, emb_cnn, T_src=1600, B=16, V=32
LIBRI = {"T_SRC": 1600, "T_TGT": 100, "V": 32, "B": 16, "MFLOP_PER_FRAME": 179.0, "enc_layers": 12, "dec_layers": 6}
PEAK_BF16_TFLOPS = 2500.0        # dense MFMA bf16 peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def conv_igemm_flops(B):
    """Algorithmic FLOPs of all asr_conv3x3_igemm launches in ONE step: 3 forward convs + 3 dgrads (2*9*Cin*Cout/pixel)."""
    px1, px2 = B * 161 * 800, B * 80 * 400
    fwd = 2 * 9 * (64 * 64 * px1 + 64 * 128 * px2 + 128 * 128 * px2)
    return 2 * fwd, 6


def measured_traffic(a):
    """HBM bytes per igemm launch from the committed rocprofv3 PMC passes (profiles/r01_roofline_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE runs of this workload, gfx950 correction applied); None for any other batch / precision."""
    if a.batch != 32 or a.precision != "bf16":
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_roofline_traffic.json")) as f:
            return json.load(f)["traffic_bytes_per_launch_avg"]
    except (OSError, KeyError, ValueError):
        return None


def labels(vocab=V):
    from utils import constant
    chars = [constant.PAD_CHAR, constant.SOS_CHAR, constant.EOS_CHAR] + [chr(0x4E00 + i) for i in range(vocab - 3)]
    l2i = {c: i for i, c in enumerate(chars)}
    return l2i, {i: c for c, i in l2i.items()}


def synthetic_batch(B, torch, t_src=T_SRC, t_tgt=T_TGT, vocab=V):
    g = torch.Generator().manual_seed(1234)
    src = torch.randn(B, 1, F_BINS, t_src, generator=g)
    tgt = torch.randint(3, vocab, (B, t_tgt - 1), generator=g)
    src_len = torch.full((B,), t_src, dtype=torch.int32)
    return src, src_len, tgt


def build_librispeech_model(args, l2i, i2l):
    """configs[3]: 12 encoder / 6 decoder layers need the constructors (the CLI has one --num-layers for both,
    reference utils/functions.py:148-151)."""
    from models.asr.transformer import Decoder, Encoder, Transformer
    t_out = (LIBRI["T_SRC"] + 20 - 11) // 2 + 1 - 10
    enc = Encoder(LIBRI["enc_layers"], 8, 512, 64, 64, 32 * 21, 2048, dropout=args.dropout, src_max_length=max(t_out, 2500))
    dec = Decoder(i2l, len(l2i), len(l2i), LIBRI["dec_layers"], 8, 512, 512, 2048, 64, 64, dropout=args.dropout,
                  trg_max_length=args.tgt_max_len, emb_trg_sharing=False)
    return Transformer(enc, dec, feat_extractor="emb_cnn")


def cpu_baseline(state_dict, dropout_free_flags, seconds_budget=25.0):
    """The CPU oracle (port of the reference step, oracle/asr_oracle.py) timed on this box's host cores on a bounded
    sample of the same workload: same model, same T_src/T_tgt/V, batch 8 instead of 32."""
    import torch
    from oracle import asr_oracle as O
    Bc = 8
    cfg = O.Cfg.from_flags(dropout_free_flags)
    w = {k: v.detach().float().cpu() for k, v in state_dict.items()}
    src, src_len, tgt = synthetic_batch(Bc, torch)
    names = O.trainable_names(w, cfg)
    opt = O.NoamAdam({k: w[k] for k in names}, model_size=5120)
    times = []
    t_start = time.time()
    for i in range(3):
        t0 = time.time()
        O.train_step(w, cfg, src, src_len, tgt, 0.1, opt=opt)
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": Bc * T_SRC / best, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/asr_oracle.py train_step (fwd+CE+bwd+Noam/Adam, fp32, dropout off), same 4-layer d512 vgg_cnn "
                      "model and shapes at batch %d, best of %d timed steps after 1 warm-up; data loading and CER "
                      "bookkeeping excluded" % (Bc, max(1, len(times) - 1))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 32; 16 for --workload librispeech)")
    ap.add_argument("--workload", default="headline", choices=["headline", "librispeech"],
                    help="headline = BASELINE configs[1] (the judged metric); librispeech = configs[3] (12/6 layers, emb_cnn)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    a = ap.parse_args()
    libri = a.workload == "librispeech"
    if a.batch is None:
        a.batch = LIBRI["B"] if libri else 32
    t_src, t_tgt, vocab = (LIBRI["T_SRC"], LIBRI["T_TGT"], LIBRI["V"]) if libri else (T_SRC, T_TGT, V)
    mflop_per_frame = LIBRI["MFLOP_PER_FRAME"] if libri else MFLOP_PER_FRAME

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    force_ddp = os.environ.get("ASR_FORCE_DDP") == "1"
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(os.environ.get("ASR_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    from asr_hip import lib as L
    from asr_hip import ops
    from utils import constant
    from utils.functions import init_optimizer, init_transformer_model
    from utils.metrics import calculate_loss

    flags = MODEL_FLAGS + ["--dropout", str(a.dropout), "--precision", a.precision, "--cuda", "--batch-size", str(a.batch)]
    if libri:
        flags += ["--feat_extractor", "emb_cnn", "--src-max-len", str(t_src)]
    if world > 1 or force_ddp:
        flags.append("--parallel")
    args = constant.parse(flags)
    l2i, i2l = labels(vocab)
    torch.manual_seed(123456)
    if libri:
        args.dim_input = 32 * 21
        ops.set_compute_dtype(torch.float32 if a.precision == "fp32" else torch.bfloat16)
        model = build_librispeech_model(args, l2i, i2l).cuda()
        if world > 1 or force_ddp:
            from asr_hip.ddp import HipDataParallel
            model = HipDataParallel(model, device_ids=args.device_ids)
    else:
        model = init_transformer_model(args, l2i, i2l).cuda()
    model.train()
    opt = init_optimizer(args, model, "noam")
    src, src_len, tgt = synthetic_batch(a.batch, torch, t_src, t_tgt, vocab)
    src, tgt = src.cuda(), tgt.cuda()
    sd_cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not libri:
        core = model.module if hasattr(model, "module") else model
        sd_cpu = {k: v.detach().cpu().clone() for k, v in core.state_dict().items()}

    def eager_step():
        opt.zero_grad()
        pred, gold, hyp, _ = model(src, src_len, tgt)
        loss = calculate_loss(pred, gold, smoothing=0.1, loss_type="ce")
        loss.backward()
        opt.step()
        return loss

    # N = 1: the whole step is one captured hipGraph (same kernels, no per-launch host cost).  N > 1 stays eager: RCCL
    # collectives inside a capture could not be validated on the single-GPU development box.
    use_graph = not a.eager and (world == 1 or os.environ.get("ASR_GRAPH_DDP") == "1")
    if use_graph:
        from asr_hip.graph import GraphedTrainStep
        gs = GraphedTrainStep(model, opt, 0.1, src, src_len, tgt, warmup_steps=max(1, a.warmup))
        step = lambda: gs()[0]
    else:
        step = eager_step
        for _ in range(a.warmup):
            step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss.item())

    # roofline pass: the dominant kernel family bracketed by HIP events on its own stream, over `prof_steps` eager steps of
    # the same workload (events cannot be recorded inside a graph replay).  Every rank takes part: the steps contain the
    # gradient all-reduce.
    prof = None
    if not a.no_roofline and not libri:
        prof_steps = min(a.steps, 3)
        ops.prof_enable(L.OP_CONV_IGEMM, True)
        for _ in range(prof_steps):
            eager_step()
        torch.cuda.synchronize()
        tot_ms, n = ops.prof_collect(L.OP_CONV_IGEMM)
        ops.prof_enable(L.OP_CONV_IGEMM, False)
        prof = (tot_ms, n, prof_steps)
    if world > 1:
        dist.barrier()

    out = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        frames = a.batch * world * t_src * a.steps
        value = frames / dt
        peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS
        out = {"metric": "input spectrogram frames/sec (training step)", "value": value, "unit": "frames/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
               "launch_mode": "hipGraph replay" if use_graph else "eager",
               "config": {"workload": ("configs[3]: 12-enc/6-dec-layer d_model=512 heads=8 dim-inner=2048 emb_cnn Transformer ASR "
                                       "training step, synthetic (B=%d/GPU,1,161,T_src=1600) -> T_tgt=100, V=32, label "
                                       "smoothing 0.1, dropout %.2f, random init" if libri else
                                       "configs[1]: 4-layer d_model=512 heads=8 dim-inner=2048 vgg_cnn Transformer ASR training "
                                       "step, synthetic (B=%d/GPU,1,161,T_src=800) -> T_tgt=100, V=4364, label smoothing 0.1, "
                                       "dropout %.2f, random init") % (a.batch, a.dropout),
                          "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                          "step_tflops_whole_model": value * mflop_per_frame * 1e6 / 1e12,
                          "frac_of_mfma_peak_whole_step": value * mflop_per_frame * 1e6 / 1e12 / (peak * world),
                          "final_loss": final_loss}}
        if prof is not None:
            tot_ms, n, prof_steps = prof
            fl, per_step = conv_igemm_flops(a.batch)
            if n > 0 and tot_ms > 0:
                ach = fl * prof_steps / (tot_ms * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "kernel": "conv3x3_igemm_kernel (3 fwd + 3 dgrad launches per step)",
                                   "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                   "traffic": measured_traffic(a),
                                   "launches": n, "avg_launch_ms": tot_ms / n,
                                   "algorithmic_flop_per_launch_avg": fl / per_step}
        if sd_cpu is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(sd_cpu, " ".join(MODEL_FLAGS))
            except Exception as e:           # the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
