"""Model / optimiser factories and checkpoint I/O with the reference's signatures and checkpoint schema
(reference: utils/functions.py).  Under --parallel the reference wraps the model in a single-process nn.DataParallel;
here every rank of a torch.distributed job owns one GPU and gradients are all-reduced over RCCL (asr_hip/ddp.py).
"""
import math
import os

import logging

import torch
import torch.distributed as dist

from asr_hip import ops
from asr_hip import params as P
from asr_hip.ddp import HipDataParallel
from models.asr.transformer import Decoder, Encoder, Transformer
from utils import constant
from utils.optimizer import AnnealingOpt, FusedAdam, NoamOpt


def _unwrap(model):
    return model.module if isinstance(model, HipDataParallel) else model


def save_model(model, epoch, opt, metrics, label2id, id2label, best_model=False):
    """Same file names and dict keys as the reference (functions.py:11-59).  Only rank 0 writes."""
    if dist.is_initialized() and dist.get_rank() != 0:
        return
    folder = os.path.join(constant.args.save_folder, constant.args.name)
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "best_model.th" if best_model else "epoch_{}.th".format(epoch))
    print("SAVE MODEL to", path)
    ckpt = {
        'label2id': label2id, 'id2label': id2label, 'args': constant.args, 'epoch': epoch,
        'model_state_dict': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        'optimizer_state_dict': opt.optimizer.state_dict(),
        'optimizer_params': {'_step': opt._step, '_rate': opt._rate, 'warmup': opt.warmup, 'factor': opt.factor,
                             'model_size': opt.model_size},
        'metrics': metrics,
    }
    torch.save(ckpt, path)


def load_model(load_path):
    """-> (model, opt, epoch, metrics, args, label2id, id2label)   (reference: functions.py:62-98)"""
    ckpt = torch.load(load_path, map_location="cpu", weights_only=False)
    args = ckpt.get('args', constant.args)
    cur = constant.args
    if args is not cur:
        # What describes THIS run, not the run that wrote the file, comes from the current command line: how the job is
        # laid out over GPUs (the reference re-wraps according to the current --parallel, train.py:92-99; a checkpoint saved
        # without it must not silently train un-synchronised replicas), where it runs, and the MI355X-path switches a
        # reference-written checkpoint does not carry at all.
        for k in ("parallel", "device_ids", "dist_backend", "bucket_mb", "grad_wire"):
            if hasattr(cur, k):
                setattr(args, k, getattr(cur, k))
        # numerics switches: what the checkpoint was trained with stays, unless the user typed the option on THIS command line or
        # the (reference-written) checkpoint does not carry it -- an fp32-trained model must not silently resume in bf16
        for k in ("precision", "gpu_frontend"):
            if hasattr(cur, k) and (k in getattr(constant, "explicit", ()) or not hasattr(args, k)):
                if hasattr(args, k) and getattr(args, k) != getattr(cur, k):
                    logging.info("load_model: --%s %s from the command line overrides the checkpoint's %s", k.replace("_", "-"),
                                 getattr(cur, k), getattr(args, k))
                setattr(args, k, getattr(cur, k))
        args.cuda = bool(getattr(args, "cuda", False) or getattr(cur, "cuda", False))
    label2id, id2label = ckpt['label2id'], ckpt['id2label']
    model = init_transformer_model(args, label2id, id2label)
    sd = ckpt['model_state_dict']
    wrapped = isinstance(model, HipDataParallel)
    has_prefix = any(k.startswith("module.") for k in sd)
    if has_prefix and not wrapped:
        sd = {k[len("module."):]: v for k, v in sd.items()}
    elif wrapped and not has_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    model.load_state_dict(sd)
    if getattr(args, "cuda", False):
        model = model.cuda()
    opt = init_optimizer(args, model)
    if getattr(args, "parallel", False) and dist.is_initialized() and dist.get_world_size() > 1:
        assert opt.optimizer.reducer is not None, "--parallel with world_size > 1 needs the gradient reducer"
    if opt is not None:
        opt.optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        op = ckpt['optimizer_params']
        opt._step, opt._rate, opt.warmup = op['_step'], op['_rate'], op['warmup']
        opt.factor, opt.model_size = op['factor'], op['model_size']
    return model, opt, ckpt['epoch'], ckpt['metrics'], args, label2id, id2label


def init_optimizer(args, model, opt_type="noam"):
    """Noam(model_size = args.dim_input AFTER init_transformer_model mutated it) over Adam(0.9, 0.98, 1e-9)
    (reference: functions.py:101-114).  Parameters are moved into one flat fp32 buffer; under --parallel a GradReducer
    is attached so that backward all-reduces gradient buckets as they complete."""
    if opt_type == "noam":
        core = _unwrap(model)
        bucket = int(getattr(args, "bucket_mb", 32.0) * (1 << 20)) if getattr(args, "parallel", False) else None
        adam = FusedAdam(list(core.parameters()), betas=(0.9, 0.98), eps=1e-9, ddp_bucket_bytes=bucket,
                         ddp_wire=getattr(args, "grad_wire", "fp32"))
        return NoamOpt(args.dim_input, args.k_lr, args.warmup, adam, min_lr=args.min_lr)
    if opt_type == "sgd":
        return AnnealingOpt(args.lr, args.lr_anneal, torch.optim.SGD(model.parameters(), lr=args.lr, momentum=args.momentum,
                                                                     nesterov=True))
    print("Optimizer is not defined")
    return None


def init_transformer_model(args, label2id, id2label):
    """Builds Encoder / Decoder / Transformer from the flags; mutates args.dim_input exactly like the reference
    (functions.py:116-162): 5120 for vgg_cnn, 672 for emb_cnn, unchanged (161) without a CNN."""
    n_fft_bins = int(math.floor((args.sample_rate * args.window_size) / 2) + 1)       # 161
    if args.feat_extractor == 'emb_cnn':
        h = int(math.floor(n_fft_bins - 41) / 2 + 1)
        h = int(math.floor(h - 21) / 2 + 1)
        args.dim_input = h * 32
    elif args.feat_extractor == 'vgg_cnn':
        args.dim_input = int(math.floor(int(math.floor(n_fft_bins) / 2) / 2)) * 128
    else:
        print("the model is initialized without feature extractor")
    ops.set_compute_dtype(torch.float32 if getattr(args, "precision", "bf16") == "fp32" else torch.bfloat16)
    ops.set_fp8(getattr(args, "precision", "bf16") == "fp8")
    encoder = Encoder(args.num_layers, num_heads=args.num_heads, dim_model=args.dim_model, dim_key=args.dim_key,
                      dim_value=args.dim_value, dim_input=args.dim_input, dim_inner=args.dim_inner,
                      src_max_length=args.src_max_len, dropout=args.dropout, rank=getattr(args, "rank", 0))
    decoder = Decoder(id2label, num_src_vocab=len(label2id), num_trg_vocab=len(label2id), num_layers=args.num_layers,
                      num_heads=args.num_heads, dim_emb=args.dim_emb, dim_model=args.dim_model, dim_inner=args.dim_inner,
                      dim_key=args.dim_key, dim_value=args.dim_value, trg_max_length=args.tgt_max_len, dropout=args.dropout,
                      emb_trg_sharing=args.emb_trg_sharing, rank=getattr(args, "rank", 0))
    model = Transformer(encoder, decoder, feat_extractor=args.feat_extractor)
    if args.parallel:
        model = HipDataParallel(model, device_ids=args.device_ids)
    return model
