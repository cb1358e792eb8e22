"""Training entry point -- same command line as the reference (reference: train.py).

  python train.py --train-manifest-list ... --valid-manifest-list ... --labels-path ... --cuda [--parallel --device-ids 0 1 ...]

--parallel: the reference wraps the model in a single-process nn.DataParallel.  Here one process per GPU is started
(unless the job was already launched by torch.distributed.run) and gradients are all-reduced over RCCL.
"""
import json
import logging
import os
import sys

import torch
import torch.distributed as dist

from utils import constant


def build_labels(path):
    with open(path) as f:
        chars = ''.join(json.load(f))
    labels = constant.PAD_CHAR + constant.SOS_CHAR + constant.EOS_CHAR + chars
    label2id, id2label = {}, {}
    for ch in labels:
        if ch in label2id:
            print("multiple label: ", ch)
            continue
        label2id[ch] = len(label2id)
        id2label[label2id[ch]] = ch
    return label2id, id2label


DEFAULT_GRAPH_BUCKET = 64


def resolve_graph_buckets(args, explicit, model=None):
    """The fast path is the default path: `train.py --cuda` with the vgg_cnn front end and the cross-entropy loss replays one captured
    hipGraph per (batch, frames padded to a multiple of 64) shape -- 5.75 against 7.86 ms per step of the eager loop at configs[1]
    (profiles/r04_trainer_rate.txt).  `--graph-buckets 0` opts out; an explicit value always wins.  Since round 6 also for emb_cnn: its
    BatchNorm statistics are length-masked on the device to the batch as collated (what the reference's BatchNorm sees), so the bucket
    padding no longer enters them (asr_bn_batch_stats_v; tests/test_gpu_graph.py).
    `model`: the front end is read from the model that will train (main() calls this AFTER init / load_model: a --continue-from
    checkpoint carries its own --feat_extractor, the command line's default must not decide for it).
    Parity caveat of the default (README, --help): for the LONGEST utterance of a batch the convolutions see bias + ReLU'd zero frames
    where --graph-buckets 0 (and the reference) see the image border -- 1e-3 of the loss on vgg_tiny padded 64 -> 96 frames
    (tests/test_gpu_graph.py); every other utterance already sees exactly that through the collate padding."""
    if "graph_buckets" in explicit:
        return args.graph_buckets
    core = model.module if hasattr(model, "module") else model
    feat = getattr(core, "feat_extractor", None) if core is not None else None
    if feat is None:
        feat = getattr(args, "feat_extractor", "")
    if getattr(args, "cuda", False) and feat in ("vgg_cnn", "emb_cnn") and getattr(args, "loss", "ce") == "ce":
        args.graph_buckets = DEFAULT_GRAPH_BUCKET
    return args.graph_buckets


def main():
    from trainer.asr.trainer import Trainer
    from utils.data_loader import AudioDataLoader, BucketingSampler, SpectrogramDataset
    from utils.functions import init_optimizer, init_transformer_model, load_model

    args = constant.args
    if "OMP_NUM_THREADS" not in os.environ:
        # the training process only issues kernel launches and small host tensor ops (collation runs in the loader's worker
        # processes): with torch's default of one intra-op thread per core, every host-side tensor op wakes a pool whose workers
        # then spin and slow the launch path of the next step (measured on a 128-thread box: 36 vs 10.7 ms per trainer step)
        torch.set_num_threads(min(8, os.cpu_count() or 1))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.parallel and world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        ids = args.device_ids or list(range(world))
        torch.cuda.set_device(ids[local % len(ids)])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend)
    rank0 = not dist.is_initialized() or dist.get_rank() == 0
    if rank0:
        print("=" * 50)
        print("THE EXPERIMENT LOG IS SAVED IN: " + "log/" + args.name)
        print("TRAINING MANIFEST: ", args.train_manifest_list)
        print("VALID MANIFEST: ", args.valid_manifest_list)
        print("TEST MANIFEST: ", args.test_manifest_list)
        print("=" * 50)
    os.makedirs("./log", exist_ok=True)
    logging.basicConfig(filename="log/" + args.name, filemode='w+', format='%(asctime)s - %(message)s', level=logging.INFO)
    audio_conf = dict(sample_rate=args.sample_rate, window_size=args.window_size, window_stride=args.window_stride,
                      window=args.window, noise_dir=args.noise_dir, noise_prob=args.noise_prob,
                      noise_levels=(args.noise_min, args.noise_max))
    logging.info(audio_conf)
    label2id, id2label = build_labels(args.labels_path)

    train_data = SpectrogramDataset(audio_conf, manifest_filepath_list=args.train_manifest_list, label2id=label2id,
                                    normalize=True, augment=args.augment)
    train_sampler = BucketingSampler(train_data, batch_size=args.batch_size)
    train_loader = AudioDataLoader(train_data, num_workers=args.num_workers, batch_sampler=train_sampler)
    valid_loader_list = []
    for m in args.valid_manifest_list or []:
        valid_data = SpectrogramDataset(audio_conf, manifest_filepath_list=[m], label2id=label2id, normalize=True, augment=False)
        valid_loader_list.append(AudioDataLoader(valid_data, num_workers=args.num_workers, batch_size=args.batch_size))

    start_epoch, metrics = 0, None
    if args.continue_from != "":
        logging.info("Continue from checkpoint: " + args.continue_from)
        model, opt, start_epoch, metrics, loaded_args, label2id, id2label = load_model(args.continue_from)
    elif args.model == "TRFS":
        model = init_transformer_model(args, label2id, id2label)
        opt = init_optimizer(args, model, "noam")
    else:
        raise SystemExit("The model is not supported, check args --h")
    if constant.USE_CUDA:
        model = model.cuda()
    resolve_graph_buckets(args, constant.explicit, model)          # after the model exists: its front end decides, not the command line's default
    logging.info("training step: %s", ("hipGraph replay per (batch, frames padded to a multiple of %d) shape (--graph-buckets 0: eager launches)"
                                       % args.graph_buckets) if args.graph_buckets > 0 else "eager launches (--graph-buckets N: captured hipGraphs)")
    logging.info(model)
    Trainer().train(model, train_loader, train_sampler, valid_loader_list, opt, args.loss, start_epoch, args.epochs, label2id,
                    id2label, metrics)
    if dist.is_initialized():
        dist.destroy_process_group()


def _spawn_ranks():
    """--parallel without an external launcher: re-exec this command line under torch.distributed.run, one rank per
    device id (the reference's --device-ids flag keeps its meaning)."""
    n = len(constant.args.device_ids) if constant.args.device_ids else torch.cuda.device_count()
    if n <= 1:
        return False
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29400 + os.getpid() % 500)] + sys.argv
    os.execv(sys.executable, cmd)


if __name__ == '__main__':
    if constant.args.parallel and "WORLD_SIZE" not in os.environ:
        _spawn_ranks()
    main()
