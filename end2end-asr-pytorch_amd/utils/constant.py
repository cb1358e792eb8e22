"""Command-line flags and special tokens of the training / test entry points.

Flag names, dests and defaults follow the reference interface (reference: utils/constant.py:6-94) so that existing
command lines keep working; `args` is the same process-global Namespace the reference exposes (constant.py:99).

Differences that do not change old command lines:
  * sys.argv is parsed only when the entry script is train.py / test.py (the reference parses at import time from ANY
    script, which makes its modules unusable from tests); otherwise `args` holds the defaults and can be replaced with
    `set_args(ns)` / `parse(argv)`;
  * `--precision {bf16,fp32}` selects the kernels' storage type (bf16 = perf mode, fp32 = parity mode);
  * `--dist-backend` / `--bucket-mb` / `--grad-wire` tune the RCCL data-parallel path that replaces nn.DataParallel under --parallel;
  * `--gpu-frontend`: the loader ships padded waveforms and the log-spectrogram is computed on the GPU (asr_stft_frames +
    fp32 MFMA DFT + asr_spect_finish) instead of on the host in the DataLoader workers.
"""
import argparse
import os
import sys

import torch

# (flags, kwargs) -- one row per option of the reference CLI, grouped as in its --help
_S, _I, _F = str, int, float
_FLAGS = [
    # experiment
    (("--model",), dict(default="TRFS", type=_S)), (("--name",), dict(default="model")),
    # data
    (("--train-manifest-list",), dict(nargs="+", type=_S)), (("--valid-manifest-list",), dict(nargs="+", type=_S)),
    (("--test-manifest-list",), dict(nargs="+", type=_S)), (("--lang-list",), dict(nargs="+", type=_S)),
    (("--sample-rate",), dict(default=16000, type=_I)), (("--batch-size",), dict(default=20, type=_I)),
    (("--num-workers",), dict(default=4, type=_I)), (("--labels-path",), dict(default="labels.json")),
    (("--label-smoothing",), dict(default=0.0, type=_F)),
    (("--window-size",), dict(default=0.02, type=_F)), (("--window-stride",), dict(default=0.01, type=_F)),
    (("--window",), dict(default="hamming")),
    # run control
    (("--epochs",), dict(default=1000, type=_I)), (("--cuda",), dict(dest="cuda", action="store_true")),
    (("--device-ids",), dict(default=None, nargs="+", type=_I)), (("--lr", "--learning-rate"), dict(default=3e-4, type=_F)),
    (("--save-every",), dict(default=5, type=_I)), (("--save-folder",), dict(default="models/")),
    (("--emb_trg_sharing",), dict(action="store_true")), (("--feat_extractor",), dict(default="vgg_cnn", type=_S)),
    (("--verbose",), dict(action="store_true")), (("--continue-from",), dict(default="")),
    # augmentation
    (("--augment",), dict(dest="augment", action="store_true")), (("--noise-dir",), dict(default=None)),
    (("--noise-prob",), dict(default=0.4)), (("--noise-min",), dict(default=0.0, type=_F)),
    (("--noise-max",), dict(default=0.5, type=_F)),
    # model
    (("--num-layers",), dict(default=3, type=_I)), (("--num-heads",), dict(default=5, type=_I)),
    (("--dim-model",), dict(default=512, type=_I)), (("--dim-key",), dict(default=64, type=_I)),
    (("--dim-value",), dict(default=64, type=_I)), (("--dim-input",), dict(default=161, type=_I)),
    (("--dim-inner",), dict(default=1024, type=_I)), (("--dim-emb",), dict(default=512, type=_I)),
    (("--src-max-len",), dict(default=4000, type=_I)), (("--tgt-max-len",), dict(default=1000, type=_I)),
    # optimiser
    (("--warmup",), dict(default=4000, type=_I)), (("--min-lr",), dict(default=1e-5, type=_F)),
    (("--k-lr",), dict(default=1, type=_F)), (("--momentum",), dict(default=0.9, type=_F)),
    (("--lr-anneal",), dict(default=1.1, type=_F)),
    # decoding
    (("--beam-search",), dict(action="store_true")), (("--beam-width",), dict(default=3, type=_I)),
    (("--beam-nbest",), dict(default=5, type=_I)), (("--lm-rescoring",), dict(action="store_true")),
    (("--lm-path",), dict(type=_S, default="lm_model.pt")), (("--lm-weight",), dict(default=0.1, type=_F)),
    (("--c-weight",), dict(default=0.1, type=_F)), (("--prob-weight",), dict(default=1.0, type=_F)),
    # loss / regularisation
    (("--loss",), dict(type=_S, default="ce")), (("--clip",), dict(action="store_true")),
    (("--max-norm",), dict(default=400, type=_F)), (("--dropout",), dict(default=0.1, type=_F)),
    (("--parallel",), dict(action="store_true")), (("--shuffle",), dict(action="store_true")),
    # MI355X path (additions)
    # fp8: bf16 storage, fp8 (e4m3) MFMA for the forward --rank projections -- functional, SLOWER than bf16 (22.97 vs 12.55 ms per step
    # of configs[4]: quantisation passes around K = 64 contractions); parity unpinned (no reference code for the low-rank variant)
    (("--precision",), dict(default="bf16", choices=["bf16", "fp32", "fp8"],
                            help="bf16 (default) | fp32 (parity mode) | fp8 (low-rank projections only; functional, slower than bf16)")),
    (("--dist-backend",), dict(default="nccl")), (("--bucket-mb",), dict(default=32.0, type=_F)),
    (("--grad-wire",), dict(default="fp32", choices=["fp32", "bf16"])),
    (("--gpu-frontend",), dict(action="store_true")),
    # 0: eager launches on the batch exactly as collated.  N > 0: training batches are zero-padded along time to a multiple of N
    # frames (the way the collate function pads shorter utterances) and to --tgt-max-len - 1 target columns, and the step is a
    # captured hipGraph per (batch, frames) shape, replayed (trainer/asr/trainer.py)
    (("--graph-buckets",), dict(default=0, type=_I,
                                help="0: eager launches; N > 0: hipGraph replay per (batch, frames padded to a multiple of N) shape. Default: 64 for a "
                                     "vgg_cnn model with --cuda and the CE loss (decided from the MODEL, after --continue-from is resolved), else 0. "
                                     "Parity caveat: the longest utterance of a batch sees zero frames instead of the image border behind its "
                                     "last frame (as every shorter one already does through the collate padding)")),
    # data parallel under graph replay: `four` = four hipGraphs with the RCCL all-reduces between them (ordering proven under two ranks);
    # `one` = the collectives captured inside ONE hipGraph; `auto` = try one, verify its first replayed step against the four-graph
    # step's loss and gradient checksum on the same batch, fall back LOUDLY (asr_hip/graph.py)
    (("--ddp-graph",), dict(default="auto", choices=["one", "four", "auto"])),
    # Low-Rank Transformer (arXiv:1910.13923, BASELINE configs[4]): rank of every attention / feed-forward projection, 0 = full rank
    (("--rank",), dict(default=0, type=_I)),
]

parser = argparse.ArgumentParser(description="Transformer ASR on MI355X")
for _names, _kw in _FLAGS:
    parser.add_argument(*_names, **_kw)

# same global seeding as the reference (constant.py:96-97) so `--continue-from`-less runs start from the same init
torch.manual_seed(123456)
if torch.cuda.is_available():
    torch.cuda.manual_seed_all(123456)


def _given(argv):
    """Destinations of the options that appear on this command line (as opposed to argparse defaults)."""
    out = set()
    for act in parser._actions:
        if any(tok == o or tok.startswith(o + "=") for o in act.option_strings for tok in argv):
            out.add(act.dest)
    return out


def parse(argv):
    """Parse an explicit argv (list of strings) into the process-global Namespace."""
    global args, USE_CUDA, explicit
    args = parser.parse_args(argv)
    explicit = _given(list(argv))
    USE_CUDA = args.cuda
    return args


def set_args(ns):
    global args, USE_CUDA
    args = ns
    USE_CUDA = getattr(ns, "cuda", False)
    return args


_entry = os.path.basename(sys.argv[0]) if sys.argv and sys.argv[0] else ""
args = parser.parse_args(sys.argv[1:] if _entry in ("train.py", "test.py") else [])
explicit = _given(sys.argv[1:] if _entry in ("train.py", "test.py") else [])      # options the user actually typed (load_model)
USE_CUDA = args.cuda

PAD_TOKEN, SOS_TOKEN, EOS_TOKEN = 0, 1, 2
PAD_CHAR, SOS_CHAR, EOS_CHAR = "\u00b6", "\u00a7", "\u00a4"
