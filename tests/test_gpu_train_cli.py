"""End-to-end through the reference's entry-point structure on the GPU: manifests + label file -> train.py's main()
(Trainer.train, validation, checkpointing) -> test.py's evaluate() (greedy decode, CER/WER).  Tiny synthetic corpus."""
import json
import os
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _corpus(tmp_path, n=6):
    rng = np.random.RandomState(0)
    words = ["ab", "ba", "abba", "bab", "aab", "bba"]
    lines = []
    for i in range(n):
        w = tmp_path / ("u%d.wav" % i)
        ns = 4000 + 800 * i
        with wave.open(str(w), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((rng.randn(ns) * 2000).astype("<i2").tobytes())
        t = tmp_path / ("u%d.txt" % i)
        t.write_text(words[i % len(words)] + "\n")
        lines.append("%s,%s" % (w, t))
    man = tmp_path / "train.csv"
    man.write_text("\n".join(lines))
    lab = tmp_path / "labels.json"
    lab.write_text(json.dumps([" ", "a", "b"]))
    return str(man), str(lab)


def test_train_and_test_entry_points(tmp_path, monkeypatch):
    from utils import constant
    man, lab = _corpus(tmp_path)
    monkeypatch.chdir(tmp_path)
    argv = ["--train-manifest-list", man, "--valid-manifest-list", man, "--test-manifest-list", man, "--labels-path", lab,
            "--cuda", "--batch-size", "3", "--num-workers", "0", "--epochs", "2", "--save-every", "1", "--name", "tiny",
            "--save-folder", str(tmp_path / "save"), "--num-layers", "1", "--num-heads", "2", "--dim-model", "32", "--dim-key",
            "16", "--dim-value", "16", "--dim-inner", "64", "--dim-emb", "32", "--tgt-max-len", "12", "--src-max-len", "64",
            "--label-smoothing", "0.1", "--dropout", "0.1", "--k-lr", "20", "--warmup", "5", "--clip", "--shuffle"]
    constant.parse(argv)
    import train as train_mod
    train_mod.main()
    ck = tmp_path / "save" / "tiny"
    assert (ck / "epoch_1.th").exists() and (ck / "best_model.th").exists()
    state = torch.load(str(ck / "best_model.th"), map_location="cpu", weights_only=False)
    assert set(state) >= {"label2id", "id2label", "args", "epoch", "model_state_dict", "optimizer_state_dict", "optimizer_params",
                          "metrics"}
    assert state["optimizer_params"]["_step"] >= 2 and np.isfinite(state["metrics"]["train_loss"])
    hist = state["metrics"]["history"]
    # evaluation entry point on the saved checkpoint (greedy decode runs its 300 steps -> tgt-max-len >= 301)
    constant.parse(argv + ["--continue-from", str(ck / "best_model.th"), "--tgt-max-len", "301"])
    from utils.data_loader import AudioDataLoader, BucketingSampler, SpectrogramDataset
    from utils.functions import load_model
    import test as test_mod
    model, opt, epoch, metrics, largs, l2i, i2l = load_model(str(ck / "best_model.th"))
    largs.tgt_max_len = 301
    conf = dict(sample_rate=16000, window_size=.02, window_stride=.01, window="hamming", noise_dir=None, noise_prob=0.4,
                noise_levels=(0.0, 0.5))
    ds = SpectrogramDataset(conf, [man], l2i, normalize=True)
    loader = AudioDataLoader(ds, num_workers=0, batch_sampler=BucketingSampler(ds, batch_size=3))
    # positional-encoding buffer of the decoder must cover 301 positions for decoding
    from models.common_layers import PositionalEncoding
    model.decoder.positional_encoding = PositionalEncoding(model.decoder.dim_model, 301).cuda()
    cer, wer = test_mod.evaluate(model, loader)
    assert np.isfinite(cer) and np.isfinite(wer) and cer >= 0


def test_gpu_frontend_batches_equal_host_batches_and_train(tmp_path, monkeypatch):
    """--gpu-frontend: the loader ships padded waveforms; utils.audio.gpu_front_end must rebuild exactly the batch the host
    path produces (features, frame counts, --src-max-len cut after normalisation), and a training epoch must run on it."""
    from utils import constant
    man, lab = _corpus(tmp_path)
    monkeypatch.chdir(tmp_path)
    base = ["--train-manifest-list", man, "--valid-manifest-list", man, "--test-manifest-list", man, "--labels-path", lab,
            "--cuda", "--batch-size", "3", "--num-workers", "0", "--epochs", "1", "--save-every", "1", "--name", "tinyg",
            "--save-folder", str(tmp_path / "save"), "--num-layers", "1", "--num-heads", "2", "--dim-model", "32", "--dim-key",
            "16", "--dim-value", "16", "--dim-inner", "64", "--dim-emb", "32", "--tgt-max-len", "12", "--src-max-len", "40",
            "--label-smoothing", "0.1", "--dropout", "0.0", "--k-lr", "20", "--warmup", "5"]
    from utils.audio import gpu_front_end
    from utils.data_loader import AudioDataLoader, BucketingSampler, SpectrogramDataset
    conf = dict(sample_rate=16000, window_size=.02, window_stride=.01, window="hamming", noise_dir=None, noise_prob=0.4,
                noise_levels=(0.0, 0.5))
    l2i = {c: i for i, c in enumerate(["\xa0", "<", ">", " ", "a", "b"])}
    batches = {}
    for mode in ("host", "gpu"):
        args = constant.parse(base + (["--gpu-frontend"] if mode == "gpu" else []))
        ds = SpectrogramDataset(conf, [man], l2i, normalize=True)
        np.random.seed(0)
        loader = AudioDataLoader(ds, num_workers=0, batch_sampler=BucketingSampler(ds, batch_size=3))
        out = []
        for src, tgt, _, sizes, _ in loader:
            if mode == "gpu":
                src, sizes = gpu_front_end(src.cuda(), sizes, src_max_len=args.src_max_len)
                src = src.cpu()
            out.append((src, tgt, sizes))
        batches[mode] = out
    assert len(batches["host"]) == len(batches["gpu"]) == 2
    def canon(b):       # the collate order is by length; the --src-max-len cut creates ties on the host path only
        src, tgt, sizes = b
        order = sorted(range(tgt.shape[0]), key=lambda i: tuple(tgt[i].tolist()))
        return src[order], tgt[order], torch.as_tensor(sizes)[order]
    for hb, gb in zip(batches["host"], batches["gpu"]):
        (hs, ht, hz), (gs, gt, gz) = canon(hb), canon(gb)
        assert hs.shape == gs.shape and torch.equal(ht, gt) and torch.equal(hz.int(), gz.int())
        assert gs.shape[-1] <= 40                      # 4000..8000 samples -> 26..51 frames, cut to --src-max-len
        np.testing.assert_allclose(gs.numpy(), hs.numpy(), rtol=0, atol=5e-4)
    assert max(g[0].shape[-1] for g in batches["gpu"]) == 40
    constant.parse(base + ["--gpu-frontend"])
    import train as train_mod
    train_mod.main()
    assert (tmp_path / "save" / "tinyg" / "best_model.th").exists()
