"""The captured whole-step hipGraph must be the same training step as the eager launch sequence: same losses, same
weights after N steps (dropout off), a learning rate computed on the device that equals Noam's host formula, and
different dropout masks on every replay (dropout on)."""
import numpy as np
import pytest
import torch

from test_gpu_model import build

pytestmark = pytest.mark.gpu


def _batch(z):
    return torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["src_len"]), torch.from_numpy(z["tgt"]).cuda()


def test_graph_replay_equals_eager(golden_dir):
    from asr_hip.graph import GraphedTrainStep
    from oracle import asr_oracle as O
    from utils.metrics import calculate_loss
    z, args, m1, o1 = build(golden_dir, "vgg_tiny", "fp32")
    src, src_len, tgt = _batch(z)
    sm = float(z["smoothing"])
    losses = []
    for _ in range(4):
        o1.zero_grad()
        pred, gold, _, _ = m1(src, src_len, tgt)
        loss = calculate_loss(pred, gold, smoothing=sm)
        loss.backward()
        o1.step()
        losses.append(loss.item())
    z, args, m2, o2 = build(golden_dir, "vgg_tiny", "fp32")
    gs = GraphedTrainStep(m2, o2, sm, src, src_len, tgt, warmup_steps=1)       # 1 eager + 1 replayed step
    assert o2._step == 2 and abs(gs.loss.item() - losses[1]) < 2e-5
    for k in (2, 3):
        loss, sums = gs(src, src_len, tgt)
        assert abs(loss.item() - losses[k]) < 5e-5, (k, loss.item(), losses[k])
        assert abs(gs.lr_dev.item() - O.noam_rate(k + 1, int(z["dim_input"]), 1.0, 4000, 1e-5)) < 1e-11
    assert o2._step == 4 and abs(o2._rate - o1._rate) < 1e-15
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if k.endswith("key_linear.bias"):
            continue
        assert torch.allclose(a, b, atol=1e-5), k
    # new data through the static buffers: a different batch gives a different loss
    src2 = src.flip(0).contiguous()
    l_a = gs(src2, src_len, tgt)[0].item()
    assert abs(l_a - losses[3]) > 1e-6


def test_graph_replay_changes_dropout_masks(golden_dir):
    """Weights frozen (lr = 0 through k_lr = 0 and min_lr = 0): the loss still changes between replays because the
    device-side seed counter advances."""
    from asr_hip.graph import GraphedTrainStep
    z, args, m, o = build(golden_dir, "vgg_tiny", "bf16")
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.3
    o.factor, o.min_lr = 0.0, 0.0
    src, src_len, tgt = _batch(z)
    gs = GraphedTrainStep(m, o, float(z["smoothing"]), src, src_len, tgt, warmup_steps=1)
    vals = [gs()[0].item() for _ in range(4)]
    assert len(set(round(v, 6) for v in vals)) >= 3, vals
    w = {k: v.clone() for k, v in m.state_dict().items()}
    gs()
    for k, v in m.state_dict().items():
        assert torch.equal(v, w[k]), k            # lr = 0: nothing moves
