"""CTC loss parity (reference: utils/metrics.py:133-154).  The reference's CTC path IS
F.log_softmax + torch's F.ctc_loss(reduction="mean", blank=0): the truth here is exactly that call on the CPU (float64),
loss and d(loss)/d(logits), on ragged input / target lengths, repeated labels (the s-2 transition rule), an empty target,
the benchmark's vocabulary (V = 4364, T = 100) and an unreachable target (loss = +inf, which the reference's trainer skips)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, T, V, Lmax, in_len, tg_len, seed, repeat=False):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g) * 2.0
    tgt = torch.randint(1, V, (B, Lmax), generator=g)
    if repeat:
        tgt[:, 1::2] = tgt[:, 0::2][:, :tgt[:, 1::2].shape[1]]          # "aabbcc": every second label repeats its neighbour
    return logits, tgt, torch.tensor(in_len, dtype=torch.int32), torch.tensor(tg_len, dtype=torch.int32)


def _truth(logits, tgt, il, tl, dtype=torch.float64):
    x = logits.to(dtype).clone().requires_grad_()
    lp = F.log_softmax(x.transpose(0, 1), dim=2)
    loss = F.ctc_loss(lp, tgt, il.long(), tl.long(), reduction="mean")
    if torch.isfinite(loss):
        loss.backward()
    return loss.detach(), x.grad


@pytest.mark.parametrize("case", [
    dict(B=3, T=12, V=7, Lmax=4, in_len=[12, 9, 5], tg_len=[4, 2, 1], seed=1),
    dict(B=4, T=30, V=35, Lmax=10, in_len=[30, 30, 21, 12], tg_len=[10, 6, 0, 3], seed=2, repeat=True),
    dict(B=2, T=100, V=4364, Lmax=46, in_len=[100, 73], tg_len=[46, 20], seed=3),
    dict(B=2, T=300, V=32, Lmax=140, in_len=[300, 290], tg_len=[140, 97], seed=4, repeat=True),
])
def test_ctc_loss_and_gradient_match_torch(case):
    from utils.metrics import calculate_loss, calculate_metrics
    logits, tgt, il, tl = _case(**case)
    loss_ref, grad_ref = _truth(logits, tgt, il, tl)
    x = logits.cuda().requires_grad_()
    loss = calculate_loss(x, tgt.cuda(), input_lengths=il, target_lengths=tl, loss_type="ctc")
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    loss.backward()
    err = (x.grad.double().cpu() - grad_ref).abs().max().item()
    # the lattice holds log-probabilities of magnitude ~ 5 T in fp32: the bound is a small multiple of what torch's own fp32
    # ctc_loss loses against float64 on the same tensors
    _, grad32 = _truth(logits, tgt, il, tl, torch.float32)
    err32 = (grad32.double() - grad_ref).abs().max().item()
    assert err <= max(2e-6 + 2e-5 * grad_ref.abs().max().item(), 4 * err32), (err, err32)
    l2, none = calculate_metrics(logits.cuda(), tgt.cuda(), input_lengths=il.cuda(), target_lengths=tl.cuda(), loss_type="ctc")
    assert none is None and abs(l2.item() - loss_ref.item()) <= 2e-5 * max(1.0, abs(loss_ref.item()))


def test_ctc_unreachable_target_is_infinite():
    """5 frames cannot emit "aa..." of length 4 with its mandatory blanks: torch returns +inf (zero_infinity=False); the
    reference's trainer then skips the batch (trainer.py:87-90)."""
    from utils.metrics import calculate_loss
    logits, tgt, il, tl = _case(2, 8, 9, 4, [8, 5], [2, 4], seed=7, repeat=True)
    loss_ref, _ = _truth(logits, tgt, il, tl)
    loss = calculate_loss(logits.cuda(), tgt.cuda(), input_lengths=il, target_lengths=tl, loss_type="ctc")
    assert torch.isinf(loss_ref) and torch.isinf(loss).item() and loss.item() > 0
