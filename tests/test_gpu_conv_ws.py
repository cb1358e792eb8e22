"""The weight-stationary convolution kernel of csrc/conv_ws.hip (round 5: conv.5 forward in one pass, conv.7 forward with its pooled
epilogue, conv.7's data gradient with conv.5's ReLU mask, conv.5's data gradient -- reference models/asr/transformer.py:48-52,74-76 and
their autograd) through the C ABI, bit for bit.

Exact-integer data: inputs in {-3 .. 3}, weights in {-2 .. 2}, biases multiples of 0.5 -- every partial sum is a multiple of 0.5 below
2^23, so ANY summation order gives the same fp32 value, the bf16 rounding of the result is the same single rounding everywhere, and the
comparison is equality:
  * against torch's float64 convolution (the reference's nn.Conv2d + ReLU (+ mask) (+ MaxPool2d + view / transpose) on the same data);
  * against the kernels it replaces (tuning WS128 / WS64 = 0: the generic implicit GEMM, the two-pass c64 form), incl. the pooled form's
    selection bytes and both forms of the pooled epilogue (vertical tile pairs at H % 16 == 0, single tiles otherwise).
The same comparisons run without torch in tools/conv_ws_test.cpp (the development harness)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _data(B, H, W, Cin, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(-3, 4, (B, H, W, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, 3, 3), generator=g).float()
    bias = torch.randint(-8, 9, (Cout,), generator=g).float() * 0.5
    mask = torch.randint(-1, 2, (B, H, W, Cout), generator=g).float()
    return x, w, bias, mask


def _packed(w):
    """(Cout, 9 taps, Cin) bf16: the layout asr_conv_pack_weight produces (tap = ky * 3 + kx)."""
    Cout, Cin = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().cuda().bfloat16()


def _reference(x, w, bias, relu, mask):
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if relu:
        y = y.clamp_min(0)
    if mask is not None:
        y = y * (mask > 0)
    return y.float().bfloat16()          # exact fp32 value -> ONE round-to-nearest-even, as the kernels' v_cvt_pk_bf16_f32


CASES = [  # B, H, W, Cin, Cout, relu, mask
    (1, 8, 16, 128, 128, True, False), (2, 19, 37, 128, 128, False, True), (2, 24, 48, 128, 128, True, False),
    (2, 21, 50, 128, 64, False, True), (1, 16, 32, 128, 64, False, False),
    (1, 8, 16, 64, 128, True, False), (2, 21, 50, 64, 128, True, False), (2, 24, 48, 64, 128, False, False),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,relu,use_mask", CASES)
def test_weight_stationary_conv_is_exact(B, H, W, Cin, Cout, relu, use_mask):
    from asr_hip import lib as L
    from asr_hip import ops
    x, w, bias, mask = _data(B, H, W, Cin, Cout, seed=B * 1000 + H * 10 + Cout)
    xd, wk, bd = x.cuda().bfloat16(), _packed(w), bias.cuda()
    md = mask.cuda().bfloat16() if use_mask else None
    knob = "WS64" if Cin == 64 else "WS128"
    try:
        L.set_tuning(knob, 1)
        y_new = ops.conv3x3(xd, wk, bd, Cout, relu=relu, mask_src=md)
        L.set_tuning(knob, 0)
        y_old = ops.conv3x3(xd, wk, bd, Cout, relu=relu, mask_src=md)
    finally:
        L.set_tuning(knob, None)
    torch.cuda.synchronize()
    ref = _reference(x, w, bias, relu, mask if use_mask else None)
    assert torch.equal(y_new.cpu(), ref), "weight-stationary kernel vs float64 convolution"
    assert torch.equal(y_new, y_old), "weight-stationary kernel vs the kernel it replaces"


@pytest.mark.parametrize("B,H,W", [(3, 32, 64), (2, 24, 48), (1, 80, 400)])       # H % 16 == 0: vertical tile pairs; 24: single tiles
def test_pooled_epilogue_is_exact(B, H, W):
    """conv.7 + ReLU + MaxPool2d(2, 2) + view / transpose to (B, T', C F') from the convolution's epilogue, and the selection byte of every
    pooled element (0: the maximum is 0; 1 + k: first maximum at window position k in scan order, as torch's max_pool2d picks it)."""
    from asr_hip import lib as L
    from asr_hip import ops
    Cin = Cout = 128
    x, w, bias, _ = _data(B, H, W, Cin, Cout, seed=7 * H + W)
    xd, wk, bd = x.cuda().bfloat16(), _packed(w), bias.cuda()
    got = {}
    try:
        for ws in (1, 0):
            L.set_tuning("WS128", ws)
            got[ws] = ops.conv3x3_relu_pool_tcf_code(xd, wk, bd, Cout)
        L.set_tuning("WS128", 1)
        L.set_tuning("WS_PAIR", 0)
        got["single"] = ops.conv3x3_relu_pool_tcf_code(xd, wk, bd, Cout)
    finally:
        L.set_tuning("WS128", None)
        L.set_tuning("WS_PAIR", None)
    torch.cuda.synchronize()
    assert got[1] is not None and got["single"] is not None
    y = _reference(x, w, bias, True, None).float()                              # (B, H, W, C), exact bf16 values
    win = torch.stack([y[:, 0::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 0::2], y[:, 1::2, 1::2]], dim=-1)     # scan order of a 2x2 window
    m = win.max(dim=-1).values
    first = (win == m.unsqueeze(-1)).float().argmax(dim=-1)                     # the FIRST maximum of the window, explicitly
    code = torch.where(m > 0, first + 1, torch.zeros_like(first)).to(torch.uint8)
    want_pool = m.permute(0, 2, 3, 1).reshape(B, W // 2, Cout * (H // 2)).bfloat16()          # (B, W/2, C, H/2) flattened
    want_code = code.permute(0, 2, 3, 1).reshape(B, W // 2, Cout * (H // 2))
    for tag in (1, "single"):
        pool, cd = got[tag]
        assert torch.equal(pool.cpu(), want_pool), tag
        assert torch.equal(cd.cpu(), want_code), tag
    if got[0] is not None:                                                     # the generic kernel's pooled form needs H % 16 == 0
        assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


def test_many_tiles_per_workgroup_and_both_kernels_agree():
    """Nine images of the benchmark's 80 x 400: 2250 - 4500 tiles on 256 - 512 persistent workgroups (several rounds, the XCD walk, the
    tail) for every form, against the kernels they replace."""
    from asr_hip import lib as L
    from asr_hip import ops
    B, H, W = 9, 80, 400
    for Cin, Cout, relu, use_mask in ((128, 128, False, True), (128, 64, False, False), (64, 128, True, False)):
        x, w, bias, mask = _data(B, H, W, Cin, Cout, seed=Cin + Cout)
        xd, wk, bd = x.cuda().bfloat16(), _packed(w), bias.cuda()
        md = mask.cuda().bfloat16() if use_mask else None
        knob = "WS64" if Cin == 64 else "WS128"
        try:
            L.set_tuning(knob, 1)
            a = ops.conv3x3(xd, wk, bd, Cout, relu=relu, mask_src=md)
            L.set_tuning(knob, 0)
            b = ops.conv3x3(xd, wk, bd, Cout, relu=relu, mask_src=md)
        finally:
            L.set_tuning(knob, None)
        assert torch.equal(a, b), (Cin, Cout)


@pytest.mark.parametrize("B,H,W", [(1, 8, 16), (2, 21, 50), (2, 24, 48), (3, 80, 400)])
def test_relu_mask_as_bits_is_exact(B, H, W):
    """conv.5's ReLU mask for conv.7's data gradient at ONE BIT per element (asr_conv3x3_igemm_bits; reference: the autograd of
    transformer.py:50's ReLU): conv.5's epilogue writes the bits beside its output, the gradient kernel applies them.  The output, the
    bits (against their definition in include/asr_hip.h) and the masked gradient are those of the 16-bit-mask forms, bit for bit."""
    from asr_hip import lib as L
    from asr_hip import ops
    x, w5, b5, _ = _data(B, H, W, 64, 128, seed=H + W)
    g, w7, _, _ = _data(B, H, W, 128, 128, seed=H + W + 1)
    xd, wk5, bd, gd, wk7 = x.cuda().bfloat16(), _packed(w5), b5.cuda(), g.cuda().bfloat16(), _packed(w7)
    y_ref = ops.conv3x3(xd, wk5, bd, 128, relu=True)
    got = ops.conv3x3_relu_bits(xd, wk5, bd, 128)
    assert got is not None
    y, bits = got
    H4, W16 = 2 * ((H + 7) // 8), (W + 15) // 16
    assert bits.numel() == L.load().asr_relu_bits_bytes(B, H, W, 128) == B * H4 * W16 * 4 * 64 * 4
    assert torch.equal(y, y_ref)
    assert torch.equal(y.cpu(), _reference(x, w5, b5, True, None))
    # the layout of include/asr_hip.h: dword [b][h / 4][w / 16][c / 32][lane], byte = h % 4, bit = c % 8
    lane = torch.arange(64)
    l, grp = lane & 15, lane >> 4
    a = l >> 2
    pix = 8 * (a & 1) + 2 * (l & 3) + (((a >> 1) ^ a) & 1)
    chunk = 2 * (grp & 1) + (grp >> 1)                                     # 8-channel chunk of the wave's 32 channels
    by = bits.cpu().view(B, H4, W16, 4, 64, 4)                         # (..., wave, lane, row)
    unpacked = torch.zeros(B, H4 * 4, W16 * 16, 128, dtype=torch.bool)
    cols = torch.arange(W16).view(-1, 1) * 16 + pix.view(1, -1)        # (W16, 64)
    for wave in range(4):
        for k in range(8):
            ch = wave * 32 + chunk * 8 + k                                                                  # (64,)
            v = ((by[:, :, :, wave] >> k) & 1).bool().permute(0, 1, 4, 2, 3).reshape(B, H4 * 4, W16, 64)     # (B, rows, W16, lane)
            unpacked[:, :, cols, ch.view(1, -1).expand(W16, 64)] = v
    assert torch.equal(unpacked[:, :H, :W], y_ref.cpu().float() > 0)
    z_ref = ops.conv3x3(gd, wk7, None, 128, relu=False, mask_src=y_ref)
    z = ops.conv3x3_masked_by_bits(gd, wk7, None, 128, bits)
    assert z is not None and torch.equal(z, z_ref)
    assert torch.equal(z.cpu(), _reference(g, w7, torch.zeros(128), False, y_ref.cpu().float()))


def test_relu_bits_outside_their_domain_are_refused():
    from asr_hip import ops
    x = torch.zeros(1, 8, 16, 128, device="cuda", dtype=torch.bfloat16)
    wk = torch.zeros(128, 9, 128, device="cuda", dtype=torch.bfloat16)
    assert ops.conv3x3_relu_bits(x, wk, torch.zeros(128, device="cuda"), 128) is None            # bits out: 64 -> 128 only
    x32 = torch.zeros(1, 8, 16, 64, device="cuda")
    assert ops.conv3x3_relu_bits(x32, torch.zeros(128, 9, 64, device="cuda"), torch.zeros(128, device="cuda"), 128) is None      # fp32
