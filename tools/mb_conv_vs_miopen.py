"""The 3x3 convolutions of vgg_cnn at the benchmark shapes: this library (implicit GEMM, fused bias + ReLU) against torch's conv2d
(MIOpen), bf16 channels-last, forward only.  Development tool.  python tools/mb_conv_vs_miopen.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "end2end-asr-pytorch_amd"))
from asr_hip import ops  # noqa: E402

D = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ops.set_compute_dtype(torch.bfloat16)
    torch.backends.cudnn.benchmark = True
    print("%-28s | ours (us, TF/s) | torch conv2d / MIOpen" % "B H W Cin Cout")
    for B, H, W, Ci, Co in [(32, 161, 800, 64, 64), (32, 80, 400, 64, 128), (32, 80, 400, 128, 128)]:
        x = torch.randn(B, H, W, Ci, device=D).bfloat16()
        wk = torch.randn(Co, 9, Ci, device=D).bfloat16()
        bias = torch.randn(Co, device=D)
        t = timeit(lambda: ops.conv3x3(x, wk, bias, Co, relu=True))
        xc = x.permute(0, 3, 1, 2)                                  # NCHW view of NHWC storage = channels_last
        wc = torch.randn(Co, Ci, 3, 3, device=D).bfloat16().contiguous(memory_format=torch.channels_last)
        bc = bias.bfloat16()
        try:
            tm = timeit(lambda: F.conv2d(xc, wc, bc, padding=1))
        except Exception as e:          # MIOpen may have no solver for a shape in this image
            tm = float("nan")
            print("  torch conv2d failed:", repr(e)[:120])
        fl = 2 * 9 * Ci * Co * B * H * W
        print("%-28s | %7.1f %6.0f | %7.1f %6.0f" % ((B, H, W, Ci, Co), t, fl / t / 1e6, tm, fl / tm / 1e6))


if __name__ == "__main__":
    main()
