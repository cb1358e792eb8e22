# quick unit check of the emb_cnn pieces against torch on the GPU box
import sys, os, torch
sys.path.insert(0, "end2end-asr-pytorch_amd")
from asr_hip import ops
import torch.nn.functional as F
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    ops.set_compute_dtype(dt)
    B,H,W,C=2,61,53,32
    x = torch.randn(B,H,W,C,device="cuda").to(dt)
    g = ops.conv_geom(B,H,W,C,21,11,2,1,0,0)
    M=B*g[10]*g[11]; K=C*21*11; Mp=(M+127)//128*128; ld=ops._pad8(K)
    col = ops.im2col(x, g, torch.full((Mp,ld), 7.0, device="cuda", dtype=dt))
    ref = F.unfold(x.permute(0,3,1,2).float(), (21,11), stride=(2,1))   # (B, C*KH*KW, L) with (c,ky,kx)
    ref = ref.view(B,C,21*11,-1).permute(0,3,2,1).reshape(M,K)
    print(dt, "im2col", (col[:M,:K].float()-ref).abs().max().item(), col[M:].abs().max().item(), col[:, K:].abs().max().item())
    dcol = torch.randn(Mp, ld, device="cuda").to(dt)
    dx = ops.col2im(dcol, g)
    dref = F.fold(dcol[:M,:K].float().view(B,-1,21*11,C).permute(0,3,2,1).reshape(B,C*231,-1), (H,W),(21,11),stride=(2,1))
    print(dt, "col2im", (dx.float()-dref.permute(0,2,3,1)).abs().max().item(), dref.abs().max().item())
    # convA style: C=1, fp32 input
    xs = torch.randn(2,1,161,96,device="cuda")
    gA = ops.conv_geom(2,161,96,1,41,11,2,2,0,10)
    MA=2*gA[10]*gA[11]; MAp=(MA+127)//128*128
    colA = ops.im2col(xs.view(2,161,96,1), gA, torch.full((MAp,512),3.0,device="cuda",dtype=dt))
    refA = F.unfold(xs, (41,11), stride=(2,2), padding=(0,10)).permute(0,2,1).reshape(MA,451)
    print(dt, "im2colA", (colA[:MA,:451].float()-refA).abs().max().item(), colA[:,451:].abs().max().item())
