import os, sys, torch
sys.path.insert(0, "end2end-asr-pytorch_amd")
from asr_hip import ops
torch.manual_seed(0)
B,H,Tq,Tk,d = 1,1,16,64,64
D="cuda"
q=torch.zeros(B,Tq,H*d,device=D).bfloat16(); k=torch.zeros(B,Tk,H*d,device=D).bfloat16()
v=torch.eye(64,device=D).view(1,64,64).bfloat16().contiguous()
kp=torch.zeros(B,Tk,dtype=torch.uint8,device=D)
kp[0,[1,5,6,17,34,35,63]]=1
o1,l1,_=ops.attn_fwd(q,k,v,H,d,key_pad=kp,scale=0.125)
print("masked keys expected:", kp[0].nonzero().flatten().tolist())
for row in (0,3,15):
    print("row",row,"zero-prob keys (fast):", (o1[0,row].float()==0).nonzero().flatten().tolist(), "sum", o1[0,row].float().sum().item())
