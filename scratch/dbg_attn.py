import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import gpu_ops as G
from test_ops_gpu import _attn_ref
B,h,Tq,Tk,d,causal = 3,4,5,5,16,False
Hd=h*d
g=torch.Generator().manual_seed(1)
qkv=torch.randn(B*Tq,3*Hd,generator=g)
Q,K,V=qkv[:,:Hd],qkv[:,Hd:2*Hd],qkv[:,2*Hd:]
for variant in range(3):
    ids=torch.randint(1,50,(B,Tk),generator=g,dtype=torch.int32)
    if variant>=1: ids[0,Tk-2:]=0
    if variant>=2: ids[1,:]=0
    ref=_attn_ref(Q.clone(),K.clone(),V.clone(),ids,B,h,Tq,Tk,d,causal,d**-0.5)
    qd=qkv.cuda()
    O=G.attn_fwd(qd[:,:Hd],qd[:,Hd:2*Hd],qd[:,2*Hd:],ids.cuda(),B,h,Tq,Tk,d,causal,d**-0.5)
    diff=(O.cpu()-ref).abs().view(B,Tq,Hd).amax(dim=(1,2))
    print("variant",variant,"per-batch max diff",diff.tolist())
