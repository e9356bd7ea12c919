import os, sys, torch
sys.path.insert(0, '/root/repo')
from simple3d_former_amd import _lib as L, ops
B=64; Bb,H,hd=15,4,192; N,D=B*196,H*hd
g=torch.Generator().manual_seed(6)
qkv=(torch.randn(Bb*N,3*D,generator=g)*0.5).cuda()
hi,lo=ops.split_bf16(qkv); del qkv
seed=torch.tensor([4321],dtype=torch.int64,device='cuda')
T=(N+31)//32
mbuf=torch.zeros(Bb*H*T*T*32+256,dtype=torch.int32,device='cuda')
def timed(fn,reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for name,kw in (('dropout, mask stored',dict(drop=(0.1,seed,0),drop_mask=mbuf)),('dropout, no mask store',dict(drop=(0.1,seed,0))),('no dropout',dict())):
    for flag in (0,1):
        t=timed(lambda: ops.attention_fwd(hi,lo,Bb,H,N,D,1,Bb,split=True,p_single_plane=flag,**kw))
        print(f'{name}, p_single_plane {flag}: {t:.3f} ms',flush=True)
