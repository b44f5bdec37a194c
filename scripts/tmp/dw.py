import torch, sys
sys.path.insert(0,'/root/repo')
from sm3det_amd import _lib_backbone as LB
call=LB.call
def t(fn,n=50):
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        g=torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
tot=0
for (B,H,W,C,cnt) in [(2,256,256,96,3),(2,128,128,192,3),(2,64,64,384,9),(2,32,32,768,3)]:
    T=B*H*W
    x=torch.randn(T,C,device='cuda'); w=torch.randn(49,C,device='cuda'); b=torch.randn(C,device='cuda'); y=torch.empty_like(x); add=torch.randn_like(x)
    a=t(lambda: call('dwconv7_fwd', x,w,b,None,y,B,H,W,C,0)); d=t(lambda: call('dwconv7_fwd', x,w,None,add,y,B,H,W,C,1))
    du=torch.randn_like(x); dw=torch.zeros(49,C,device='cuda'); db=torch.zeros(C,device='cuda')
    ww=t(lambda: call('dwconv7_bwd_weight', x,du,dw,db,B,H,W,C))
    print(f'{(B,H,W,C)} fwd {a:6.1f} us {8*T*C/a/1e6:5.2f} TB/s | dgrad+addend {d:6.1f} us {12*T*C/d/1e6:5.2f} TB/s | wgrad {ww:6.1f} us {8*T*C/ww/1e6:5.2f} TB/s')
    tot+=cnt*(a+d+ww)
print('per step total %.1f us'%tot)
