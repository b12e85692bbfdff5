import sys, torch, time
sys.path.insert(0,'/root/repo')
from torchrl_amd.contrib.controllers import PPOController
dev='cuda:0'
for obs,A,hid in ((8,4,(256,)),(15,4,(256,)),(12,3,(256,)),(8,4,(128,))):
    c=PPOController(obs,A,device=dev,hidden=hid)
    print(obs,A,hid,'fused',c._fused is not None,'epoch',c._epoch is not None)
    B=1<<20
    o=torch.randn(B,obs,device=dev); a=torch.randn(B,A,device=dev); r=torch.randn(B,1,device=dev); l=torch.randn(B,A,device=dev)-1; ad=torch.randn(B,1,device=dev)
    try:
        for _ in range(3): info=c.learn(o,a,None,None,None,r,l,ad)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(10): info=c.learn(o,a,None,None,None,r,l,ad)
        torch.cuda.synchronize(); print('  ms per 1M-row learn', (time.perf_counter()-t)/10*1e3, {k:float(v) for k,v in info.items()})
    except Exception as e:
        print('  error', str(e)[:200])
