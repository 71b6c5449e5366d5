import os, sys, numpy as np, torch
ROOT="/root/repo" if os.path.isdir("/root/repo/trex-gym_amd") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding
dev=torch.device("cuda:0"); n=4096
m=_capi.Model(); b=_capi.Batch(m,n)
obs=torch.zeros(n,75,device=dev); rew=torch.zeros(n,device=dev); done=torch.zeros(n,dtype=torch.uint8,device=dev)
b.reset(obs); ids=torch.arange(n,device=dev)
pool=[sharding.synthetic_actions(ids,t,m.lower,m.upper,device=dev) for t in range(16)]
cnt=torch.zeros(n,dtype=torch.int32,device=dev)
lo=torch.tensor(m.lower,dtype=torch.float32,device=dev); hi=torch.tensor(m.upper,dtype=torch.float32,device=dev)
for t in range(231):
    b.step(pool[t%16],obs,rew,done)
    if t in (30,60,100,150,230):
        b.contact_stats(cnt,None); c=cnt.cpu().numpy()
        q=obs[:,:25]; atlim=((q<=lo)|(q>=hi)).sum(1).cpu().numpy()
        st=torch.zeros(n,63,device=dev); b.get_state(st); z=st[:,2].cpu().numpy()
        print("step",t,"contacts hist",np.bincount(c,minlength=17),"envs with a joint on a stop: %.2f"%((atlim>0).mean()),"mean #stops %.1f"%atlim.mean(),"base z mean %.2f min %.2f"%(z.mean(),z.min()))
