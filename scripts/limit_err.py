import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/trex-gym_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'trex-gym_amd')
from oracle import oracle as O, trex_model as tm
from trex_gym.vec_env import TrexVecEnv
m=tm.compile_model(O.default_asset_urdf())
lo=m["q_lower"][m["obs_order"]]
st=np.zeros((3,63),np.float32); st[:,2]=50; st[:,6]=1; st[:,13:38]=lo-0.05
v=TrexVecEnv(3, device="cuda:0"); v.reset(); v.set_state(torch.tensor(st))
a=np.tile(lo,(3,1)).astype(np.float32)
obs,_,_,_=v.step(a)
o64=O.Oracle(m,precision="f64"); o32=O.Oracle(m,precision="f32")
res={}
for name,orc in (("f64",o64),("f32",o32)):
    s=orc.new_state(); orc.set_state(s, st[0].astype(np.float64)); o,_,_=orc.step(s,a[0].astype(np.float64)); res[name]=o
scale=np.abs(res["f64"][25:50]).max()
print("qd scale", scale)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
print("gpu-f64 / scale:", (obs[0,25:50]-res["f64"][25:50])/scale)
print("f32-f64 / scale:", (res["f32"][25:50]-res["f64"][25:50])/scale)
print("names", [n.replace('joint_','') for n in m["obs_joint_names"]])
print("max |gpu-f64|/scale %.4f ; max |f32-f64|/scale %.4f"%(np.abs(obs[0,25:50]-res["f64"][25:50]).max()/scale, np.abs(res["f32"][25:50]-res["f64"][25:50]).max()/scale))
# sensitivity of the f64 oracle itself to a 1e-7 perturbation of the state
s=o64.new_state(); stp=st[0].astype(np.float64).copy(); stp[13:38]*= (1+1e-7); o64.set_state(s,stp); op,_,_=o64.step(s,a[0].astype(np.float64))
print("f64 oracle, q perturbed by 1e-7 relative: max |dqd|/scale %.2e"%(np.abs(op[25:50]-res["f64"][25:50]).max()/scale))
for it in (60, 200, 1000):
    oo=O.Oracle(m,precision="f64",params=dict(iterations=it)); s=oo.new_state(); oo.set_state(s, st[0].astype(np.float64)); o,_,_=oo.step(s,a[0].astype(np.float64))
    o3=O.Oracle(m,precision="f32",params=dict(iterations=it)); s=o3.new_state(); o3.set_state(s, st[0].astype(np.float64)); o2,_,_=o3.step(s,a[0].astype(np.float64))
    print("iterations %d: f32-f64 max %.4f of scale %.2f"%(it, np.abs(o2[25:50]-o[25:50]).max()/np.abs(o[25:50]).max(), np.abs(o[25:50]).max()))
