import torch,time,numpy as np,sys
sys.path.insert(0,".")
from sgam_neurips22_amd.tsdf import TsdfVolume, frustum_bounds
from sgam_neurips22_amd.inference_pipeline import synthetic_seed_frame, intrinsics
K=intrinsics("google_earth",(256,256)); T=np.eye(4)
lo,hi=frustum_bounds(K,[T],256,256,4.8,0.2)
v=TsdfVolume(0.01,0.03,lo,hi,"cuda",memory_budget_bytes=4<<30)
d=torch.from_numpy(synthetic_seed_frame("google_earth",0,256)[1]).cuda()
v.integrate(d,K,T)
o=v.render_depth(K,T,256,256,0.05,4.8).cpu().numpy()
c=(o//10000); f=o%10000
print("coarse steps mean/max",c.mean(),c.max(),"fine mean/max",f.mean(),f.max(), "near bricks", int(v.brick_near.sum()), v.stats(), "depth range", float(d.min()), float(d.max()))
