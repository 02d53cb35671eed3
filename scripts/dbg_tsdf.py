import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle.tsdf import TsdfOracle
from test_tsdf_cpu import _K, _pose, plane_depth
from test_gpu_tsdf import _scene, _bricks
H=W=64; K=_K(120.0,31.5)
for poses in ([_pose()], [_pose(), _pose(tx=0.21,yaw=0.07)]):
    vol, ora = _scene(0.05,0.5,H,W,K,poses, lambda T: plane_depth(K,T,H,W,8.0))
    got=_bricks(vol)
    nbad=0; tot=0; mx=0
    for key,(t,w) in got.items():
        o=ora.units[key][0]
        d=(t.view(np.int32).astype(np.int64)-o.view(np.int32).astype(np.int64))
        nbad+=(d!=0).sum(); tot+=d.size; mx=max(mx,np.abs(d).max())
        if (d!=0).any() and nbad<5:
            idx=np.argwhere(d!=0)[0]; print(key, idx, t[tuple(idx)], o[tuple(idx)], w[tuple(idx)])
    print(len(poses), "mismatch", nbad, "of", tot, "max ulp", mx)
F=np.float32
vol, ora = _scene(0.05,0.5,H,W,K,[_pose()], lambda T: plane_depth(K,T,H,W,8.0))
got=_bricks(vol)
for key,(t,w) in got.items():
    o=ora.units[key][0]
    d=(t.view(np.int32).astype(np.int64)-o.view(np.int32).astype(np.int64))
    if (d!=0).any():
        z,y,x=np.argwhere(d!=0)[0]
        ux,uy,uz=key
        voxel=F(0.05); unit=F(voxel*F(16))
        px=F(F(ux)*unit+F(F(x+0.5)*voxel)); py=F(F(uy)*unit+F(F(y+0.5)*voxel)); pz=F(F(uz)*unit+F(F(z+0.5)*voxel))
        fx=F(120.0); cx=F(31.5)
        uf=F(F(F(F(px*fx)/pz)+cx)+F(0.5)); vf=F(F(F(F(py*fx)/pz)+cx)+F(0.5))
        u=int(uf); v=int(vf)
        dd=plane_depth(K,_pose(),H,W,8.0)[v,u]
        rx=F(F(F(u)-cx)/fx); ry=F(F(F(v)-cx)/fx)
        mult=F(np.sqrt(F(F(F(rx*rx)+F(ry*ry))+F(1))))
        sdf=F(F(dd-pz)*mult)
        print(key,(x,y,z),"gpu",repr(t[z,y,x]),"ora",repr(o[z,y,x]),"hand",repr(min(F(1),F(sdf*F(2)))),"px,py,pz",px,py,pz,"uf,vf",uf,vf,"dd",dd,"mult",repr(mult),"sdf",repr(sdf))
        break
