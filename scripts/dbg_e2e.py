import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch
from alicevision_amd import abi
from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
from common import small_case, make_oracle, make_hip_from_oracle
from oracle import oracle
mode=abi.FILTER_CUDA_FIXED8
sc, sgm, ref, depths = small_case(width=320, height=240, n_planes=48, seed=11)
o = make_oracle(sc, sgm, ref, filter_mode=mode)
oracle.load().avo_set_ncc_precision(1); oracle.load().avo_set_exact_rc_pixel(1)
o.run_sgm(0,[1,2],depths); want=o.run_refine(0,[1,2])
own = len(sys.argv)>1
pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, mode) for i in range(3)] if own else None
h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref) if own else make_hip_from_oracle(o, sc, sgm, ref)
h.run_sgm(0,[1,2],depths, keep_raw=True); got=h.run_refine(0,[1,2]).cpu().numpy(); torch.cuda.synchronize()
Z=len(depths)
def st(name,a,b):
    d=np.abs(a.astype(np.float64)-b.astype(np.float64)); print(name,'mismatch frac',(d>0).mean(),'max',d.max(),'mean',d.mean())
st('second', o.second[...,:Z], h.second.cpu().numpy()[...,:Z])
st('filtered', o.filtered[...,:Z], h.best.cpu().numpy()[...,:Z])
st('sgm depth', o.sgm_depth_thickness[...,0], h.sgm_depth_sim.cpu().numpy()[...,0])
st('sgm thick smooth', o.sgm_depth_thickness_smooth[...,1], h.sgm_depth_thickness.cpu().numpy()[...,1])
st('upscaled d', o.sgm_upscaled[...,0], h.sgm_upscaled.cpu().numpy()[...,0])
Zr=31
a=o.refine_volume[...,:Zr].astype(np.float32); b=h.refine_volume.cpu().numpy()[...,:Zr].astype(np.float32)
d=np.abs(a-b); print('refine vol: max',d.max(),'frac>2e-3',(d>2e-3).mean(),'frac>0.02',(d>0.02).mean())
ys,xs,zs=np.nonzero(d>0.02); print(list(zip(ys[:10],xs[:10],zs[:10])), a[d>0.02][:10], b[d>0.02][:10])
st('refined d', o.refined[...,0], h.refined.cpu().numpy()[...,0])
st('opt d', want[...,0], got[...,0])
both=(want[...,0]>0)&(got[...,0]>0); err=(got[...,0]-want[...,0])[both]
print('rmse all',np.sqrt(np.mean(err**2)),'rmse 99.5%',np.sqrt(np.mean(np.sort(err**2)[:int(0.995*err.size)])),'frac |err|>1e-3',(np.abs(err)>1e-3).mean())
pix=o.sgm_upscaled[...,1][both]; print('median pixSize',np.median(pix),'rmse/pixSize',np.sqrt(np.mean((err/pix)**2)))
