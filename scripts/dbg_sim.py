import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np, torch
from alicevision_amd import abi
from common import small_case, make_oracle, make_hip_from_oracle
import ctypes
if len(sys.argv)>1:
    abi._lib=None; abi.LIB_PATH=sys.argv[1]
sc, sgm, ref, depths = small_case()
o = make_oracle(sc, sgm, ref)
from oracle import oracle as _o
_o.load().avo_set_ncc_precision(int(os.environ.get('F64','1')))
o.run_sgm(0, [1], depths, optimize=False)
h = make_hip_from_oracle(o, sc, sgm, ref)
h.run_sgm(0, [1], depths, optimize=False)
torch.cuda.synchronize()
Z=len(depths)
a=o.best_raw[...,:Z].astype(int); b=h.best.cpu().numpy()[...,:Z].astype(int)  # after update_uninit best unchanged
print('shape',a.shape,'oracle 255 frac',(a==255).mean(),'hip 255 frac',(b==255).mean())
d=np.abs(a-b)
print('mismatch frac',(d>0).mean(),'>1',(d>1).mean(),'mask mismatch',((a==255)!=(b==255)).mean())
both=(a!=255)&(b!=255)
print('valid both: mismatch',(d[both]>0).mean(),'>1',(d[both]>1).mean(),'mean abs',d[both].mean())
print('per z mismatch',[(round((d[...,z]>0).mean(),3)) for z in range(0,Z,4)])
ys,xs,zs=np.nonzero(d>1)
for i in range(min(10,len(ys))): print(ys[i],xs[i],zs[i],a[ys[i],xs[i],zs[i]],b[ys[i],xs[i],zs[i]])
print(a[24,32,:16]); print(b[24,32,:16])
