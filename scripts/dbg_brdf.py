import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import path_tracer_amd as pt
import _oracle
port=_oracle.Port()
rng=np.random.default_rng(3)
n=400000
def unit(n):
    v=rng.normal(size=(n,3)); return v/np.linalg.norm(v,axis=1,keepdims=True)
a=port.vector_to_spherical(unit(n)); b=port.vector_to_spherical(unit(n))
ang=np.concatenate([a,b],axis=1)
r=pt.Renderer(pt.preset('nee15_measured',8,8))
tab=np.concatenate([np.arange(pt.BRDF_TABLE_DOUBLES//3,dtype=np.float64)]*3)
r.brdf_upload(0,tab)
rgb,ind=r.brdf_lookup(0,ang)
rgb0,ind0=port.brdf_lookup(tab,ang)
bad=np.nonzero(ind!=ind0)[0]
print('mismatch',len(bad),'of',n)
for i in bad[:12]:
    def dec(x): return (x//(180*90), (x//180)%90, x%180)
    print(i, ang[i], 'gpu',dec(ind[i]),'cpu',dec(ind0[i]))
