import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import path_tracer_amd as pt
import _oracle
port=_oracle.Port()
rng=np.random.default_rng(3)
n=20000
def unit(n):
    v=rng.normal(size=(n,3)); return v/np.linalg.norm(v,axis=1,keepdims=True)
r=pt.Renderer(pt.preset('nee15_measured',8,8))
tab=np.concatenate([np.arange(pt.BRDF_TABLE_DOUBLES//3,dtype=np.float64)]*3)
r.brdf_upload(0,tab)
def dec(x): return (x//(180*90), (x//180)%90, x%180)
for label,special in (('-0',-0.0),('+0',0.0),('pi/2',np.pi/2),('-pi/2',-np.pi/2),('nan',np.nan)):
  for which in (0,2):
    a=port.vector_to_spherical(unit(n)); b=port.vector_to_spherical(unit(n))
    ang=np.concatenate([a,b],axis=1); ang[:,which]=special
    rgb,ind=r.brdf_lookup(0,ang); rgb0,ind0=port.brdf_lookup(tab,ang)
    bad=np.nonzero(ind!=ind0)[0]
    print(label,'col',which,'mismatch',len(bad),'of',n)
    for i in bad[:3]: print('   ',ang[i],'gpu',dec(ind[i]),'cpu',dec(ind0[i]))
