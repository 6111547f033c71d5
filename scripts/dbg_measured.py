import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import path_tracer_amd as pt
import _oracle
from test_gpu_parity import oracle_scene
port=_oracle.Port()
rng=np.random.default_rng(11); n=40000
r=pt.Renderer(pt.preset('nee15_measured',64,64,seed=99))
tables=[pt.synthetic_brdf_table(k) for k in range(4)]
for k,t in enumerate(tables): r.brdf_upload(k,t)
s=oracle_scene(port,r,64,64,99,tables=tables)
it_idx=np.stack([rng.integers(0,7200,n),rng.integers(0,1<<20,n)],axis=1).astype(np.int32)
xy=rng.uniform(0,1,size=(n,2))
g=r.trace_paths(it_idx,xy); o=port.trace_paths(s,it_idx,xy)
print('bodies mism',(g['bodies']!=o['bodies']).any(axis=1).sum(),'vis mism',(g['visible']!=o['visible']).sum(),'draws mism',(g['draws']!=o['draws']).sum())
bad=np.nonzero((g['rgb']!=o['rgb']).any(axis=1))[0]
rel=np.abs(g['rgb']-o['rgb'])/np.maximum(1e-300,np.abs(o['rgb']))
print('rgb mism',len(bad),'max rel',rel.max(), 'count rel>1e-9',(rel>1e-9).any(axis=1).sum())
big=np.nonzero((rel>1e-9).any(axis=1))[0]
for i in big[:10]:
    print(i,g['bodies'][i],bin(g['visible'][i]),g['rgb'][i],o['rgb'][i])
    print('   pts',g['points'][i].tolist())
