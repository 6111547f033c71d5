import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np
import path_tracer_amd as pt
import _oracle
from test_gpu_parity import oracle_scene
port=_oracle.Port()
name=sys.argv[1] if len(sys.argv)>1 else 'nee15'
rng = np.random.default_rng(11)
n = 20000
r = pt.Renderer(pt.preset(name, 64, 64, seed=99))
s = oracle_scene(port, r, 64, 64, 99)
it_idx = np.stack([rng.integers(0, 7200, n), rng.integers(0, 1 << 20, n)], axis=1).astype(np.int32)
xy = rng.uniform(0, 1, size=(n, 2))
g = r.trace_paths(it_idx, xy)
o = port.trace_paths(s, it_idx, xy)
bad = np.nonzero((g['bodies']!=o['bodies']).any(axis=1))[0]
print('mismatching paths', len(bad), 'of', n)
for i in bad[:8]:
    print(i, 'gpu', g['bodies'][i], 'cpu', o['bodies'][i], 'draws', g['draws'][i], o['draws'][i])
    k=np.nonzero(g['bodies'][i]!=o['bodies'][i])[0][0]
    print('  first diff at vertex',k,'pts gpu',g['points'][i][max(k-1,0)], 'cpu', o['points'][i][max(k-1,0)])
print('rgb mismatches', (g['rgb']!=o['rgb']).any(axis=1).sum(), 'visible mism', (g['visible']!=o['visible']).sum(), 'draws mism',(g['draws']!=o['draws']).sum())
pd = np.abs(g['points']-o['points']); print('max point diff', pd.max())
ok = ~(g['bodies']!=o['bodies']).any(axis=1)
print('max point diff among matching', pd[ok].max(), 'frac of exactly equal points', (g['points'][ok]==o['points'][ok]).mean())
