"""Development: where does a fuzz seed differ?  Nearest hits of random rays and per-path diagnostics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracer_amd as pt  # noqa: E402
import _oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402

seed, max_spheres = int(sys.argv[1]), int(sys.argv[2])
W, H = 48, 30
port = _oracle.Port()
d = T.random_scene(seed, W, H, max_spheres)
d.flags = seed % 2
r = pt.Renderer(d)
s = T.oracle_scene(port, r, W, H, seed, flags=d.flags)
print("bodies", d.body_count, "estimator", d.estimator, "flags", d.flags)
rng = np.random.default_rng(3)
n = 4000
rays = np.concatenate([np.stack([rng.uniform(-1.8, 1.8, n), rng.uniform(-2.4, 4.4, n), rng.uniform(-0.45, 2.4, n)], axis=1),
                       rng.normal(size=(n, 3))], axis=1)
idx, t = r.intersect_batch(rays)
_, idx0, t0 = port.intersect(s, rays)
badi = np.nonzero(idx != idx0)[0]
print("nearest-hit mismatches", len(badi), "of", n, badi[:10], idx[badi[:10]], idx0[badi[:10]], t[badi[:5]], t0[badi[:5]])
m = 3000
it_idx = np.stack([rng.integers(0, 3, m), rng.integers(0, W * H, m)], axis=1).astype(np.int32)
xy = rng.uniform(0, 1, size=(m, 2))
g = r.trace_paths(it_idx, xy)
o = port.trace_paths(s, it_idx, xy)
for k in ("bodies", "visible", "draws"):
    bad = np.nonzero((g[k] != o[k]).reshape(m, -1).any(axis=1))[0]
    print(k, "mismatching paths", len(bad), bad[:5])
    if len(bad):
        print("  gpu", g[k][bad[0]], "oracle", o[k][bad[0]])
bad = np.nonzero((g["rgb"].view(np.uint64) != o["rgb"].view(np.uint64)).any(axis=1))[0]
print("rgb mismatching paths", len(bad))
r.close()
