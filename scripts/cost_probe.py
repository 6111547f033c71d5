"""Instruction-cost probe: runs the diagnostic kernels on realistic inputs so that a
`rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES` pass gives VALU instructions per scan and per path.
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d OUT -- python3 scripts/cost_probe.py [scene]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pt  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "nee15"
W, H = 1024, 768
r = pt.Renderer(pt.preset(scene, W, H, seed=1))
n = 1 << 20
rs = np.random.RandomState(3)
# paths of iteration 0 in pixel order (coherent waves, like the render kernel's)
idx = np.arange(n, dtype=np.int32) % (W * H)
it_idx = np.stack([np.zeros(n, dtype=np.int32), idx], axis=1)
xy = np.stack([(idx % W + 0.5) / W, (idx // W + 0.5) / H], axis=1)
out = r.trace_paths(it_idx, xy)
pts = out["points"]
nrec = out["nrec"]
print("paths", n, "mean vertices", nrec.mean(), "mean draws", out["draws"].mean())
# gather rays between consecutive vertices, grouped by depth so that a wave holds one depth
for k in range(1, 5):
    ok = nrec > k
    o = pts[ok, k - 1]
    d = pts[ok, k] - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rs.uniform(0.3, 1.0, size=(d.shape[0], 1))  # bounce directions are not normalised
    rays = np.concatenate([o, d], axis=1)
    hit, t = r.intersect_batch(rays)
    print("depth", k, "rays", rays.shape[0], "hit fraction", (hit >= 0).mean())
