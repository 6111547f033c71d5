"""One-off fuzzing beyond the committed seeds: random scenes, both pipelines, against the oracle.
    python scripts/fuzz_scenes.py FIRST LAST [MAX_SPHERES [W H]]    (needs a GPU; prints the seeds that differ)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracer_amd as pt  # noqa: E402
import _oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from _oracle import MODE_PIXEL  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
max_spheres = int(sys.argv[3]) if len(sys.argv) > 3 else 40
port = _oracle.Port()
W, H = (32, 20) if max_spheres <= 40 else (48, 30)
if len(sys.argv) > 5:
    W, H = int(sys.argv[4]), int(sys.argv[5])
bad = []
for seed in range(first, last):
    d = T.random_scene(seed, W, H, max_spheres)
    d.flags = seed % 2  # every other scene casts shadow rays from the light's real centre
    r = pt.Renderer(d)
    s = T.oracle_scene(port, r, W, H, seed, flags=d.flags)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    a = r.render(0, 3)  # fused up to 127 bodies, else the wavefront pipeline
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    b = r.render(0, 3)
    na = int((a.view(np.uint64) != want.view(np.uint64)).sum())
    nb = int((b.view(np.uint64) != want.view(np.uint64)).sum())
    if na or nb:
        bad.append((seed, na, nb))
        print("seed", seed, "fused mismatches", na, "wavefront mismatches", nb, flush=True)
    r.close()
    if seed % 50 == 0:
        print("... seed", seed, flush=True)
print("done", first, last, "bad:", bad)
