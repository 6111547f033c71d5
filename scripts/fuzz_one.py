"""Development: one seed of scripts/fuzz_scenes.py, wavefront pipeline only, with the mismatching pixels listed.
    python scripts/fuzz_one.py SEED MAX_SPHERES [W H]"""
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

seed, max_spheres = int(sys.argv[1]), int(sys.argv[2])
W, H = (48, 30) if len(sys.argv) < 5 else (int(sys.argv[3]), int(sys.argv[4]))
port = _oracle.Port()
d = T.random_scene(seed, W, H, max_spheres)
d.flags = seed % 2
r = pt.Renderer(d)
s = T.oracle_scene(port, r, W, H, seed, flags=d.flags)
want, _ = port.render(s, MODE_PIXEL, 0, 3)
r.set_pipeline(pt.PIPELINE_WAVEFRONT)
b = r.render(0, 3)
bad = (b.view(np.uint64) != want.view(np.uint64)).any(axis=2)
print("seed", seed, "bodies", d.body_count, "mismatching pixels", int(bad.sum()), "of", bad.size, flush=True)
r.close()
