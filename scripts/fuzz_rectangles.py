"""One-off fuzzing of raytrace3.cpp's path tracer beyond the committed seeds: random rectangle
scenes (tests/_scenes.py: rectangle_scene) against the oracle, frames and nearest hits.
    python scripts/fuzz_rectangles.py FIRST LAST    (needs a GPU; prints the seeds that differ)"""
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
from _scenes import rectangle_scene  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
port = _oracle.Port()
W, H = 40, 28
bad = []
for seed in range(first, last):
    d = rectangle_scene(seed, W, H, n_rect=1 + seed % 12, n_sphere=seed % 9, n_plane=seed % 5)
    r = pt.Renderer(d)
    s = T.oracle_scene(port, r, W, H, seed, flags=pt.F_RAYTRACE3)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    got = r.render(0, 3)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    got_wf = r.render(0, 3)
    r.set_pipeline(pt.PIPELINE_FUSED)
    rng = np.random.default_rng(seed)
    rays = np.concatenate([rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5], (3000, 3)),
                           rng.normal(size=(3000, 3)) * rng.uniform(0.05, 4.0, (3000, 1))], axis=1)
    t_all, idx, t = port.intersect(s, rays)
    gi, gt = r.intersect_batch(rays)
    nf = int((got.view(np.uint64) != want.view(np.uint64)).sum()) + int((got_wf.view(np.uint64) != want.view(np.uint64)).sum())
    nr = int((gi != idx).sum()) + int((r.intersect_all(rays).view(np.uint64) != t_all.view(np.uint64)).sum())
    if nf or nr:
        bad.append((seed, nf, nr))
        print("seed", seed, "frame mismatches", nf, "ray mismatches", nr, flush=True)
    r.close()
    if seed % 100 == 0:
        print("... seed", seed, flush=True)
print("done", first, last, "bad:", bad)
