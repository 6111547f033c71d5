"""One-off: the headline resolution against the oracle, bit for bit (a few minutes of CPU).
    python scripts/full_frame_check.py [iterations] [scene] [width height]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracer_amd as pt  # noqa: E402
import _oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from _oracle import MODE_PIXEL  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = sys.argv[2] if len(sys.argv) > 2 else "nee15"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 768)
port = _oracle.Port()
r = pt.Renderer(pt.preset(scene, W, H, seed=1))
s = T.oracle_scene(port, r, W, H, 1, flags=r.desc.flags)
first = 7200 - n_it  # the last iterations of the headline render
t0 = time.time()
want, _ = port.render(s, MODE_PIXEL, first, n_it)
print(f"oracle: {time.time() - t0:.1f} s", flush=True)
got = r.render(first, n_it)
bad = int((got.view(np.uint64) != want.view(np.uint64)).sum())
print(scene, W, H, "iterations", first, "..", first + n_it - 1, "mismatching entries:", bad, "mean", got.mean() / n_it)
for world in (2, 8):
    total = np.zeros_like(got)
    for rank in range(world):
        r.set_partition(rank, world)
        total += r.render(first, n_it)
    r.set_partition(0, 1)
    print("  partition over", world, "ranks: mismatching entries", int((total.view(np.uint64) != want.view(np.uint64)).sum()))
