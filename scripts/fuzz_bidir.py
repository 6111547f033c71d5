"""One-off fuzzing of the bidirectional estimator beyond the committed seeds: random scenes whose lights
have colour zero (tests/test_bidirectional.py: random_dark_scene) against the oracle.
    python scripts/fuzz_bidir.py FIRST LAST    (needs a GPU; prints the seeds that differ)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_tracer_amd as pt  # noqa: E402
import _oracle  # noqa: E402
import test_bidirectional as TB  # noqa: E402
from _oracle import MODE_PIXEL  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
port = _oracle.Port()
W, H = 48, 32
bad = []
for seed in range(first, last):
    try:
        r = pt.Renderer(TB.random_dark_scene(seed, W, H))
    except pt.PtError:
        continue  # outside the estimator's domain (no light, too many bodies)
    want, _ = port.render(TB.oracle_scene(port, r, W, H, seed), MODE_PIXEL, 1, 3)
    if not TB.same_bits(r.render(1, 3), want):
        bad.append(seed)
    r.close()
    if seed % 100 == 0:
        print("... seed", seed, flush=True)
print("done", first, last, "bad:", bad)
