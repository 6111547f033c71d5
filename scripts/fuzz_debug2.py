"""Development: variants of one fuzz scene (fewer lights, no glass, ...) against the oracle."""
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
W, H = 48, 30
port = _oracle.Port()


def run(tag, edit):
    d = T.random_scene(seed, W, H, max_spheres)
    d.flags = seed % 2
    edit(d)
    r = pt.Renderer(d)
    s = T.oracle_scene(port, r, W, H, seed, flags=d.flags)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    b = r.render(0, 3)
    bad = (b.view(np.uint64) != want.view(np.uint64)).any(axis=2)
    print(tag, "bodies", d.body_count, "mismatching pixels", int(bad.sum()), "of", bad.size, flush=True)
    r.close()


def few_lights(d):
    seen = 0
    for i in range(d.body_count):
        b = d.bodies[i]
        if max(b.emission) >= 1.0:
            seen += 1
            if seen > 2:
                for c in range(3):
                    b.emission[c] = 0.5


def no_glass(d):
    for i in range(d.body_count):
        if d.bodies[i].material == pt.GLASS:
            d.bodies[i].material = pt.DIFFUSE


def no_chrome(d):
    for i in range(d.body_count):
        if d.bodies[i].material == pt.CHROME:
            d.bodies[i].material = pt.DIFFUSE


def real_centres(d):
    d.flags = 1


def origin0(d):
    d.flags = 0


def keep_lights(which):
    def edit(d):
        idx = [i for i in range(d.body_count) if max(d.bodies[i].emission) >= 1.0]
        keep = set(which(idx))
        for i in idx:
            if i not in keep:
                for c in range(3):
                    d.bodies[i].emission[c] = 0.5
    return edit


run("original      ", lambda d: None)
run("first 2 lights", keep_lights(lambda idx: idx[:2]))
run("last 2 lights ", keep_lights(lambda idx: idx[-2:]))
run("first 16      ", keep_lights(lambda idx: idx[:16]))
run("first 40      ", keep_lights(lambda idx: idx[:40]))
run("last 40       ", keep_lights(lambda idx: idx[-40:]))
