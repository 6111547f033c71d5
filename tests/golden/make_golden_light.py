"""Golden vectors of light_tracing.cpp's Rectangle (two Moller-Trumbore triangle tests,
light_tracing.cpp:477-575), made by the reference's own class.  Run in the build container only
(needs oracle/_ref: `make -C oracle ref`).

light_tracing.cpp cannot be built as a program here (it needs kdtree.h for its photon map), but
everything before `struct Scene` — vectors, rays, shapes, materials — needs neither that nor Boost;
oracle/Makefile compiles that head of the file, and the harness builds Rectangle(p1, p2, p3, p4)
objects from constructor arguments and calls their intersect().  For each of a few seeded scenes:
the corner points of its rectangles (planar ones, and warped ones whose two triangles are not
coplanar), a ray batch (random rays, rays aimed at corners, edges and the shared diagonal, rays in
the plane, rays starting on it) and per rectangle the stored normal and every ray's intersect().
Output: light_rectangle.npz / light_rectangle.json.

    python tests/golden/make_golden_light.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import path_tracer_amd as pt  # noqa: E402  (host-side scene helpers only; no GPU is touched)
from _oracle import Ref  # noqa: E402
from _scenes import mt_edge_rays, mt_rectangle_scene  # noqa: E402

CASES = [(61, 0.0), (62, 0.2), (63, 0.0), (64, 0.05), (65, 1.0)]  # (seed, warp)


def main():
    ref = Ref()
    L = ref.lib
    arrays, meta = {}, []
    for seed, warp in CASES:
        d = mt_rectangle_scene(seed, 16, 16, n_rect=3 + seed % 5, n_sphere=0, n_plane=0, warp=warp)
        rects = [np.array(list(d.bodies[i].corners)) for i in range(d.body_count) if d.bodies[i].shape == pt.RECTANGLE_MT]
        rng = np.random.default_rng(seed)
        rays = np.concatenate([rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5], (600, 3)),
                               rng.normal(size=(600, 3)) * rng.uniform(0.1, 3.0, (600, 1))], axis=1)
        rays = np.ascontiguousarray(np.concatenate([rays, mt_edge_rays(d, rng, n_each=120)]))
        t = np.zeros((len(rects), len(rays)))
        normals = np.zeros((len(rects), 3))
        for k, c in enumerate(rects):
            c = np.ascontiguousarray(c)
            L.ptor_lt_rectangle(c.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), len(rays),
                                t[k].ctypes.data_as(C.c_void_p), normals[k].ctypes.data_as(C.c_void_p))
        key = f"s{seed}"
        arrays[key + "_corners"] = np.array(rects)
        arrays[key + "_rays"] = rays
        arrays[key + "_t"] = t
        arrays[key + "_normals"] = normals
        hits = int(np.count_nonzero(t > 0))
        meta.append(dict(key=key, seed=seed, warp=warp, rectangles=len(rects), rays=len(rays), positive=hits))
        print(key, len(rects), "rectangles,", len(rays), "rays,", hits, "positive distances")
    np.savez_compressed(os.path.join(HERE, "light_rectangle.npz"), **arrays)
    with open(os.path.join(HERE, "light_rectangle.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
