"""Golden vectors of the reference's raytrace3.cpp (Rectangle shape, hit threshold 1e-7).

Run in the build container only (needs oracle/_ref: `make -C oracle ref`).  Everything expected is
produced by the reference's own classes: the scene raytrace3.cpp builds in its main(), per-body
Shape::intersect values on a ray batch, frames of its Renderer::iterate() with per-pixel random
streams, and frames of the same Renderer on body lists built from constructor arguments
(tests/_scenes.py: rectangle_scene).  Outputs: raytrace3.npz (232-byte body records as bytes, rays,
t values, frames), raytrace3.json.

    python tests/golden/make_golden_raytrace3.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _oracle import MODE_PIXEL, Body, Ref, bodies_to_bytes  # noqa: E402
from _scenes import rectangle_scene  # noqa: E402

PROG = 2  # raytrace3.cpp in the harness


def ray_batch(bodies, n, rng):
    """Rays inside the room: random, towards rectangle corners and edges, parallel to walls."""
    N = 3000
    o = rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5], (N, 3))
    d = rng.normal(size=(N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.uniform(0.2, 3.0, (N, 1))
    rays = [np.concatenate([o, d], axis=1)]
    special = []
    for i in range(n):
        if bodies[i].shape != 2:
            continue
        c = np.array(list(bodies[i].corners)).reshape(4, 3)
        for q in range(4):
            for origin in ([0.0, 1.0, 1.0], [0.3, -1.0, 0.2]):
                origin = np.array(origin)
                special.append(np.concatenate([origin, c[q] - origin]))            # exactly at a corner, t = 1
                special.append(np.concatenate([origin, 0.5 * (c[q] - origin)]))    # t = 2
                special.append(np.concatenate([origin, 2.0 * (c[q] - origin)]))    # t = 0.5: a miss by t < 1
                mid = 0.5 * (c[q] + c[(q + 1) % 4])
                special.append(np.concatenate([origin, 0.7 * (mid - origin)]))     # on an edge
    for axis in range(3):  # parallel to the walls
        e = np.zeros(3)
        e[axis] = 1.0
        special.append(np.concatenate([[0.1, 1.0, 0.3], e]))
        special.append(np.concatenate([[0.1, 1.0, 0.3], -e]))
    rays.append(np.array(special))
    return np.concatenate(rays, axis=0)


def main():
    ref = Ref()
    refh = Ref(hooked=True)
    rng = np.random.default_rng(33)
    arrays, meta = {}, dict(body_record=C.sizeof(Body), frames=[], custom=[])
    bodies, n, cam = ref.scene_bodies(PROG)
    arrays["scene_bodies"] = bodies_to_bytes(bodies, n)
    meta["bodies"] = n
    rays = ray_batch(bodies, n, rng)
    arrays["rays"] = rays
    arrays["t_all"] = ref.intersect_all(PROG, rays, n)
    for key, W, H, first, n_it, seed in [("a", 48, 36, 0, 3, 5), ("b", 33, 20, 7, 2, 9), ("c", 1024, 3, 2, 1, 4)]:
        frame, draws = refh.render(PROG, W, H, first, n_it, MODE_PIXEL, seed=seed)
        arrays["frame_" + key] = frame
        meta["frames"].append(dict(key=key, W=W, H=H, first=first, n_it=n_it, seed=seed, draws=int(draws)))
        print("frame", key, int(draws), "draws, mean", frame.mean())
    for seed in (1, 2, 3, 4):
        W, H, first, n_it = 40, 28, 1, 2
        d = rectangle_scene(seed, W, H)
        nb = d.body_count
        ob = (Body * nb)()
        for i in range(nb):
            C.memmove(C.byref(ob[i]), C.byref(d.bodies[i]), C.sizeof(Body))
        frame, draws = refh.render_custom(PROG, ob, nb, W, H, first, n_it, MODE_PIXEL, seed=seed)
        key = "custom%d" % seed
        arrays[key + "_bodies"] = bodies_to_bytes(ob, nb)
        arrays[key + "_frame"] = frame
        meta["custom"].append(dict(key=key, bodies=nb, W=W, H=H, first=first, n_it=n_it, seed=seed, draws=int(draws)))
        print(key, nb, "bodies,", int(draws), "draws, mean", frame.mean())
    np.savez_compressed(os.path.join(HERE, "raytrace3.npz"), **arrays)
    with open(os.path.join(HERE, "raytrace3.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
