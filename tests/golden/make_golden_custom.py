"""Golden frames of the reference's own Renderer::iterate() on scenes it does not hard-code.

Run in the build container only (needs oracle/_ref: `make -C oracle ref`).  The body lists are
constructor arguments (taken from this repository's scene presets / test scene variants); the
harness builds the reference's Sphere / Plane / Material / Glass / Chrome / Body objects from them
with the reference's constructors and runs the reference's unmodified Renderer on that scene, with
per-pixel random streams.  Every expected frame is therefore produced by the reference's code.
Outputs: frames_custom.npz (body records as bytes + frames), frames_custom.json.

    python tests/golden/make_golden_custom.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import path_tracer_amd as pt  # noqa: E402  (host-side scene presets only; no GPU is touched)
from _oracle import MODE_PIXEL, Body, Ref, bodies_to_bytes  # noqa: E402
from _scenes import open_scene, scene_with_extra_planes  # noqa: E402

CASES = [
    # key, reference program (0 main.cpp, 1 next_event.cpp), scene, W, H, first iteration, iterations, seed
    ("bipath23", 1, lambda W, H: pt.preset("bipath23", W, H), 36, 20, 2, 3, 5),
    ("nee32", 1, lambda W, H: pt.preset("nee32", W, H), 28, 20, 0, 3, 6),
    ("nee15_extra_planes", 1, lambda W, H: scene_with_extra_planes("nee15", W, H, 1), 30, 18, 0, 3, 9),
    ("main32_extra_planes", 0, lambda W, H: scene_with_extra_planes("main32", W, H, 1), 30, 18, 1, 2, 9),
    ("nee15_open", 1, lambda W, H: open_scene("nee15", W, H, 1, drop=(7, 11, 12, 13)), 24, 16, 0, 3, 17),
    ("synthetic1k", 1, lambda W, H: pt.preset("synthetic1k", W, H), 12, 8, 0, 1, 4),
]


def main():
    refh = Ref(hooked=True)
    arrays, meta = {}, []
    for key, prog, make, W, H, first, n_it, seed in CASES:
        d = make(W, H)
        n = d.body_count
        ob = (Body * n)()
        for i in range(n):
            C.memmove(C.byref(ob[i]), C.byref(d.bodies[i]), C.sizeof(Body))
        frame, draws = refh.render_custom(prog, ob, n, W, H, first, n_it, MODE_PIXEL, seed=seed)
        arrays[key + "_bodies"] = bodies_to_bytes(ob, n)
        arrays[key + "_frame"] = frame
        meta.append(dict(key=key, prog=prog, body_record=C.sizeof(Body), bodies=n, W=W, H=H, first=first, n_it=n_it, seed=seed, draws=int(draws)))
        print(key, n, "bodies,", int(draws), "draws, mean", frame.mean())
    np.savez_compressed(os.path.join(HERE, "frames_custom.npz"), **arrays)
    with open(os.path.join(HERE, "frames_custom.json"), "w") as f:
        json.dump(meta, f, indent=1)

    # save() on values a converged render does not contain: NaN, infinities, negatives, huge and
    # tiny numbers, exact boundaries of the truncation to int
    ref = Ref()
    W, H, samples = 6, 3, 3
    vals = [float("nan"), float("inf"), -float("inf"), -0.0, -1e-300, -0.5, -7.0, 1e300, 3.0 * 2147483647.0 / 255.0,
            3.0 * 2147483648.0 / 255.0, 3.0 * 256.0 / 255.0, 3.0 * 255.999999 / 255.0, 3.0, 2.9999999999999996,
            3.0000000000000004, 1e-320, 0.011764705882352941 * 3.0, 0.5]
    buf = np.resize(np.array(vals, dtype=np.float64), (H, W, 3))
    ppm = {}
    for prog in (0, 1):
        data, fname = ref.save(prog, buf, samples, W, H)
        ppm[str(prog)] = dict(name=fname, text=data.decode())
    with open(os.path.join(HERE, "save_edge.json"), "w") as f:
        json.dump(dict(W=W, H=H, samples=samples, values_hex=[float(v).hex() for v in buf.reshape(-1)], ppm=ppm), f)


if __name__ == "__main__":
    main()
