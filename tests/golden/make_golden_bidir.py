"""Golden vectors of bidirectional.cpp's estimator (bidirectional.cpp:374-497), made by the reference's
own code.  Run in the build container only (needs oracle/_ref: `make -C oracle ref ref-O0`).

1. bidir_ub_check.json — the estimator indexes flux_lookup[5][4] with [shoot_hits.size()][shoot_i]
   (:377-382, :484); a light path of five vertices reads past the end of the array.  The value only
   scales emission*colour of the chosen light, (0,0,0) for every light of the program's scene, so the
   frame should not depend on it.  Checked here: the reference built at -O3 and at -O0 (different
   stack layouts), each run in two fresh processes, 32x24 pixels x 4 iterations from one sequential
   random stream: four identical hashes, no NaN.
2. bidir.npz / bidir.json — the scene main() builds (body records, camera), frames of the reference's
   Renderer::iterate() with per-pixel random streams on that scene, and frames on body lists built
   from constructor arguments (an open room, whose light paths leave the scene after fewer than five
   vertices; a room with a second glass ball and a chrome wall; lights on planes only).

    python tests/golden/make_golden_bidir.py
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import _oracle  # noqa: E402
from _oracle import MODE_PIXEL, MODE_SEQ, Body, Ref, bodies_to_bytes, camera_to_list  # noqa: E402

PROG = 3  # bidirectional.cpp in oracle/ref_harness.cpp


def one_hash(lib):
    _oracle.REF_SO = lib
    buf, draws = Ref().render(PROG, 32, 24, 0, 4, MODE_SEQ, seed=7)
    print(json.dumps(dict(sha256=hashlib.sha256(buf.tobytes()).hexdigest(), draws=int(draws), nan=int(np.isnan(buf).sum()),
                          inf=int(np.isinf(buf).sum()), mean=float(buf.mean() / 4))))


def variants(bodies, n):
    """Body lists derived from the program's scene (constructor arguments)."""
    def copy():
        ob = (Body * 64)()
        for i in range(n):
            C.memmove(C.byref(ob[i]), C.byref(bodies[i]), C.sizeof(Body))
        return ob

    out = []
    # an open room: without the back wall, the front wall and the two dividers many gather and light
    # paths end early (fewer than five vertices: rows 0..4 of flux_lookup)
    ob = copy()
    keep = [i for i in range(n) if i not in (7, 11, 12, 13)]
    for k, i in enumerate(keep):
        C.memmove(C.byref(ob[k]), C.byref(bodies[i]), C.sizeof(Body))
    out.append(("open", ob, len(keep)))
    # a second large glass ball, and the right wall turned into chrome
    ob = copy()
    ob[9].material = _oracle.CHROME
    ob[n].shape, ob[n].material = _oracle.SPHERE, _oracle.GLASS
    ob[n].center[0], ob[n].center[1], ob[n].center[2], ob[n].radius = -0.6, 1.7, 0.2, 0.45
    ob[n].color[0], ob[n].color[1], ob[n].color[2] = 0.9, 1.0, 0.8
    ob[n].ior, ob[n].reflection, ob[n].brdf_id = 1.33, 0.05, -1
    out.append(("glass_chrome", ob, n + 1))
    # only the ceiling and one ball light: two lights, so the pick always takes the second
    ob = copy()
    out.append(("two_lights", ob, 15))
    return out


def main():
    libs = [os.path.join(ROOT, "oracle", "_ref", f) for f in ("libptref.so", "libptref_O0.so")]
    runs = []
    for lib in libs:
        for _ in range(2):
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--hash", lib], capture_output=True, text=True, check=True)
            runs.append(dict(build=os.path.basename(lib), **json.loads(o.stdout.strip().splitlines()[-1])))
    same = len({r["sha256"] for r in runs}) == 1 and all(r["nan"] == 0 and r["inf"] == 0 for r in runs)
    with open(os.path.join(HERE, "bidir_ub_check.json"), "w") as f:
        json.dump(dict(what="bidirectional.cpp, 32x24, iterations 0-3, sequential stream seed 7; g++ -O3 -fsanitize=unreachable "
                            "and g++ -O0, two processes each", frames_agree_and_finite=same, runs=runs), f, indent=1)
    print("flux_lookup check:", "agree" if same else "DIFFER", [r["sha256"][:12] for r in runs])
    assert same

    refh = Ref(hooked=True)
    bodies, n, cam = refh.scene_bodies(PROG)
    arrays = {"scene_bodies": bodies_to_bytes(bodies, n)}
    meta = dict(body_record=C.sizeof(Body), bodies=n, camera=camera_to_list(cam), frames=[])
    for key, W, H, first, n_it, seed in (("a", 32, 24, 0, 3, 5), ("b", 45, 19, 7, 2, 2025)):
        frame, draws = refh.render(PROG, W, H, first, n_it, MODE_PIXEL, seed=seed)
        arrays["frame_" + key] = frame
        meta["frames"].append(dict(key=key, W=W, H=H, first=first, n_it=n_it, seed=seed, draws=int(draws)))
        print("frame", key, int(draws), "draws, mean", frame.mean() / n_it)
    meta["custom"] = []
    for key, ob, cnt in variants(bodies, n):
        W, H, first, n_it, seed = 26, 18, 1, 3, 31
        frame, draws = refh.render_custom(PROG, ob, cnt, W, H, first, n_it, MODE_PIXEL, seed=seed)
        arrays["custom_" + key + "_bodies"] = bodies_to_bytes(ob, cnt)
        arrays["custom_" + key + "_frame"] = frame
        meta["custom"].append(dict(key=key, bodies=cnt, W=W, H=H, first=first, n_it=n_it, seed=seed, draws=int(draws)))
        print("custom", key, cnt, "bodies,", int(draws), "draws, mean", frame.mean() / n_it, "nan", int(np.isnan(frame).sum()))
    np.savez_compressed(os.path.join(HERE, "bidir.npz"), **arrays)
    with open(os.path.join(HERE, "bidir.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--hash":
        one_hash(sys.argv[2])
    else:
        main()
