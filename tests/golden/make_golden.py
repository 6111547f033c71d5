"""Generate the golden vectors under tests/golden/ from the reference's own code.

Run in the build container only (needs oracle/_ref, i.e. `make -C oracle ref`, which compiles the
reference sources where they lie).  Every expected value below is produced by the reference's
classes/functions — Scene construction, Shape::intersect, Renderer::trace/iterate, Material::bounce,
save(), MatusikBRDF::lookup_brdf_val — never by this repository's code.  Inputs are seeded numpy
draws.  Outputs are small .json / .npz data files.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _oracle import (  # noqa: E402
    BRDF_N, MODE_PIXEL, MODE_SCRIPT, MODE_SEQ, RAND_MAX, Ref, bodies_to_list, camera_to_list,
)

ref = Ref()
refh = Ref(hooked=True)
rng = np.random.default_rng(20250211)


def room_rays(n):
    """Rays from random points inside the room in random directions (unit and non-unit)."""
    o = np.stack([rng.uniform(-1.8, 1.8, n), rng.uniform(-2.4, 4.4, n), rng.uniform(-0.45, 2.4, n)], axis=1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    scale = np.where(rng.uniform(size=n) < 0.5, 1.0, rng.uniform(0.05, 1.0, n))  # diffuse bounces are not unit
    return np.concatenate([o, d * scale[:, None]], axis=1)


def main():
    # --- scenes as the reference constructs them ---------------------------------------
    for prog, name in ((0, "main32"), (1, "nee15")):
        bodies, n, cam = ref.scene_bodies(prog)
        with open(os.path.join(HERE, f"scene_{name}.json"), "w") as f:
            json.dump(dict(program=["main.cpp", "next_event.cpp"][prog], camera=camera_to_list(cam),
                           bodies=bodies_to_list(bodies, n)), f, indent=1)

    # --- Camera::getRay + Shape::intersect + nearest hit --------------------------------
    xy = rng.uniform(0, 1, size=(256, 2))
    xy[:3] = [[0.5, 0.5], [0.8125, 0.4375], [0.2, 0.55]]
    out = {}
    for prog, name in ((0, "main32"), (1, "nee15")):
        cam_rays = ref.get_rays(prog, xy)
        rays = np.concatenate([cam_rays, room_rays(768)], axis=0)
        nb = 32 if prog == 0 else 15
        out[f"{name}_rays"] = rays
        out[f"{name}_t_all"] = ref.intersect_all(prog, rays, nb)
    idx, pts = ref.nearest_nee(out["nee15_rays"])
    out["nee15_nearest_idx"] = idx  # decided by the reference's own scan loop inside trace()
    out["xy"] = xy
    np.savez_compressed(os.path.join(HERE, "intersect.npz"), **out)

    # --- Material::bounce with scripted draws -------------------------------------------
    # every script holds at least one triple inside the unit ball, so the rejection loop of
    # getRandomNormalInHemisphere (main.cpp:107-109) terminates
    def frac(*fs):
        return [int(RAND_MAX * f) for f in fs]

    scripts = [
        [RAND_MAX // 2, RAND_MAX // 3, RAND_MAX // 5, RAND_MAX // 7, RAND_MAX // 11, int(RAND_MAX * 0.9)],
        frac(0.99, 0.98, 0.97) + frac(0.6, 0.4, 0.55) + [12345],
        [0, RAND_MAX, RAND_MAX // 2] + frac(0.5, 0.75, 0.3) + frac(0.05),
        [int(v) for v in rng.integers(0, RAND_MAX, size=31)],
    ]
    cases = []
    for prog in (0, 1):
        for body in (0, 1, 2, 5, 6, 12, 14):
            for si, sc in enumerate(scripts):
                ray = room_rays(1)[0]
                nrm = rng.normal(size=3)
                nrm /= np.linalg.norm(nrm)
                o, d = ref.bounce(prog, body, ray, nrm, sc)
                cases.append(dict(prog=prog, body=body, script=[int(v) for v in sc], ray=[float(v).hex() for v in ray],
                                  normal=[float(v).hex() for v in nrm], out=[float(v).hex() for v in o], draws=int(d)))
    with open(os.path.join(HERE, "bounce.json"), "w") as f:
        json.dump(cases, f, indent=0)

    # --- full frames through Renderer::iterate ------------------------------------------
    frames = {}
    meta = []
    for prog, name in ((0, "main32"), (1, "nee15")):
        for (W, H, first, n_it, mode, seed) in (
            (24, 16, 0, 3, MODE_SEQ, 7),
            (24, 16, 0, 3, MODE_PIXEL, 7),
            (40, 6, 5, 2, MODE_PIXEL, 12345678901234567),
            (7, 19, 0, 4, MODE_PIXEL, 1),
            (1, 1, 0, 8, MODE_PIXEL, 3),
        ):
            r = refh if mode == MODE_PIXEL else ref
            buf, draws = r.render(prog, W, H, first, n_it, mode, seed=seed)
            key = f"{name}_{W}x{H}_it{first}+{n_it}_mode{mode}_seed{seed}"
            frames[key] = buf
            meta.append(dict(key=key, scene=name, W=W, H=H, first=first, n_it=n_it, mode=mode, seed=seed, draws=int(draws)))
    # the survey-time known answer: 1x1 NEE iterate() with a cyclic 6-entry script
    script = [RAND_MAX // 2, RAND_MAX // 3, RAND_MAX // 5, RAND_MAX // 7, RAND_MAX // 11, int(RAND_MAX * 0.9)]
    buf, draws = ref.render(1, 1, 1, 0, 1, MODE_SCRIPT, script=script)
    frames["nee15_1x1_script"] = buf
    meta.append(dict(key="nee15_1x1_script", scene="nee15", W=1, H=1, first=0, n_it=1, mode=MODE_SCRIPT, seed=0,
                     script=script, draws=int(draws)))
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **frames)
    with open(os.path.join(HERE, "frames.json"), "w") as f:
        json.dump(meta, f, indent=1)

    # --- save() ------------------------------------------------------------------------
    W, H, samples = 13, 5, 7
    buf = rng.uniform(0, 1.2 * samples, size=(H, W, 3))
    buf[0, 0] = [0.0, samples * 1.0, samples * (255.0 / 255.0)]
    buf[0, 1] = [samples * 254.999999 / 255.0, samples * 0.5, samples * 2.0]
    ppm = {}
    for prog in (0, 1):
        data, fname = ref.save(prog, buf, samples, W, H)
        ppm[str(prog)] = dict(name=fname, text=data.decode())
    np.save(os.path.join(HERE, "save_buffer.npy"), buf)
    with open(os.path.join(HERE, "save.json"), "w") as f:
        json.dump(dict(W=W, H=H, samples=samples, ppm=ppm), f)

    # --- MatusikBRDF lookups on an index-valued table -----------------------------------
    # value = its own index in every channel plane, so r/RED_SCALE recovers the table index.
    table = np.concatenate([np.arange(BRDF_N, dtype=np.float64)] * 3)
    ang = np.stack([rng.uniform(0, np.pi / 2, 4096), rng.uniform(-np.pi, np.pi, 4096),
                    rng.uniform(0, np.pi / 2, 4096), rng.uniform(-np.pi, np.pi, 4096)], axis=1)
    ang[:4] = [[0.3, 0.1, 0.7, 2.0], [0, 0, 0, 0], [1.5, 3.0, 1.5, -3.0], [0.785398, 0, 0.785398, np.pi]]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "index.binary")
        with open(path, "wb") as f:
            np.array([90, 90, 180], dtype=np.int32).tofile(f)
            table.tofile(f)
        rgb = ref.brdf_lookup(path, ang)
    np.savez_compressed(os.path.join(HERE, "brdf.npz"), angles=ang, rgb=rgb)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
