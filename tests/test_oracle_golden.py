"""The C restatement (oracle/cpu_ref.c) against golden vectors produced by the reference's own
code (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

import _oracle
from _oracle import EST_NEE, EST_PT, MODE_PIXEL, MODE_SCRIPT, MODE_SEQ
from conftest import GOLDEN, same_bits

EST = {"main32": EST_PT, "nee15": EST_NEE}


def test_intersect_matches_reference(port, golden_scene):
    z = np.load(os.path.join(GOLDEN, "intersect.npz"))
    for name in ("main32", "nee15"):
        bodies, n, cam = golden_scene(name)
        s = port.scene(bodies, n, cam, 8, 8, EST[name])
        rays = z[f"{name}_rays"]
        t_all, idx, t = port.intersect(s, rays)
        assert same_bits(t_all, z[f"{name}_t_all"])  # Shape::intersect of every body, bit for bit
        # nearest = winner of (t > 0) && (t <= mint) in array order, ties to the larger index
        want = z[f"{name}_t_all"]
        for i in range(rays.shape[0]):
            best, bi = np.inf, -1
            for b in range(n):
                if want[i, b] > 0 and want[i, b] <= best:
                    best, bi = want[i, b], b
            assert idx[i] == bi
            if bi >= 0:
                assert t[i] == best
    assert np.array_equal(port.intersect(port.scene(*golden_scene("nee15"), 8, 8, EST_NEE), z["nee15_rays"])[1],
                          z["nee15_nearest_idx"])  # the reference's own scan loop


def test_camera_rays(port, golden_scene):
    z = np.load(os.path.join(GOLDEN, "intersect.npz"))
    for name in ("main32", "nee15"):
        s = port.scene(*golden_scene(name), 8, 8, EST[name])
        assert same_bits(port.get_rays(s, z["xy"]), z[f"{name}_rays"][:256])


def test_bounce_matches_reference(port, golden_scene):
    with open(os.path.join(GOLDEN, "bounce.json")) as f:
        cases = json.load(f)
    scenes = {0: port.scene(*golden_scene("main32"), 8, 8, EST_PT), 1: port.scene(*golden_scene("nee15"), 8, 8, EST_NEE)}
    assert len(cases) >= 50
    for c in cases:
        ray = [float.fromhex(v) for v in c["ray"]]
        nrm = [float.fromhex(v) for v in c["normal"]]
        out, draws = port.bounce(scenes[c["prog"]], c["body"], ray, nrm, c["script"])
        assert [float(v).hex() for v in out] == c["out"], c
        assert draws == c["draws"]


def test_frames_match_reference(port, golden_scene):
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    with open(os.path.join(GOLDEN, "frames.json")) as f:
        meta = json.load(f)
    assert len(meta) == 11
    for m in meta:
        s = port.scene(*golden_scene(m["scene"]), m["W"], m["H"], EST[m["scene"]], seed=m["seed"])
        buf, draws = port.render(s, m["mode"], m["first"], m["n_it"], script=m.get("script"))
        assert same_bits(buf, z[m["key"]]), m["key"]
        assert draws == m["draws"], m["key"]


def test_known_answer_1x1(port, golden_scene):
    # 1x1 next_event iterate() with a cyclic six-entry script: 51 draws and this pixel value
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    assert z["nee15_1x1_script"].reshape(3).tolist() == [0.64000000000000012, 0.58799999999999997, 0.92880000000000007]


def test_save_bytes(port):
    with open(os.path.join(GOLDEN, "save.json")) as f:
        g = json.load(f)
    buf = np.load(os.path.join(GOLDEN, "save_buffer.npy"))
    text = port.save_ppm(buf, g["samples"], g["W"], g["H"]).decode()
    assert text == g["ppm"]["0"]["text"] == g["ppm"]["1"]["text"]


def test_brdf_lookup(port):
    z = np.load(os.path.join(GOLDEN, "brdf.npz"))
    table = np.concatenate([np.arange(_oracle.BRDF_N, dtype=np.float64)] * 3)
    rgb, ind = port.brdf_lookup(table, z["angles"])
    assert same_bits(rgb, z["rgb"])  # same libm in this container: exact
    assert ind[:4].tolist() == [684585, 0, 1410930, 8010]


def test_plane_normal_like_ctor(port, golden_scene):
    bodies, n, cam = golden_scene("nee15")
    once = port.plane_normal([1.0, -1.0, -1.0])
    twice = port.plane_normal(once)  # main() normalises, the Plane ctor normalises again
    assert [float(v).hex() for v in twice] == [float(v).hex() for v in bodies[12].normal]


def test_frame_layout_short_rows(port, golden_scene):
    # a row is one pixel short when its jitter pushes the last x to >= 0.99999 (main.cpp:355-357)
    s = port.scene(*golden_scene("nee15"), 1024, 4, EST_NEE, seed=5)
    short = 0
    for it in range(64):
        rows, ys, xs, rl = port.frame_layout(s, it)
        assert rows in (3, 4)
        assert set(rl[:rows].tolist()) <= {1023, 1024}
        short += int((rl[:rows] == 1023).sum())
    assert 0 < short < 20  # about 1 % of rows at W = 1024


def custom_cases():
    with open(os.path.join(GOLDEN, "frames_custom.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", custom_cases(), ids=lambda c: c["key"])
def test_custom_scene_frames_match_reference(port, golden_scene, case):
    """Scenes the reference does not hard-code (ten lights, coincident and extra planes, open walls,
    1000 bodies): the restatement against frames the reference's own classes rendered from the same
    constructor arguments (tests/golden/make_golden_custom.py)."""
    z = np.load(os.path.join(GOLDEN, "frames_custom.npz"))
    args, n = _oracle.bodies_from_bytes(z[case["key"] + "_bodies"], case.get("body_record", 136))
    assert n == case["bodies"]
    stored = _oracle.constructed(port, args, n)
    _, _, cam = golden_scene("nee15")  # every program of the reference uses the same camera
    est = EST_NEE if case["prog"] == 1 else EST_PT
    s = port.scene(stored, n, cam, case["W"], case["H"], est, seed=case["seed"])
    got, draws = port.render(s, MODE_PIXEL, case["first"], case["n_it"])
    assert same_bits(got, z[case["key"] + "_frame"])
    assert draws == case["draws"]


def test_save_edge_values(port):
    with open(os.path.join(GOLDEN, "save_edge.json")) as f:
        g = json.load(f)
    buf = np.array([float.fromhex(h) for h in g["values_hex"]]).reshape(g["H"], g["W"], 3)
    assert port.save_ppm(buf, g["samples"], g["W"], g["H"]).decode() == g["ppm"]["1"]["text"]


F_RAYTRACE3 = 2


def raytrace3_golden():
    with open(os.path.join(GOLDEN, "raytrace3.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, "raytrace3.npz"))


def test_raytrace3_matches_reference(port, golden_scene):
    """raytrace3.cpp (Rectangle shape, hit threshold 1e-7, tests/golden/make_golden_raytrace3.py): the
    restatement against the reference's per-body intersect values and the frames of its iterate()."""
    meta, z = raytrace3_golden()
    bodies, n = _oracle.bodies_from_bytes(z["scene_bodies"], meta["body_record"])
    assert n == meta["bodies"] == 11
    _, _, cam = golden_scene("nee15")  # raytrace3.cpp:921-926 is the same camera
    s = port.scene(bodies, n, cam, 8, 8, EST_PT, flags=F_RAYTRACE3)
    t_all, idx, t = port.intersect(s, z["rays"])
    assert same_bits(t_all, z["t_all"])
    want = z["t_all"]
    eps = 1e-7  # t > 0.0000001L: no t of this batch lies within an ulp of the threshold
    for i in range(want.shape[0]):
        best, bi = np.inf, -1
        for b in range(n):
            if want[i, b] > eps and want[i, b] <= best:
                best, bi = want[i, b], b
        assert idx[i] == bi
    for c in meta["frames"]:
        s = port.scene(bodies, n, cam, c["W"], c["H"], EST_PT, seed=c["seed"], flags=F_RAYTRACE3)
        got, draws = port.render(s, MODE_PIXEL, c["first"], c["n_it"])
        assert same_bits(got, z["frame_" + c["key"]]), c["key"]
        assert draws == c["draws"]
    for c in meta["custom"]:
        args, nb = _oracle.bodies_from_bytes(z[c["key"] + "_bodies"], meta["body_record"])
        stored = _oracle.constructed(port, args, nb)
        s = port.scene(stored, nb, cam, c["W"], c["H"], EST_PT, seed=c["seed"], flags=F_RAYTRACE3)
        got, draws = port.render(s, MODE_PIXEL, c["first"], c["n_it"])
        assert same_bits(got, z[c["key"] + "_frame"]), c["key"]
        assert draws == c["draws"]
