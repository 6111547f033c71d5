"""CPU checks of the pieces taken from the light-tracing family (light_tracing.cpp): the row split of
its main(), the Moller-Trumbore rectangle of the oracle, and the light_room32 preset.  The rectangle
is pinned by the reference's own class: light_tracing.cpp cannot be built as a program (kdtree.h is
absent), but the head of the file that holds Rectangle compiles on its own, and
tests/golden/light_rectangle.npz holds what its intersect() returns (make_golden_light.py); closed-form
known answers stand beside it.  The row split and the grid sampling sit inside main() and the
photon-map Renderer and stay pinned by known answers only."""
import ctypes as C

import numpy as np
import pytest

import path_tracer_amd as pt
import _oracle
from _oracle import EST_PT


@pytest.mark.parametrize("H,T", [(768, 8), (48, 5), (7, 3), (3, 7), (1, 1), (100, 1), (0, 4)])
def test_row_split_as_the_reference_main_does_it(port, H, T):
    got = pt.row_split(H, T)
    assert got == port.row_split(H, T)
    g = H // T
    assert got == [(i * g, g if i < T - 1 else H - i * g) for i in range(T)]
    assert sum(n for _, n in got) == H
    rows = [r for a, n in got for r in range(a, a + n)]
    assert rows == list(range(H))  # contiguous, disjoint, in order


def test_row_split_rejects_bad_arguments():
    a, b = C.c_int(), C.c_int()
    L = pt.lib()
    assert L.pt_row_split(10, 0, 0, C.byref(a), C.byref(b)) < 0
    assert L.pt_row_split(10, 2, 2, C.byref(a), C.byref(b)) < 0
    assert L.pt_row_split(-1, 2, 0, C.byref(a), C.byref(b)) < 0
    assert L.pt_row_split(10, 2, 0, None, C.byref(b)) < 0


def _one_rectangle(port, corners):
    bodies = (_oracle.Body * 1)()
    b = bodies[0]
    b.shape = _oracle.RECTANGLE_MT
    b.material = 0
    for k, v in enumerate(np.asarray(corners, dtype=np.float64).reshape(-1)):
        b.corners[k] = float(v)
    stored = _oracle.constructed(port, bodies, 1)
    cam = _oracle.Camera()
    return port.scene(stored, 1, cam, 4, 4, EST_PT), stored


def test_moller_trumbore_rectangle_known_answers(port):
    # unit square in the plane z = 4, corners counter-clockwise seen from below
    s, stored = _one_rectangle(port, [[0, 0, 4], [1, 0, 4], [1, 1, 4], [0, 1, 4]])
    # normal = normalize((p3 - p1) x (p2 - p1)) = (0, 0, -1)
    assert list(stored[0].normal) == [0.0, 0.0, -1.0]
    up = [0.0, 0.0, 1.0]
    rays = np.array([
        [0.75, 0.25, 0.0] + up,    # inside the first triangle (p1, p2, p3)
        [0.25, 0.75, 0.0] + up,    # inside the second triangle (p1, p3, p4)
        [0.5, 0.5, 0.0] + up,      # on the shared diagonal: u + v == 1 passes the first triangle
        [0.0, 0.0, 0.0] + up,      # corner p1
        [1.0, 1.0, 0.0] + up,      # corner p3
        [1.25, 0.5, 0.0] + up,     # outside
        [0.5, -0.25, 0.0] + up,    # outside
        [0.75, 0.25, 8.0] + [0.0, 0.0, -1.0],  # from the other side: no back-face culling
        [0.75, 0.25, 8.0] + up,    # pointing away: the plane is behind, t < 0 (first triangle answers)
        [0.5, 0.5, 4.0] + [1.0, 0.0, 0.0],     # in the plane: |det| < EPSILON
        [0.75, 0.25, 0.0] + [0.0, 0.0, 2.0],   # direction of length 2: t halves
    ])
    t_all, idx, t = port.intersect(s, rays)
    want = [4.0, 4.0, 4.0, 4.0, 4.0, -1.0, -1.0, 4.0, -4.0, -1.0, 2.0]
    assert t_all[:, 0].tolist() == want
    assert idx.tolist() == [0, 0, 0, 0, 0, -1, -1, 0, -1, -1, 0]


def test_first_triangle_answer_of_minus_one_falls_through(port):
    """tri_intersect2 returning exactly -1.0 means "miss" to Rectangle::intersect
    (light_tracing.cpp:511), also when it is a genuine distance: the second triangle is then asked."""
    s, _ = _one_rectangle(port, [[0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
    # from z = 2 looking up: the plane z = 1 lies at t = -1 exactly
    rays = np.array([[0.75, 0.25, 2.0, 0.0, 0.0, 1.0], [0.25, 0.75, 2.0, 0.0, 0.0, 1.0]])
    t_all, _, _ = port.intersect(s, rays)
    # first ray: inside triangle one, t1 == -1.0 -> falls through, triangle two misses (u + v > 1) -> -1
    # second ray: triangle one misses, triangle two returns its genuine -1.0
    assert t_all[:, 0].tolist() == [-1.0, -1.0]


def test_light_room32_preset():
    d = pt.preset("light_room32", 32, 24)
    assert d.body_count == 32 and d.estimator == pt.EST_PT and d.flags == pt.F_PIXEL_GRID
    shapes = [d.bodies[i].shape for i in range(32)]
    assert shapes.count(pt.RECTANGLE_MT) == 8 and shapes.count(pt.SPHERE) == 24
    assert [i for i, s in enumerate(shapes) if s == pt.RECTANGLE_MT] == [6, 7, 8, 9, 10, 11, 12, 31]
    # the 18 ball lights share colour == emission (light_tracing.cpp:1337)
    balls = [d.bodies[i] for i in range(13, 31)]
    assert all(list(b.color) == [0.3, 0.47, 0.39] == list(b.emission) and b.radius == 0.1 for b in balls)
    b = pt.Body.from_buffer_copy(d.bodies[6])  # floor: normal +z after construction
    assert pt.lib().pt_body_construct(C.byref(b)) == 0
    assert list(b.normal) == [0.0, 0.0, 1.0]


def light_golden():
    import json
    import os

    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "light_rectangle.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, "light_rectangle.npz"))


def test_moller_trumbore_rectangle_matches_the_reference_class(port):
    """Rectangle::intersect and the constructor's normal of light_tracing.cpp:477-575, on planar and
    warped rectangles, random rays and rays aimed at corners, edges and the shared diagonal: the
    restatement gives the reference's values bit for bit."""
    from conftest import same_bits

    meta, z = light_golden()
    assert sum(m["positive"] for m in meta) > 3000
    for m in meta:
        corners, rays, want, normals = (z[m["key"] + "_" + k] for k in ("corners", "rays", "t", "normals"))
        n = corners.shape[0]
        bodies = (_oracle.Body * n)()
        for i in range(n):
            bodies[i].shape = _oracle.RECTANGLE_MT
            bodies[i].material = 0
            for k in range(12):
                bodies[i].corners[k] = float(corners[i, k])
        stored = _oracle.constructed(port, bodies, n)
        for i in range(n):
            assert same_bits(np.array(list(stored[i].normal)), normals[i]), (m["key"], i)
        s = port.scene(stored, n, _oracle.Camera(), 4, 4, EST_PT)
        t_all, _, _ = port.intersect(s, rays)
        assert same_bits(t_all.T, want), m["key"]
