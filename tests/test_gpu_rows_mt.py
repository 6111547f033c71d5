"""GPU parity of the pieces taken from the light-tracing family: the Moller-Trumbore rectangle
(light_tracing.cpp:477-575) in both pipelines, grid sampling and the row-range render of
Renderer::ray_iterate (:774-790), the row split of its main() (:1441-1451).  All through the C ABI,
bit for bit against the oracle's restatement (whose own parity is unpinned for these pieces: the
reference programs that hold them cannot be built here, they need kdtree.h)."""
import numpy as np
import pytest

import path_tracer_amd as pt
from _oracle import MODE_PIXEL
from _scenes import mt_rectangle_scene
from conftest import same_bits
from test_gpu_parity import oracle_scene

pytestmark = pytest.mark.gpu


from _scenes import mt_edge_rays as _edge_rays  # noqa: E402


@pytest.mark.parametrize("seed", range(40, 52))
def test_moller_trumbore_rectangles_match_oracle(port, seed):
    W, H = 36, 24
    d = mt_rectangle_scene(seed, W, H, n_rect=3 + seed % 8, n_sphere=seed % 6, n_plane=seed % 3, grid=bool(seed & 1),
                           warp=0.0 if seed % 3 else 0.2)
    r = pt.Renderer(d)
    s = oracle_scene(port, r, W, H, seed, flags=d.flags)
    rng = np.random.default_rng(seed)
    rays = np.concatenate([rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5], (4000, 3)),
                           rng.normal(size=(4000, 3)) * rng.uniform(0.1, 3.0, (4000, 1))], axis=1)
    rays = np.concatenate([rays, _edge_rays(d, rng)])
    t_all, idx, t = port.intersect(s, rays)
    assert same_bits(r.intersect_all(rays), t_all)
    gi, gt = r.intersect_batch(rays)
    assert np.array_equal(gi, idx) and same_bits(gt[idx >= 0], t[idx >= 0])
    assert np.count_nonzero(idx >= 0) > 0.1 * len(idx)
    if d.flags & pt.F_PIXEL_GRID:
        want = port.render_rows(s, 0, H, 0, 3)
    else:
        want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    assert same_bits(r.render(0, 3), want)
    r.close()


def test_grid_layout_is_the_pixel_grid(port):
    W, H = 37, 23
    r = pt.Renderer(pt.preset("light_room32", W, H, seed=5))
    nrows, ys, xs, rl = r.frame_layout(3)
    assert nrows == H and np.all(rl == W)
    assert same_bits(ys, np.arange(H, dtype=np.float64) / float(H))
    assert same_bits(xs, np.broadcast_to(np.arange(W, dtype=np.float64) / float(W), (H, W)))
    r.close()


@pytest.mark.parametrize("workers", [1, 5, 7])
def test_light_room_rendered_by_row_ranges(port, workers):
    """main()'s split of light_tracing.cpp: every worker renders its rows with ray_iterate; together
    they give the frame of one full-height call, and each range equals the oracle's ray_iterate."""
    W, H, n_it = 64, 48, 3
    r = pt.Renderer(pt.preset("light_room32", W, H, seed=2))
    s = oracle_scene(port, r, W, H, 2, flags=pt.F_PIXEL_GRID)
    whole = port.render_rows(s, 0, H, 0, n_it)
    assert same_bits(r.render(0, n_it), whole)
    assert np.count_nonzero(whole) > 0.9 * whole.size
    split = pt.row_split(H, workers)
    for a, n in split:
        want = port.render_rows(s, a, n, 0, n_it)
        assert np.count_nonzero(want[:a]) == 0 and np.count_nonzero(want[a + n:]) == 0
        r.iterations_done = 0
        r.buffer[:] = 0.0
        r.ray_iterate(a, n, n_it)
        assert same_bits(r.buffer, want), (a, n)
    # the workers' rows accumulated into one buffer, as the reference's workers copy theirs
    r.buffer[:] = 0.0
    for a, n in split:
        r.ray_iterate(a, n, n_it)
    assert same_bits(r.buffer, whole)
    r.close()


def test_row_ranges_in_the_wavefront_pipeline_and_with_tile_partitions(port):
    W, H, n_it = 70, 41, 2
    r = pt.Renderer(pt.preset("light_room32", W, H, seed=9))
    s = oracle_scene(port, r, W, H, 9, flags=pt.F_PIXEL_GRID)
    want = port.render_rows(s, 11, 17, 0, n_it)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    r.set_rows(11, 17)
    assert same_bits(r.render(0, n_it), want)
    r.set_pipeline(pt.PIPELINE_FUSED)
    acc = np.zeros((H, W, 3))
    for rank in range(3):
        r.set_partition(rank, 3)
        r.render(0, n_it, acc)
    assert same_bits(acc, want)
    r.set_partition(0, 1)
    r.render(0, n_it)
    assert r.stats()["paths"] == 17 * W * n_it
    with pytest.raises(pt.PtError):
        r.set_rows(30, 20)
    r.close()


def test_row_range_of_a_jittered_frame(port):
    """pt_set_rows on an ordinary (jittered) scene: the destination rows of the range equal the same
    rows of the full frame, the others stay untouched."""
    W, H, n_it = 48, 40, 4
    r = pt.Renderer(pt.preset("nee15", W, H, seed=3))
    full = r.render(0, n_it)
    r.set_rows(8, 13)
    part = r.render(0, n_it)
    assert same_bits(part[8:21], full[8:21])
    assert np.count_nonzero(part[:8]) == 0 and np.count_nonzero(part[21:]) == 0
    r.close()


def test_moller_trumbore_rectangles_match_the_reference_class():
    """pt_intersect_all on scenes made of the golden rectangles against what light_tracing.cpp's own
    Rectangle::intersect returned for the same rays (tests/golden/light_rectangle.npz)."""
    from test_light_tracing_glue import light_golden

    meta, z = light_golden()
    for m in meta:
        corners, rays, want = z[m["key"] + "_corners"], z[m["key"] + "_rays"], z[m["key"] + "_t"]
        n = corners.shape[0]
        d = mt_rectangle_scene(m["seed"], 16, 16, n_rect=3 + m["seed"] % 5, n_sphere=0, n_plane=0, warp=m["warp"])
        assert d.body_count == n
        for i in range(n):
            assert list(d.bodies[i].corners) == corners[i].tolist()  # the generator's scene, unchanged
        r = pt.Renderer(d)
        assert same_bits(r.intersect_all(rays).T, want), m["key"]
        r.close()
