"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle (oracle/cpu_ref.c)
and the committed golden vectors.  Integer/index results and FP64 radiance are compared bit for bit."""
import json
import os

import numpy as np
import pytest

import path_tracer_amd as pt
from _oracle import EST_NEE, EST_PT, MODE_PIXEL
from conftest import GOLDEN, same_bits
from _scenes import open_scene, rectangle_scene, scene_with_extra_planes

pytestmark = pytest.mark.gpu
EST = {"main32": EST_PT, "nee15": EST_NEE, "pt15": EST_PT, "nee15_lit": EST_NEE, "bipath23": EST_NEE, "nee32": EST_NEE}


def oracle_scene(port, renderer, W, H, seed, flags=0, tables=None):
    bodies, n = renderer.bodies()  # as stored by the product (normals normalised)
    import _oracle

    ob = (_oracle.Body * n)()
    for i in range(n):
        for f, _ in _oracle.Body._fields_:
            src = getattr(bodies[i], "reserved" if f == "pad" else f)
            setattr(ob[i], f, src)
    cam = _oracle.Camera()
    for f in ("origin", "topleft", "topright", "bottomleft"):
        for k in range(3):
            getattr(cam, f)[k] = getattr(renderer.desc.camera, f)[k]
    return port.scene(ob, n, cam, W, H, renderer.desc.estimator, seed=seed, flags=flags, tables=tables)


def test_rng_contract(port):
    for (seed, it, dom, idx) in ((1, 0, 0, 0), (7, 3, 0, 123456), (2**63 + 5, 7199, 1, 767), (0, 0, 2, 0), (9, 1, 3, 4)):
        k, u = pt.rng_fill(seed, it, dom, idx, 257)
        k0, u0 = port.rng_fill(seed, it, dom, idx, 257)
        assert np.array_equal(k, k0)
        assert same_bits(u, u0)


def test_intersect_bit_exact_on_fixed_ray_batch(port):
    z = np.load(os.path.join(GOLDEN, "intersect.npz"))
    for name in ("main32", "nee15"):
        r = pt.Renderer(pt.preset(name, 16, 16))
        rays = z[f"{name}_rays"]
        assert same_bits(r.intersect_all(rays), z[f"{name}_t_all"])  # vs the reference's Shape::intersect
        idx, t = r.intersect_batch(rays)
        s = oracle_scene(port, r, 16, 16, 1)
        _, idx0, t0 = port.intersect(s, rays)
        assert np.array_equal(idx, idx0) and same_bits(t, t0)
        if name == "nee15":
            assert np.array_equal(idx, z["nee15_nearest_idx"])  # the reference's own scan loop
            idx2, t2 = r.intersect_batch(rays, skip=14)
            _, idx3, t3 = port.intersect(s, rays, skip=14)
            assert np.array_equal(idx2, idx3) and same_bits(t2, t3)
        r.close()


def test_intersect_large_random_batch(port):
    rng = np.random.default_rng(5)
    n = 200_000
    o = np.stack([rng.uniform(-1.8, 1.8, n), rng.uniform(-2.4, 4.4, n), rng.uniform(-0.45, 2.4, n)], axis=1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.uniform(0.01, 1.0, size=(n, 1))
    rays = np.concatenate([o, d], axis=1)
    for name in ("main32", "nee15"):
        r = pt.Renderer(pt.preset(name, 16, 16))
        idx, t = r.intersect_batch(rays)
        _, idx0, t0 = port.intersect(oracle_scene(port, r, 16, 16, 1), rays)
        assert np.array_equal(idx, idx0) and same_bits(t, t0)
        r.close()


@pytest.mark.parametrize("W,H,seed", [(24, 16, 7), (1024, 5, 5), (3, 700, 11), (1, 1, 3)])
def test_frame_layout(port, W, H, seed):
    r = pt.Renderer(pt.preset("nee15", W, H, seed=seed))
    s = oracle_scene(port, r, W, H, seed)
    for it in (0, 1, 17, 4099):
        nrows, ys, xs, rl = r.frame_layout(it)
        rows0, ys0, xs0, rl0 = port.frame_layout(s, it)
        assert np.array_equal(rl, rl0)
        assert same_bits(ys[:rows0], ys0[:rows0])
        for row in range(rows0):
            assert same_bits(xs[row, : rl0[row]], xs0[row, : rl0[row]])
    r.close()


@pytest.mark.parametrize("name", ["nee15", "main32", "pt15", "nee15_lit", "bipath23", "nee32"])
def test_paths_match_oracle(port, name):
    rng = np.random.default_rng(11)
    n = 20000
    r = pt.Renderer(pt.preset(name, 64, 64, seed=99))
    s = oracle_scene(port, r, 64, 64, 99)
    it_idx = np.stack([rng.integers(0, 7200, n), rng.integers(0, 1 << 20, n)], axis=1).astype(np.int32)
    xy = rng.uniform(0, 1, size=(n, 2))
    g = r.trace_paths(it_idx, xy)
    o = port.trace_paths(s, it_idx, xy)
    assert np.array_equal(g["nrec"], o["nrec"])
    assert np.array_equal(g["bodies"], o["bodies"])
    assert np.array_equal(g["draws"], o["draws"])
    if EST[name] == EST_NEE:
        assert np.array_equal(g["visible"], o["visible"])
    # offset points: the x87 islands of the reference (1.0L/s, t*1.0000001L) are emulated exactly
    assert same_bits(g["points"], o["points"])
    assert same_bits(g["rgb"], o["rgb"])  # radiance is a function of the discrete path only here
    r.close()


@pytest.mark.parametrize("name", ["nee15", "bipath23", "main32", "nee15_lit"])
def test_render_kernel_equals_path_diagnostics(name):
    """The kernel variants that render frames (compact next-event form, records without positions for
    path tracing) against pt_trace_paths, which runs the plain per-path form on the same samples: one
    iteration's frame equals the per-path radiance, pixel for pixel (as numbers: the compact form may
    give a zero the other sign)."""
    W, H, it = 45, 19, 5
    r = pt.Renderer(pt.preset(name, W, H, seed=17))
    frame = r.render(it, 1).reshape(-1, 3)
    nrows, ys, xs, rl = r.frame_layout(it)
    idx, xy = [], []
    for row in range(nrows):
        for col in range(rl[row]):
            idx.append(len(idx))
            xy.append((xs[row, col], ys[row]))
    n = len(idx)
    it_idx = np.stack([np.full(n, it), np.array(idx)], axis=1).astype(np.int32)
    g = r.trace_paths(it_idx, np.array(xy))
    assert np.array_equal(frame[:n], g["rgb"])
    assert np.count_nonzero(frame[n:]) == 0
    r.close()


def test_golden_frames(port):
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    with open(os.path.join(GOLDEN, "frames.json")) as f:
        meta = [m for m in json.load(f) if m["mode"] == MODE_PIXEL]
    assert len(meta) == 8
    for m in meta:
        r = pt.Renderer(pt.preset(m["scene"], m["W"], m["H"], seed=m["seed"]))
        buf = r.render(m["first"], m["n_it"])
        assert same_bits(buf, z[m["key"]]), m["key"]  # vs the reference's own Renderer::iterate
        r.close()


@pytest.mark.parametrize("name,W,H,n_it", [("nee15", 96, 64, 6), ("main32", 80, 48, 5), ("nee15", 33, 7, 40),
                                           ("nee15_lit", 64, 40, 5), ("pt15", 40, 40, 4), ("bipath23", 72, 40, 4), ("nee32", 50, 30, 4),
                                           # the headline frame size: sixteen iterations of it (20 s of oracle time)
                                           ("nee15", 1024, 768, 16), ("main32", 1024, 768, 3)])
def test_frames_match_oracle(port, name, W, H, n_it):
    r = pt.Renderer(pt.preset(name, W, H, seed=2025))
    s = oracle_scene(port, r, W, H, 2025)
    want, _ = port.render(s, MODE_PIXEL, 3, n_it)
    got = r.render(3, n_it)
    assert same_bits(got, want)
    # the result does not depend on how the iterations are split into calls
    a = r.render(3, 2)
    r.render(5, n_it - 2, a)
    assert same_bits(a, want)
    r.close()


def test_short_row_shift_at_full_width(port):
    # W = 1024: about 1 % of rows are one pixel short and shift every later pixel (main.cpp:350,362)
    W, H = 1024, 6
    r = pt.Renderer(pt.preset("nee15", W, H, seed=5))
    s = oracle_scene(port, r, W, H, 5)
    short_its = [it for it in range(64) if (port.frame_layout(s, it)[3][: port.frame_layout(s, it)[0]] == W - 1).any()]
    assert short_its
    it = short_its[0]
    want, _ = port.render(s, MODE_PIXEL, it, 1)
    assert same_bits(r.render(it, 1), want)
    r.close()


def test_partition_tiles_sum_to_whole():
    W, H = 100, 37
    whole = pt.Renderer(pt.preset("nee15", W, H, seed=4)).render(0, 3)
    acc = np.zeros_like(whole)
    for world in (2, 3):
        acc[:] = 0
        owned = np.zeros((H, W), dtype=np.int32)
        for rank in range(world):
            r = pt.Renderer(pt.preset("nee15", W, H, seed=4))
            r.set_partition(rank, world, 32, 8)
            part = r.render(0, 3)
            from path_tracer_amd import distributed as ptdist

            assert not (part != 0).any(axis=2)[~ptdist.owned_mask(W, H, rank, world)].any()  # host and device agree on ownership
            # the path count a handle reports is that of the pixels it owns (100 is no multiple of the tile width)
            assert r.stats()["paths"] == 3 * int(ptdist.owned_mask(W, H, rank, world).sum())
            owned += (part != 0).any(axis=2)
            acc += part
            r.close()
        assert same_bits(acc, whole)
        assert owned.max() == 1


def test_partition_tiles_sum_to_whole_large_scene():
    """The same for the wavefront pipeline (1 000 bodies), on one handle whose partition changes
    between renders."""
    W, H, n_it = 70, 40, 4
    r = pt.Renderer(pt.preset("synthetic1k", W, H, seed=4))
    whole = r.render(0, n_it)
    assert np.count_nonzero(whole) > 0.3 * whole.size
    for world in (2, 5):
        acc = np.zeros_like(whole)
        for rank in range(world):
            r.set_partition(rank, world)
            acc += r.render(0, n_it)
        assert same_bits(acc, whole)
    r.set_partition(0, 1)
    assert same_bits(r.render(0, n_it), whole)
    r.close()


def test_true_light_centre_flag(port):
    d = pt.preset("nee15", 48, 32, seed=8)
    d.flags = pt.F_LIGHT_TRUE_CENTER
    r = pt.Renderer(d)
    s = oracle_scene(port, r, 48, 32, 8, flags=1)
    want, _ = port.render(s, MODE_PIXEL, 0, 4)
    assert same_bits(r.render(0, 4), want)
    r.close()


@pytest.mark.parametrize("flags", [0, 1], ids=["shadow_origin_0", "true_light_centre"])
def test_absorbing_emitters_in_shadow(port, flags):
    """The compact next-event kernel scans the vertices deeper than the shallowest lit emitter of colour 0
    only when that vertex turns out to be in shadow (a second pass of its shadow rounds).  In the
    reference's room that is one vertex in seven hundred; here diffuse balls around the shadow-ray
    origin put many of them in shadow, and a chrome and a glass ball beside them add twin vertices."""
    W, H, seed, n_it = 64, 48, 17, 4
    d = pt.preset("nee15", W, H, seed=seed)
    n = d.body_count
    extra = [(0.0, 0.0, 0.9, 0.45, pt.DIFFUSE), (0.5, 0.9, 0.3, 0.35, pt.DIFFUSE), (-0.6, 0.4, 0.2, 0.3, pt.CHROME),
             (0.1, -0.7, 0.1, 0.3, pt.GLASS)]
    bodies = (pt.Body * (n + len(extra)))()
    for i in range(n):
        bodies[i] = d.bodies[i]
    for k, (x, y, z, rad, material) in enumerate(extra):
        b = pt.Body.from_buffer_copy(d.bodies[6])  # a grey diffuse ball of the room as the template
        b.shape = pt.SPHERE
        b.center[0], b.center[1], b.center[2], b.radius = x, y, z, rad
        b.material = material
        b.ior, b.reflection = 1.5, 0.1
        bodies[n + k] = b
    d.bodies, d.body_count, d._keep = bodies, n + len(extra), bodies
    d.flags = pt.F_LIGHT_TRUE_CENTER if flags else 0
    r = pt.Renderer(d)
    s = oracle_scene(port, r, W, H, seed, flags=flags)
    want, _ = port.render(s, MODE_PIXEL, 0, n_it)
    assert same_bits(r.render(0, n_it), want)
    # the case is really there: paths whose shallowest emitter vertex is in shadow and has an emitter below it
    rng = np.random.default_rng(3)
    m = 20000
    it_idx = np.stack([rng.integers(0, 64, m), rng.integers(0, W * H, m)], axis=1).astype(np.int32)
    g = r.trace_paths(it_idx, rng.uniform(0, 1, size=(m, 2)))
    emit = [i for i in range(d.body_count) if max(bodies[i].emission) != 0.0]
    cases = 0
    for bod, k, vis in zip(g["bodies"], g["nrec"], g["visible"]):
        on = [j for j in range(k) if bod[j] in emit]
        cases += len(on) >= 2 and not (vis >> on[0]) & 1
    assert cases > 10, cases
    r.close()


def test_measured_brdf_lookup_and_render(port):
    z = np.load(os.path.join(GOLDEN, "brdf.npz"))
    d = pt.preset("nee15_measured", 40, 30, seed=6)
    r = pt.Renderer(d)
    index_table = np.concatenate([np.arange(pt.BRDF_TABLE_DOUBLES // 3, dtype=np.float64)] * 3)
    r.brdf_upload(0, index_table)
    rgb, ind = r.brdf_lookup(0, z["angles"])
    want_ind = np.rint(z["rgb"][:, 0] * 1500.0).astype(np.int64)
    # device libm and glibc round sin/cos/acos/atan2 differently in the last place: an angle on a
    # bin edge may land in the neighbouring bin
    bad = np.nonzero(ind != want_ind)[0]
    assert len(bad) <= 2, bad
    ok = ind == want_ind
    assert same_bits(rgb[ok], z["rgb"][ok])
    assert ind[:4].tolist() == [684585, 0, 1410930, 8010]
    # Rendering.  When a path re-hits the floor at t ~ 1e-14 the outgoing direction sits exactly on
    # the pole of the half/difference parametrisation; phi_diff is then rounding noise of libm's
    # sin/cos/atan2, and glibc and the device library fall on opposite sides of the 0/pi cut (cyclic
    # neighbour bins 0 and 179).  With hash-noise tables such a lookup changes the sample, so a
    # small fraction of pixels may differ; everything else is bit-exact.
    tables = [pt.synthetic_brdf_table(k) for k in range(4)]
    for k, t in enumerate(tables):
        r.brdf_upload(k, t)
    s = oracle_scene(port, r, 40, 30, 6, tables=tables)
    want, _ = port.render(s, MODE_PIXEL, 0, 4)
    got = r.render(0, 4)
    differs = (got.view(np.uint64) != want.view(np.uint64)).any(axis=2)
    assert differs.mean() < 0.01
    assert np.abs(got).sum() > 0
    # per path: same bodies, same visibility, same draw count everywhere; radiance equal except on
    # paths with such a degenerate re-hit
    rng = np.random.default_rng(12)
    n = 20000
    it_idx = np.stack([rng.integers(0, 7200, n), rng.integers(0, 1 << 20, n)], axis=1).astype(np.int32)
    xy = rng.uniform(0, 1, size=(n, 2))
    g = r.trace_paths(it_idx, xy)
    o = port.trace_paths(s, it_idx, xy)
    assert np.array_equal(g["bodies"], o["bodies"]) and np.array_equal(g["visible"], o["visible"])
    assert np.array_equal(g["draws"], o["draws"])
    bad = (g["rgb"].view(np.uint64) != o["rgb"].view(np.uint64)).any(axis=1)
    assert bad.mean() < 2e-3
    rehit = (g["bodies"][:, 1:] == g["bodies"][:, :-1]).any(axis=1)
    assert rehit[bad].all()
    r.close()


def test_measured_render_within_tolerance_on_smooth_tables(port):
    """Configuration with measured materials, on tables that vary smoothly with the table angles (as
    real measured data does): the image must match the oracle within the stated tolerance, per-pixel
    RMSE <= 1e-3 relative, and every lookup must use the oracle's table bin or — where phi_diff is
    rounding noise of the sin/cos/atan2 in use — its neighbour on the cyclic phi_diff axis."""
    W, H, n_it = 64, 48, 8
    r = pt.Renderer(pt.preset("nee15_measured", W, H, seed=6))
    tables = [pt.smooth_brdf_table(k) for k in range(4)]
    for k, t in enumerate(tables):
        r.brdf_upload(k, t)
    s = oracle_scene(port, r, W, H, 6, tables=tables)
    want, _ = port.render(s, MODE_PIXEL, 0, n_it)
    got = r.render(0, n_it)
    rmse = np.sqrt(np.mean((got - want) ** 2)) / np.mean(want)
    assert rmse <= 1e-3, rmse
    worst = np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-3))
    assert worst <= 2e-2, worst  # no single pixel off by more than a neighbouring bin can explain
    differs = (got.view(np.uint64) != want.view(np.uint64)).any(axis=2)
    assert differs.mean() < 0.02
    # lookup by lookup
    rng = np.random.default_rng(13)
    n = 30000
    it_idx = np.stack([rng.integers(0, 7200, n), rng.integers(0, 1 << 20, n)], axis=1).astype(np.int32)
    xy = rng.uniform(0, 1, size=(n, 2))
    gi = r.trace_brdf_indices(it_idx, xy)
    oi = port.trace_brdf_indices(s, it_idx, xy)
    assert np.array_equal(gi < 0, oi < 0)  # the same vertices are lit and measured
    assert (gi >= 0).sum() > n  # the test has lookups to compare
    bad = np.argwhere(gi != oi)
    assert len(bad) < 2e-3 * (gi >= 0).sum(), len(bad)
    for k, j in bad:
        g, o = int(gi[k, j]), int(oi[k, j])
        assert g // 180 == o // 180, (k, j, g, o)  # same theta_half and theta_diff bins
        dphi = abs(g % 180 - o % 180)
        assert dphi in (1, 179), (k, j, g, o)  # neighbours on the cyclic phi_diff axis
    r.close()


def test_measured_index_from_vectors(port):
    """The estimator's trigonometry-free table index (from the two direction vectors) against the
    reference's chain vector_to_spherical -> lookup_brdf_val evaluated with libm on the host."""
    r = pt.Renderer(pt.preset("nee15_measured", 8, 8, seed=1))
    index_table = np.concatenate([np.arange(pt.BRDF_TABLE_DOUBLES // 3, dtype=np.float64)] * 3)
    r.brdf_upload(0, index_table)
    rng = np.random.default_rng(77)
    n = 400000
    v = rng.normal(size=(n, 2, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True)
    # special directions: axes, coordinate planes, equal and opposite pairs, poles of the parametrisation
    ax = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0.6, 0.8, 0], [0, 0.6, -0.8],
                   [-0.8, 0, 0.6], [0.0, 0.0, 0.0]], dtype=np.float64)
    pairs = np.array([[a, b] for a in ax for b in ax])
    v[: len(pairs)] = pairs
    v[len(pairs): len(pairs) + 1000, 1] = v[len(pairs): len(pairs) + 1000, 0]            # in == out
    v[len(pairs) + 1000: len(pairs) + 2000, 1] = -v[len(pairs) + 1000: len(pairs) + 2000, 0]  # opposite
    v[len(pairs) + 2000: len(pairs) + 3000, :, 0] = 0.0                                   # x == 0
    v[len(pairs) + 3000: len(pairs) + 4000, :, 1] = 0.0                                   # y == 0
    a_in = port.vector_to_spherical(v[:, 0])
    a_out = port.vector_to_spherical(v[:, 1])
    angles = np.concatenate([a_in, a_out], axis=1)
    _, want = port.brdf_lookup(index_table, angles)
    rgb, got = r.brdf_lookup_vectors(0, v.reshape(n, 6))
    assert np.array_equal(np.rint(rgb[:, 0] * 1500.0).astype(np.int64), got)
    bad = np.nonzero(got != want)[0]
    # Generic directions: equal indices (an angle within a few ulps of a bin edge would be the
    # exception).  Special directions sit exactly on bin edges or on the pole of the half/difference
    # parametrisation, where fi_diff is rounding noise of whichever sin/cos/atan2 is used.
    generic = bad[bad >= len(pairs) + 4000]
    assert len(generic) <= 2, (len(generic), generic[:10])
    special = bad[bad < len(pairs) + 4000]
    for i in special:
        gi, wi = int(got[i]), int(want[i])
        same_half = gi // (180 * 90) == wi // (180 * 90)
        near = abs(gi - wi) in (1, 179, 180, 180 * 90) or same_half
        assert near, (i, v[i], gi, wi)
    r.close()


def test_errors_are_reported_not_fatal():
    d = pt.preset("nee15", 8, 8)
    d.body_count = 0
    with pytest.raises(pt.PtError):
        pt.Renderer(d)
    r = pt.Renderer(pt.preset("nee15_measured", 8, 8))
    with pytest.raises(pt.PtError):  # tables not uploaded
        r.render(0, 1)
    with pytest.raises(pt.PtError):
        r.set_partition(0, 1, 10, 10)  # tile must hold 256 pixels
    r.close()


@pytest.mark.parametrize("name,W,H,n_it", [("nee15", 70, 40, 4), ("main32", 48, 48, 3), ("nee15_lit", 64, 24, 3),
                                           ("bipath23", 40, 30, 3)])
def test_wavefront_equals_fused_and_oracle(port, name, W, H, n_it):
    r = pt.Renderer(pt.preset(name, W, H, seed=31))
    want, _ = port.render(oracle_scene(port, r, W, H, 31), MODE_PIXEL, 2, n_it)
    fused = r.render(2, n_it)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    wave = r.render(2, n_it)
    assert same_bits(fused, want) and same_bits(wave, want)
    r.close()


@pytest.mark.parametrize("W,H,n_it,seed", [(70, 40, 3, 31), (33, 17, 5, 2), (64, 8, 2, 9)])
def test_measured_scene_fused_equals_wavefront(W, H, n_it, seed):
    """Measured materials in both pipelines: the fused kernel and the wavefront stages evaluate the
    lit vertices with the same device code and must give the same bits, with the hash tables (any
    misplaced lookup shows) and with partial waves."""
    r = pt.Renderer(pt.preset("nee15_measured", W, H, seed=seed))
    for k in range(4):
        r.brdf_upload(k, pt.synthetic_brdf_table(k))
    fused = r.render(1, n_it)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    wave = r.render(1, n_it)
    assert same_bits(fused, wave)
    assert np.count_nonzero(fused) > 0.9 * fused.size
    r.close()


@pytest.mark.parametrize("name", ["nee15", "main32"])
def test_open_scene_paths_terminate(port, name):
    # only the floor plane and the spheres are kept (every other plane is infinite and would close the
    # scene again): upward rays escape, queues shrink
    W, H = 64, 48
    r = pt.Renderer(open_scene(name, W, H, 17))
    s = oracle_scene(port, r, W, H, 17)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    assert same_bits(r.render(0, 3), want)
    rng = np.random.default_rng(5)
    n = 4000
    it_idx = np.stack([rng.integers(0, 100, n), rng.integers(0, W * H, n)], axis=1).astype(np.int32)
    xy = rng.uniform(0, 1, size=(n, 2))
    r.set_pipeline(pt.PIPELINE_FUSED)
    g = r.trace_paths(it_idx, xy)
    o = port.trace_paths(s, it_idx, xy)
    assert np.array_equal(g["nrec"], o["nrec"]) and (g["nrec"] < 5).mean() > 0.3
    assert np.array_equal(g["bodies"], o["bodies"]) and same_bits(g["rgb"], o["rgb"])
    r.close()


def test_large_scene_uses_grouped_scan(port):
    # 1000 and 10000 bodies: beyond the fused kernel's limit, the wavefront pipeline is selected
    for name, W, H, n_it in (("synthetic1k", 24, 16, 2), ("synthetic10k", 10, 6, 1)):
        r = pt.Renderer(pt.preset(name, W, H, seed=9))
        s = oracle_scene(port, r, W, H, 9)
        want, _ = port.render(s, MODE_PIXEL, 0, n_it)
        assert same_bits(r.render(0, n_it), want), name
        rng = np.random.default_rng(3)
        rays = np.concatenate([np.stack([rng.uniform(-1.8, 1.8, 500), rng.uniform(-2.4, 4.4, 500), rng.uniform(-0.45, 2.4, 500)], axis=1),
                               rng.normal(size=(500, 3))], axis=1)
        idx, t = r.intersect_batch(rays)
        _, idx0, t0 = port.intersect(s, rays)
        assert np.array_equal(idx, idx0) and same_bits(t, t0)
        with pytest.raises(pt.PtError):
            r.set_pipeline(pt.PIPELINE_FUSED)
            r.render(0, 1)
        r.close()


def test_ten_thousand_bodies_across_many_blocks(port):
    """The 10 000-body scene at a size where every stage of the wavefront pipeline spans many blocks
    (64x48x2 = 6144 paths, 24 blocks of 256): ray sorting, queue compaction and the grouped scan
    against the oracle, bit for bit (about ten seconds of CPU)."""
    W, H, n_it = 64, 48, 2
    r = pt.Renderer(pt.preset("synthetic10k", W, H, seed=4))
    s = oracle_scene(port, r, W, H, 4)
    want, _ = port.render(s, MODE_PIXEL, 0, n_it)
    got = r.render(0, n_it)
    assert same_bits(got, want)
    assert np.count_nonzero(got) > 0.3 * got.size
    r.close()


def test_large_random_scenes_with_many_lights(port):
    """Random scenes of several hundred to a few thousand bodies under next-event estimation: a tenth
    of the spheres are lights (more than a hundred), shadow rays start at the lights' centres in every
    other scene, glass and chrome among the bodies.  Wavefront pipeline against the oracle, bit for bit."""
    W, H = 48, 30
    for seed in (6015, 6017, 6011):
        d = random_scene(seed, W, H, 3000)
        d.flags = seed % 2
        assert d.estimator == pt.EST_NEE and d.body_count > 127
        r = pt.Renderer(d)
        s = oracle_scene(port, r, W, H, seed, flags=d.flags)
        want, _ = port.render(s, MODE_PIXEL, 0, 3)
        assert same_bits(r.render(0, 3), want), seed
        with pytest.raises(pt.PtError):  # the per-path diagnostics are the fused estimator's
            r.trace_paths(np.zeros((1, 2), dtype=np.int32), np.zeros((1, 2)))
        r.close()


def test_forty_thousand_bodies(port):
    """Scenes beyond the 64 section halves the visiting-order table covers (35 000 and 38 000
    spheres, 3 500 lights): index order at the top of the hierarchy, both estimators, a frame small
    enough for the oracle."""
    W, H = 12, 8
    for seed in (8002, 8001):
        d = random_scene(seed, W, H, 40000)
        d.flags = seed % 2
        assert d.body_count > 33000
        r = pt.Renderer(d)
        s = oracle_scene(port, r, W, H, seed, flags=d.flags)
        want, _ = port.render(s, MODE_PIXEL, 0, 2)
        assert same_bits(r.render(0, 2), want), seed
        r.close()


def test_visiting_order_of_the_scan_is_invisible(monkeypatch):
    """The grouped scan visits the top of its hierarchy nearest first from a table built at upload;
    scenes too large for the table are visited in index order.  PTGPU_INDEX_ORDER selects that path
    for any scene: same frame and same nearest hits, bit for bit."""
    W, H, n_it = 48, 32, 2
    rng = np.random.default_rng(11)
    rays = np.concatenate([np.stack([rng.uniform(-1.8, 1.8, 800), rng.uniform(-2.4, 4.4, 800), rng.uniform(-0.45, 2.4, 800)], axis=1),
                           rng.normal(size=(800, 3))], axis=1)
    out = []
    for index_order in (False, True):
        if index_order:
            monkeypatch.setenv("PTGPU_INDEX_ORDER", "1")
        r = pt.Renderer(pt.preset("synthetic10k", W, H, seed=6))
        out.append((r.render(0, n_it),) + tuple(r.intersect_batch(rays)))
        r.close()
    assert same_bits(out[0][0], out[1][0]) and np.count_nonzero(out[0][0]) > 0.3 * out[0][0].size
    assert np.array_equal(out[0][1], out[1][1]) and same_bits(out[0][2], out[1][2])
    # Likewise the ray sort: only the leading part of a queue is sorted (as much as an earlier batch
    # filled), the rest is scanned in queue order.  With a fixed part sorted, or nothing at all:
    for fraction in ("0.4", "0"):
        monkeypatch.setenv("PTGPU_SORT_FRACTION", fraction)
        r = pt.Renderer(pt.preset("synthetic10k", W, H, seed=6))
        assert same_bits(r.render(0, n_it), out[0][0]), fraction
        r.close()


def test_drop_in_executables_write_the_reference_ppm(port, tmp_path):
    """`next_event W H iters threads` and `main ...`: PPM bytes equal save() of the oracle's buffer, with
    the reference's worker arithmetic (iters/threads frames per worker, divide by the requested total)."""
    import subprocess

    from conftest import ROOT

    for prog, preset, (W, H, iters, threads) in (("next_event", "nee15", (40, 24, 7, 2)), ("main", "main32", (32, 32, 4, 1)),
                                                 ("raytrace3", "raytrace3", (36, 20, 6, 2))):
        out = str(tmp_path / f"{prog}.ppm")
        exe = os.path.join(ROOT, "path_tracer_amd", "bin", prog)
        subprocess.run([exe, str(W), str(H), str(iters), str(threads), "--seed", "77", "--out", out, "--quiet"],
                       check=True, timeout=120)
        r = pt.Renderer(pt.preset(preset, W, H, seed=77))
        s = oracle_scene(port, r, W, H, 77, flags=r.desc.flags)
        per = iters // threads
        total = np.zeros((H, W, 3))
        for k in range(threads):
            priv, _ = port.render(s, MODE_PIXEL, k * per, per)
            total = priv if threads == 1 else total + priv
        with open(out, "rb") as f:
            assert f.read() == port.save_ppm(total, iters, W, H)
        r.close()


def test_drop_in_executable_on_a_node(port, tmp_path):
    """`next_event W H iters threads --gpus 1`: the same render through libptnode.so — per-device
    scenes, tile partition, RCCL gather of the owned tiles — gives the same PPM bytes."""
    import subprocess

    from conftest import ROOT

    W, H, iters, threads = 70, 37, 6, 2  # partial edge tiles in both directions
    exe = os.path.join(ROOT, "path_tracer_amd", "bin", "next_event")
    outs = []
    for extra in ([], ["--gpus", "1"]):
        out = str(tmp_path / f"node{len(extra)}.ppm")
        subprocess.run([exe, str(W), str(H), str(iters), str(threads), "--seed", "9", "--out", out, "--quiet"] + extra,
                       check=True, timeout=120)
        with open(out, "rb") as f:
            outs.append(f.read())
    assert outs[0] == outs[1]
    r = pt.Renderer(pt.preset("nee15", W, H, seed=9))
    s = oracle_scene(port, r, W, H, 9)
    per = iters // threads
    total = np.zeros((H, W, 3))
    for k in range(threads):
        priv, _ = port.render(s, MODE_PIXEL, k * per, per)
        total = total + priv
    assert outs[1] == port.save_ppm(total, iters, W, H)
    r.close()
    bad = subprocess.run([exe, "16", "16", "1", "1", "--gpus", "99", "--quiet", "--out", str(tmp_path / "x.ppm")],
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "does not exist" in bad.stderr


def edge_rays(rng, n):
    """Origins inside, outside, exactly on and astronomically far from the walls; directions with
    exact zeros, negative zeros and components below the parallel threshold."""
    o = np.stack([rng.uniform(-6, 6, n), rng.uniform(-8, 9, n), rng.uniform(-3, 6, n)], axis=1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = np.arange(n)
    walls = [(0, -1.9), (0, 1.9), (0, 0.5), (1, 4.5), (1, -2.5), (2, -0.5), (2, 2.5), (2, 3.0)]
    for j, (axis, c) in enumerate(walls):
        o[k % 23 == j, axis] = c  # origin exactly on a plane
    o[k % 23 == 9] *= 1e15
    o[k % 23 == 10, 0] = 1e22
    o[k % 23 == 11, 2] = -3e300
    d[k % 19 == 0, 0] = 0.0
    d[k % 19 == 1, 1] = -0.0
    d[k % 19 == 2, 2] = 3e-6  # inside the +-1e-5 parallel band
    d[k % 19 == 3, 0] = -9.99e-6
    d[k % 19 == 4] = [0.0, 0.0, 1.0]
    d[k % 19 == 5] = [-1.0, 0.0, 0.0]
    return np.concatenate([o, d], axis=1)


@pytest.mark.parametrize("name", ["nee15", "main32"])
def test_slab_scan_edge_cases(port, name):
    rng = np.random.default_rng(77)
    rays = edge_rays(rng, 60_000)
    for desc in (pt.preset(name, 16, 16), scene_with_extra_planes(name, 16, 16, 1)):
        r = pt.Renderer(desc)
        s = oracle_scene(port, r, 16, 16, 1)
        nb = desc.body_count
        for skip in [-1] + list(range(6, nb)):  # every plane (and light) as the excluded body
            idx, t = r.intersect_batch(rays, skip=skip)
            _, idx0, t0 = port.intersect(s, rays, skip=skip)
            assert np.array_equal(idx, idx0), (name, nb, skip)
            assert same_bits(t, t0), (name, nb, skip)
        r.close()


@pytest.mark.parametrize("name,W,H,n_it", [("nee15", 70, 41, 5), ("main32", 37, 29, 4)])
def test_frames_with_extra_planes(port, name, W, H, n_it):
    # several emitters, a light inside a slab, coincident planes, image sizes that leave partial waves
    r = pt.Renderer(scene_with_extra_planes(name, W, H, 123))
    s = oracle_scene(port, r, W, H, 123)
    want, _ = port.render(s, MODE_PIXEL, 1, n_it)
    assert same_bits(r.render(1, n_it), want)
    r.set_partition(1, 3)
    part = r.render(1, n_it)
    r.set_partition(0, 1)
    assert np.count_nonzero(part) > 0 and same_bits(part[part != 0], want[part != 0])
    r.close()


def test_progressive_output_loop(port, tmp_path):
    """--progressive: the save-every-round loop the reference keeps commented out (main.cpp:614-640)."""
    import subprocess

    from conftest import ROOT

    W, H, iters, threads = 24, 16, 7, 2
    out = str(tmp_path / "prog.ppm")
    exe = os.path.join(ROOT, "path_tracer_amd", "bin", "main")
    res = subprocess.run([exe, str(W), str(H), str(iters), str(threads), "--seed", "5", "--out", out, "--quiet",
                          "--progressive"], check=True, timeout=120, capture_output=True, text=True)
    assert res.stdout.count("Iterations between output: 3") == 1 and res.stdout.count("Iterations between output: 1") == 1
    r = pt.Renderer(pt.preset("main32", W, H, seed=5))
    s = oracle_scene(port, r, W, H, 5)
    total = np.zeros((H, W, 3))
    first = 0
    for step, done in ((3, 6), (1, 8)):  # rounds of min(10, max(remaining / T, 1)) frames per worker
        for _ in range(threads):
            priv, _ = port.render(s, MODE_PIXEL, first, step)
            total = total + priv
            first += step
        with open(f"{out}.{done}", "rb") as f:
            assert f.read() == port.save_ppm(total, done, W, H)
    with open(out, "rb") as f:
        assert f.read() == port.save_ppm(total, iters, W, H)  # the final save divides by the requested total
    r.close()


def _custom_cases():
    with open(os.path.join(GOLDEN, "frames_custom.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _custom_cases(), ids=lambda c: c["key"])
def test_custom_scene_golden_frames(case):
    """Scenes the reference does not hard-code, rendered by the reference's own classes from the same
    constructor arguments (ten lights, coincident/extra planes, open walls, 1000 bodies -> wavefront)."""
    import ctypes as C

    import _oracle

    z = np.load(os.path.join(GOLDEN, "frames_custom.npz"))
    args, n = _oracle.bodies_from_bytes(z[case["key"] + "_bodies"], case.get("body_record", 136))
    assert n == case["bodies"]
    d = pt.preset("nee15" if case["prog"] == 1 else "main32", case["W"], case["H"], seed=case["seed"])
    bodies = (pt.Body * n).from_buffer_copy(bytes(args))  # same field layout as the oracle's record
    d.bodies = bodies
    d.body_count = n
    d._keep = bodies
    r = pt.Renderer(d)
    assert same_bits(r.render(case["first"], case["n_it"]), z[case["key"] + "_frame"])
    if n <= 127:
        r.set_pipeline(pt.PIPELINE_WAVEFRONT)
        assert same_bits(r.render(case["first"], case["n_it"]), z[case["key"] + "_frame"])
    r.close()


@pytest.mark.parametrize("keep", [(6, 7, 8, 9, 10, 11, 12, 13), (0, 1, 2, 3, 4, 5, 14), (6, 10, 14), (10, 14)],
                         ids=["planes_only", "spheres_only", "floor_top_light", "two_lights"])
def test_degenerate_body_lists_in_both_pipelines(port, keep):
    """Scenes without spheres, without planes, and with next to nothing: the list, slab and group
    builders must cope with empty lists, and both pipelines must still agree with the oracle."""
    W, H = 40, 24
    drop = tuple(i for i in range(15) if i not in keep)
    r = pt.Renderer(open_scene("nee15", W, H, 21, drop=drop))
    s = oracle_scene(port, r, W, H, 21)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    assert same_bits(r.render(0, 3), want)
    r.close()


def random_scene(seed, W, H, max_spheres=40):
    """A seeded random scene: axis-aligned and oblique planes, spheres of mixed sizes, all four
    combinations of material and emission, lights with and without colour (so that both the
    live and the dead direct term occur), either estimator."""
    rng = np.random.default_rng(1000 + seed)
    d = pt.preset("nee15", W, H, seed=seed)
    d.estimator = int(rng.integers(0, 2))
    n_planes = int(rng.integers(0, 9))
    n_spheres = int(rng.integers(0 if n_planes else 1, max_spheres))
    n = n_planes + n_spheres + 2
    bodies = (pt.Body * n)()

    def material(b, light):
        kind = int(rng.integers(0, 3))
        b.material = kind
        for c in range(3):
            b.color[c] = float(rng.uniform(0.1, 0.95))
            b.emission[c] = 0.0
        if kind == pt.GLASS:
            b.ior = float(rng.uniform(0.8, 1.8))
            b.reflection = float(rng.uniform(0.0, 0.5))
        b.brdf_id = -1
        if light:
            for c in range(3):
                b.emission[c] = float(rng.uniform(1.0, 2.0))
                b.color[c] = 0.0 if light == 1 else float(rng.uniform(0.0, 1.0))
        elif rng.uniform() < 0.15:  # glows, but below the light threshold
            for c in range(3):
                b.emission[c] = float(rng.uniform(0.0, 0.9))

    dark_lights = rng.uniform() < 0.6  # every light with colour 0 -> the direct term is dead scene-wide
    k = 0
    for i in range(n_planes):
        b = bodies[k]
        b.shape = pt.PLANE
        if rng.uniform() < 0.65:
            axis = int(rng.integers(0, 3))
            sign = 1.0 if rng.uniform() < 0.5 else -1.0
            c = [0.0, 0.0, 0.0]
            c[axis] = -sign * float(rng.uniform(0.6, 4.0)) + (2.0 if axis == 1 else 0.0)
            nrm = [0.0, 0.0, 0.0]
            nrm[axis] = sign
        else:
            nrm = list(rng.normal(size=3))
            c = list(rng.uniform(-3, 3, size=3) + np.array([0.0, 2.0, 0.0]))
        for q in range(3):
            b.center[q] = float(c[q])
            b.normal[q] = float(nrm[q])
        material(b, 0)
        k += 1
    for i in range(n_spheres + 2):
        b = bodies[k]
        b.shape = pt.SPHERE
        c = rng.uniform([-1.8, 0.6, -0.5], [1.8, 4.4, 2.4])
        for q in range(3):
            b.center[q] = float(c[q])
        b.radius = float(rng.choice([0.05, 0.1, 0.3, 0.6])) * float(rng.uniform(0.7, 1.3))
        light = 0
        if i >= n_spheres or rng.uniform() < 0.1:
            light = 1 if dark_lights else 2
        material(b, light)
        k += 1
    d.bodies = bodies
    d.body_count = n
    d._keep = bodies
    return d


@pytest.mark.parametrize("seed", range(16))
def test_random_scenes_match_oracle(port, seed):
    W, H = 32, 20
    r = pt.Renderer(random_scene(seed, W, H))
    s = oracle_scene(port, r, W, H, seed)
    want, _ = port.render(s, MODE_PIXEL, 1, 2)
    assert same_bits(r.render(1, 2), want)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    assert same_bits(r.render(1, 2), want)
    rng = np.random.default_rng(seed)
    rays = edge_rays(rng, 4000)
    bodies, nb = r.bodies()
    for skip in (-1, nb - 1, nb // 2):
        idx, t = r.intersect_batch(rays, skip=skip)
        _, idx0, t0 = port.intersect(s, rays, skip=skip)
        assert np.array_equal(idx, idx0) and same_bits(t, t0)
    r.close()


def test_raytrace3_golden(port):
    """raytrace3.cpp's path tracer (Rectangle shape, hit threshold 1e-7) against vectors produced by
    the reference itself: per-body intersect values, nearest hits, frames of its own scene and of
    body lists built from constructor arguments (tests/golden/make_golden_raytrace3.py)."""
    import _oracle

    with open(os.path.join(GOLDEN, "raytrace3.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, "raytrace3.npz"))
    r = pt.Renderer(pt.preset("raytrace3", 8, 8))
    assert same_bits(r.intersect_all(z["rays"]), z["t_all"])
    want = z["t_all"]
    idx, t = r.intersect_batch(z["rays"])
    for i in range(want.shape[0]):
        best, bi = np.inf, -1
        for b in range(want.shape[1]):
            if want[i, b] > 1e-7 and want[i, b] <= best:
                best, bi = want[i, b], b
        assert idx[i] == bi, i
        if bi >= 0:
            assert t[i] == best
    r.close()
    for c in meta["frames"]:
        r = pt.Renderer(pt.preset("raytrace3", c["W"], c["H"], seed=c["seed"]))
        assert same_bits(r.render(c["first"], c["n_it"]), z["frame_" + c["key"]]), c["key"]
        r.close()
    for c in meta["custom"]:
        d = rectangle_scene(c["seed"], c["W"], c["H"])
        args, nb = _oracle.bodies_from_bytes(z[c["key"] + "_bodies"], meta["body_record"])
        assert bytes(_oracle.bodies_to_bytes(args, nb)) == bytes(d._keep)  # the generator's scene, unchanged
        r = pt.Renderer(d)
        assert same_bits(r.render(c["first"], c["n_it"]), z[c["key"] + "_frame"]), c["key"]
        r.close()


@pytest.mark.parametrize("seed", range(10, 34))
def test_random_rectangle_scenes_match_oracle(port, seed):
    W, H = 36, 24
    d = rectangle_scene(seed, W, H, n_rect=3 + seed % 9, n_sphere=seed % 7, n_plane=seed % 4)
    r = pt.Renderer(d)
    s = oracle_scene(port, r, W, H, seed, flags=pt.F_RAYTRACE3)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    rng = np.random.default_rng(seed)
    rays = np.concatenate([rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5], (4000, 3)),
                           rng.normal(size=(4000, 3)) * rng.uniform(0.1, 3.0, (4000, 1))], axis=1)
    t_all, idx, t = port.intersect(s, rays)
    assert same_bits(r.intersect_all(rays), t_all)
    gi, gt = r.intersect_batch(rays)
    assert np.array_equal(gi, idx) and same_bits(gt[idx >= 0], t[idx >= 0])
    r.close()


def test_rectangles_need_the_path_tracing_estimator():
    d = rectangle_scene(1, 16, 16)
    d.estimator = pt.EST_NEE
    with pytest.raises(pt.PtError):
        pt.Renderer(d)


@pytest.mark.parametrize("seed", [3, 14, 25])
def test_rectangles_in_the_wavefront_pipeline(port, seed):
    """Rectangles and the 1e-7 hit threshold in the large-scene pipeline: forced on a small scene it
    equals the fused kernel and the oracle; with 150 extra spheres it is selected by itself."""
    W, H = 36, 24
    d = rectangle_scene(seed, W, H, n_rect=6, n_sphere=5, n_plane=2)
    r = pt.Renderer(d)
    s = oracle_scene(port, r, W, H, seed, flags=pt.F_RAYTRACE3)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    assert same_bits(r.render(0, 3), want)
    r.close()
    d = rectangle_scene(seed, W, H, n_rect=6, n_sphere=150, n_plane=2)
    r = pt.Renderer(d)
    s = oracle_scene(port, r, W, H, seed, flags=pt.F_RAYTRACE3)
    want, _ = port.render(s, MODE_PIXEL, 0, 2)
    assert same_bits(r.render(0, 2), want)
    r.close()


def free_port():
    """A TCP port nobody listens on right now (two sessions sharing a box must not pick the same one)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def test_two_ranks_share_one_gpu():
    """Two ranks (gloo, both on cuda:0) render their tile partitions concurrently and all-reduce; the
    reduced frame equals the single-partition frame bit for bit.  Concurrent processes stretch the gaps
    between the layout, estimator and accumulation streams, which is what exposes table-reuse races."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", free_port(),
           os.path.join(root, "scripts", "rehearse_ranks.py"), "60"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatching entries 0 " in out.stdout


# Converged anchors of the reference itself: its unmodified iterate() with glibc rand(), srand(1),
# 256x192 at 64 samples per pixel (mean radiance per channel, then the four image quadrants).
ANCHORS = {
    "nee15": dict(mean=(0.6039, 0.5014, 0.4825), TL=(0.6011, 0.4769, 0.3590), TR=(0.6279, 0.5570, 0.6307),
                  BL=(0.5993, 0.4782, 0.4504), BR=(0.5873, 0.4933, 0.4898)),
    "main32": dict(mean=(0.6519, 0.5092, 0.5251), TL=(0.8110, 0.5490, 0.5053), TR=(0.6191, 0.5376, 0.6451),
                   BL=(0.6076, 0.4777, 0.4576), BR=(0.5698, 0.4724, 0.4923)),
}


@pytest.mark.parametrize("name", ["nee15", "main32"])
def test_converged_image_statistics_match_the_reference_run(name):
    """Independent random streams cannot agree per pixel, but region means must: the GPU render at the
    anchors' size and sample count, against the reference's own run with its own rand().  The standard
    error comes from the spread of eight 8-iteration sub-renders; both runs carry it, hence sqrt(2)."""
    W, H, groups, per = 256, 192, 8, 8
    r = pt.Renderer(pt.preset(name, W, H, seed=1))
    subs = np.stack([r.render(g * per, per) / per for g in range(groups)])  # (8, H, W, 3)
    r.close()
    regions = dict(mean=(slice(0, H), slice(0, W)), TL=(slice(0, H // 2), slice(0, W // 2)),
                   TR=(slice(0, H // 2), slice(W // 2, W)), BL=(slice(H // 2, H), slice(0, W // 2)),
                   BR=(slice(H // 2, H), slice(W // 2, W)))
    for key, (ys, xs) in regions.items():
        m = subs[:, ys, xs, :].mean(axis=(1, 2))  # (8, 3) region means of the sub-renders
        mean = m.mean(axis=0)
        se = m.std(axis=0, ddof=1) / np.sqrt(groups)
        tol = 4.0 * np.sqrt(2.0) * se + 5e-5  # + the rounding of the four-digit anchors
        want = np.array(ANCHORS[name][key])
        assert (np.abs(mean - want) < tol).all(), (name, key, mean, want, tol)
        assert (tol < 0.02).all()  # the test has teeth: a percent-level bias would fail


def test_rccl_exchange_with_one_rank():
    """The multi-GPU exchange through RCCL itself: backend nccl, one rank on cuda:0; the all-gather of
    the owned tiles and the all-reduce of the FP64 accumulator are really issued (no single-rank
    shortcut) and the frame equals the plain render bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTGPU_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "rehearse_ranks.py"), "40"], env=env,
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "backend nccl" in out.stdout and "mismatching entries 0 " in out.stdout and "gather equals reduce True" in out.stdout


def test_tile_exchange_layout_matches_host_mirror():
    """pt_tiles_pack / pt_tiles_unpack against the numpy mirror in path_tracer_amd.distributed, on a frame
    with partial edge tiles: pack per rank, concatenate as an all-gather would, unpack."""
    import torch

    from path_tracer_amd import distributed as ptdist

    W, H = 100, 37
    rng = np.random.default_rng(4)
    frame = rng.normal(size=(H, W, 3))
    dev = torch.from_numpy(frame.reshape(-1)).cuda()
    for world in (1, 2, 3, 8):
        r = pt.Renderer(pt.preset("nee15", W, H, seed=1))
        T = r.tiles_per_rank(world)
        assert T == ptdist.tiles_per_rank(W, H, world)
        gathered = torch.full((world, T, 256, 3), -7.0, dtype=torch.float64, device="cuda")
        for rank in range(world):
            r.set_partition(rank, world)
            r.tiles_pack(dev.data_ptr(), gathered[rank].data_ptr(), T)
            torch.cuda.synchronize()
            assert same_bits(gathered[rank].cpu().numpy(), ptdist.pack_tiles_host(frame, rank, world))
        out = torch.zeros_like(dev)
        r.tiles_unpack(world, T, gathered.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        assert same_bits(out.cpu().numpy().reshape(H, W, 3), frame)
        assert same_bits(ptdist.unpack_tiles_host(gathered.cpu().numpy(), W, H, world), frame)
        with pytest.raises(pt.PtError):
            r.tiles_pack(dev.data_ptr(), gathered.data_ptr(), 0 if world > 1 else -1)
        r.close()


def test_overlapped_batches_equal_one_call_per_iteration():
    """One call over many batches (layout, estimator and accumulation overlap on three streams and
    reuse two table sets) gives the same bits as one call per iteration, where nothing is reused."""
    import torch
    W, H, IT = 1024, 768, 1200
    r = pt.Renderer(pt.preset("nee15", W, H, seed=1))
    stream = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    r.render_device(0, IT, whole.data_ptr(), stream)
    single = torch.zeros_like(whole)
    for k in range(IT):
        r.render_device(k, 1, single.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(whole.view(torch.int64), single.view(torch.int64))


def test_4k_frame_partitions_and_full_width_strip(port):
    """3840x2160 (the largest configured frame): eight tile partitions sum to the single-partition
    frame bit for bit; a full-width strip of it equals the oracle."""
    W, H = 3840, 2160
    r = pt.Renderer(pt.preset("nee15", W, H, seed=3))
    whole = r.render(5, 2)
    acc = np.zeros_like(whole)
    for rank in range(8):
        r.set_partition(rank, 8)
        acc += r.render(5, 2)
    assert same_bits(acc, whole)
    r.close()
    W, H = 3840, 6
    r = pt.Renderer(pt.preset("nee15", W, H, seed=3))
    s = oracle_scene(port, r, W, H, 3)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    r.close()


def test_brdf_table_files(tmp_path):
    """read_brdf (MatusikBRDF.h:84-108): a table file in the reference's format gives the lookups of
    the uploaded table; wrong dimensions, short and missing files are reported, not fatal."""
    r = pt.Renderer(pt.preset("nee15_measured", 8, 8, seed=1))
    table = pt.synthetic_brdf_table(2)
    path = str(tmp_path / "synthetic.binary")
    pt.write_brdf_file(path, table)
    r.brdf_upload(0, table)
    r.brdf_load_file(1, path)
    z = np.load(os.path.join(GOLDEN, "brdf.npz"))
    a_rgb, a_ind = r.brdf_lookup(0, z["angles"])
    b_rgb, b_ind = r.brdf_lookup(1, z["angles"])
    assert np.array_equal(a_ind, b_ind) and same_bits(a_rgb, b_rgb)
    # the value of an index: red, green and blue planes of the file, scaled (MatusikBRDF.h:68-72)
    n = pt.BRDF_TABLE_DOUBLES // 3
    i = int(a_ind[0])
    want = np.array([table[i] * (1.0 / 1500.0), table[n + i] * (1.15 / 1500.0), table[2 * n + i] * (1.66 / 1500.0)])
    if (want < 0).any():
        want[:] = 0.0
    assert same_bits(a_rgb[0], want)
    bad = str(tmp_path / "bad_dims.binary")
    with open(bad, "wb") as f:
        np.array([90, 90, 181], dtype=np.int32).tofile(f)
        table.tofile(f)
    short = str(tmp_path / "short.binary")
    with open(short, "wb") as f:
        np.array([90, 90, 180], dtype=np.int32).tofile(f)
        table[:1000].tofile(f)
    for p in (bad, short, str(tmp_path / "missing.binary")):
        with pytest.raises(pt.PtError):
            r.brdf_load_file(2, p)
    with pytest.raises(pt.PtError):
        r.brdf_upload(9, table)
    r.close()


def test_renderer_mirror_iterate_and_save(port, tmp_path):
    """The host mirror used the way worker() and main() use the reference class (main.cpp:436-442,
    656): construct, iterate() n times, save() — buffer and PPM bytes equal the oracle's."""
    W, H, n = 48, 32, 5
    r = pt.Renderer(pt.preset("nee15", W, H, seed=21))
    for _ in range(n):
        r.iterate()
    s = oracle_scene(port, r, W, H, 21)
    want, _ = port.render(s, MODE_PIXEL, 0, n)
    assert r.iterations_done == n and same_bits(r.buffer, want)
    out = str(tmp_path / "mirror.ppm")
    r.save(out)
    with open(out, "rb") as f:
        assert f.read() == port.save_ppm(want, n, W, H)
    r.close()


def test_two_scenes_alive_at_once():
    """Handles own their buffers and side streams: renders of two scenes queued on two torch streams
    without synchronising in between equal their solo renders."""
    import torch

    W, H, IT = 256, 192, 40
    ra = pt.Renderer(pt.preset("nee15", W, H, seed=5))
    rb = pt.Renderer(pt.preset("main32", W, H, seed=6))
    solo_a, solo_b = ra.render(0, IT), rb.render(0, IT)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ta = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    tb = torch.zeros_like(ta)
    torch.cuda.synchronize()
    for k in range(4):
        ra.render_device(k * 10, 10, ta.data_ptr(), sa.cuda_stream)
        rb.render_device(k * 10, 10, tb.data_ptr(), sb.cuda_stream)
    torch.cuda.synchronize()
    assert same_bits(ta.cpu().numpy().reshape(H, W, 3), solo_a)
    assert same_bits(tb.cpu().numpy().reshape(H, W, 3), solo_b)
    ra.close()
    rb.close()


def test_degenerate_rectangles(port):
    """Rectangles with coincident or collinear corners (the constructor normalises zero vectors:
    NaN normal or NaN edge directions), a huge and a tiny one: every intersect value, the nearest
    hits and the frames equal the oracle's (which never hits the NaN ones, like the reference)."""
    W, H = 30, 20
    d = rectangle_scene(5, W, H, n_rect=6, n_sphere=3, n_plane=1)
    cases = [
        [0.0, 2.0, 0.0] * 4,                                                    # a point
        [-1.0, 2.0, 0.0, 1.0, 2.0, 0.0, 2.0, 2.0, 0.0, 3.0, 2.0, 0.0],          # collinear corners
        [-1.0, 3.0, 0.0, -1.0, 3.0, 0.0, 1.0, 3.0, 1.0, -1.0, 3.0, 1.0],        # p1 == p2
        [-1e9, 4.0, -1e9, 1e9, 4.0, -1e9, 1e9, 4.0, 1e9, -1e9, 4.0, 1e9],       # huge
        [0.0, 1.5, 0.0, 1e-9, 1.5, 0.0, 1e-9, 1.5, 1e-9, 0.0, 1.5, 1e-9],       # tiny
    ]
    for k, c in enumerate(cases):
        for q in range(12):
            d.bodies[k].corners[q] = c[q]
    r = pt.Renderer(d)
    s = oracle_scene(port, r, W, H, 5, flags=pt.F_RAYTRACE3)
    want, _ = port.render(s, MODE_PIXEL, 0, 3)
    assert same_bits(r.render(0, 3), want)
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    assert same_bits(r.render(0, 3), want)
    rng = np.random.default_rng(8)
    rays = np.concatenate([rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5], (3000, 3)),
                           rng.normal(size=(3000, 3)) * rng.uniform(0.1, 3.0, (3000, 1))], axis=1)
    t_all, idx, t = port.intersect(s, rays)
    assert same_bits(r.intersect_all(rays), t_all)
    gi, gt = r.intersect_batch(rays)
    assert np.array_equal(gi, idx) and same_bits(gt[idx >= 0], t[idx >= 0])
    r.close()
