"""path_tracer_amd — host-side mirror of the reference's Renderer seam over the ptgpu C ABI.

The reference (Melraidin/path_tracer) is C++ with no plugin interface; what calls its hot path is
``Renderer(Scene)`` / ``Renderer::iterate()`` / ``save()`` inside ``main.cpp`` and
``next_event.cpp``.  This module exposes the same three steps from Python on top of
``libptgpu.so`` (HIP, gfx950) for tests, benchmarks and multi-GPU orchestration.  It contains no
compute of its own and raises if the HIP library is missing: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTGPU_LIB selects another build of the same ABI (development builds with counters)
LIB_PATH = os.environ.get("PTGPU_LIB") or os.path.join(_HERE, "libptgpu.so")

SPHERE, PLANE, RECTANGLE, RECTANGLE_MT = 0, 1, 2, 3  # RECTANGLE_MT: light_tracing.cpp:477-575
DIFFUSE, CHROME, GLASS, MEASURED = 0, 1, 2, 3
EST_PT, EST_NEE, EST_BIDIR = 0, 1, 2  # EST_BIDIR: bidirectional.cpp:374-497
F_LIGHT_TRUE_CENTER = 1
F_RAYTRACE3 = 2  # raytrace3.cpp's path tracer: hit threshold 1e-7, see include/ptgpu.h
F_PIXEL_GRID = 4  # samples on the pixel grid as Renderer::ray_iterate takes them, light_tracing.cpp:785-790
BRDF_TABLE_DOUBLES = 3 * 90 * 90 * 180
DOM_PIXEL, DOM_ROW, DOM_FRAME, DOM_SEQ = 0, 1, 2, 3
PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_WAVEFRONT = 0, 1, 2


class PtError(RuntimeError):
    pass


class Body(C.Structure):
    """pt_body (include/ptgpu.h)."""

    _fields_ = [
        ("shape", C.c_int32),
        ("material", C.c_int32),
        ("center", C.c_double * 3),
        ("normal", C.c_double * 3),
        ("radius", C.c_double),
        ("color", C.c_double * 3),
        ("emission", C.c_double * 3),
        ("ior", C.c_double),
        ("reflection", C.c_double),
        ("brdf_id", C.c_int32),
        ("reserved", C.c_int32),
        ("corners", C.c_double * 12),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("topleft", C.c_double * 3),
        ("topright", C.c_double * 3),
        ("bottomleft", C.c_double * 3),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("camera", Camera),
        ("bodies", C.POINTER(Body)),
        ("body_count", C.c_int32),
        ("estimator", C.c_int32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
        ("seed", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_double), ("kernel_ms", C.c_double), ("launches", C.c_double)]


_lib = None


def lib():
    """Load libptgpu.so (built in-tree by path_tracer_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(
                f"{LIB_PATH} is missing: build it with `make -C path_tracer_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise PtError(f"ptgpu error {rc}: {lib().pt_last_error().decode()}")
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_device(device):
    _check(lib().pt_set_device(int(device)))


def preset(name, width, height, seed=1, cap=16384):
    """The reference's hard-coded scenes: 'nee15', 'main32', 'pt15', 'nee15_measured', 'synthetic10k'."""
    bodies = (Body * cap)()
    desc = SceneDesc()
    rc = lib().pt_preset_scene(name.encode(), width, height, C.byref(desc), bodies, cap)
    if rc != 0:
        raise PtError(f"unknown preset {name!r} (rc {rc})")
    desc.seed = seed
    desc._keep = bodies
    return desc


def synthetic_brdf_table(table_id, negative_fraction=0.03):
    """Synthetic MERL-format table: v = 1500*(0.05 + 0.9*h) with h a hash of (id, index); a fraction
    of entries is negative to exercise the below-horizon rule (MatusikBRDF.h:73-79)."""
    n = BRDF_TABLE_DOUBLES
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        salt = np.array([table_id + 1], dtype=np.uint64) * np.uint64(0xD1B54A32D192ED03)
        x = idx * np.uint64(0x9E3779B97F4A7C15) + salt
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    h = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    v = 1500.0 * (0.05 + 0.9 * h)
    neg = ((x & np.uint64(0xFFFF)).astype(np.float64) / 65536.0) < negative_fraction
    v[neg] = -v[neg]
    return v


def smooth_brdf_table(table_id):
    """Synthetic MERL-format table whose value is a smooth function of the three table angles at the
    bin centres, pi-periodic in phi_diff like a real isotropic BRDF: neighbouring bins (bins 0 and 179
    of phi_diff included) hold nearly equal values.  Positive everywhere; channels differ by a factor."""
    k = np.arange(90, dtype=np.float64) + 0.5
    theta_half = (k / 90.0) ** 2 * (np.pi / 2.0)
    theta_diff = k * (np.pi / 180.0)
    phi_diff = (np.arange(180, dtype=np.float64) + 0.5) * (np.pi / 180.0)
    lobe = np.cos(theta_half)[:, None, None] ** (4 + 2 * table_id)
    fres = (1.0 - np.cos(theta_diff)[None, :, None]) ** 3
    aniso = np.sin(theta_diff)[None, :, None] ** 2 * (1.0 + 0.8 * np.cos(2.0 * phi_diff)[None, None, :])
    base = 0.15 + 0.45 * lobe + 0.25 * fres + 0.15 * aniso  # (90, 90, 180), in [0.15, 1.3]
    plane = (1500.0 * base).reshape(-1)
    return np.concatenate([plane * (0.9 + 0.02 * table_id), plane * 0.8, plane * (0.6 + 0.05 * table_id)])


def write_brdf_file(path, table):
    """Write a table in the reference's .binary format (MatusikBRDF.h:90-103)."""
    with open(path, "wb") as f:
        np.array([90, 90, 180], dtype=np.int32).tofile(f)
        np.ascontiguousarray(table, dtype=np.float64).tofile(f)


class Renderer:
    """Mirror of the reference's ``Renderer`` (main.cpp:330-406, next_event.cpp:347-531).

    ``Renderer(desc)`` copies the scene to the GPU (the reference's constructor copies the Scene and
    builds the light list); ``iterate()`` adds one sample per pixel into ``buffer`` — an FP64 array
    of shape (H, W, 3), the reference's ``V3* buffer``; ``save()`` writes the reference's PPM.
    """

    def __init__(self, desc, device=None):
        L = lib()
        if device is not None:
            set_device(device)
        self.desc = desc
        self.width, self.height = desc.width, desc.height
        self._h = C.c_void_p()
        _check(L.pt_scene_create(C.byref(desc), C.byref(self._h)))
        self.buffer = np.zeros((self.height, self.width, 3), dtype=np.float64)
        self.iterations_done = 0

    def close(self):
        if self._h:
            lib().pt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the reference seam ---------------------------------------------------------------
    def iterate(self, n=1):
        """n calls of Renderer::iterate(): iterations [done, done+n) are added into self.buffer."""
        self.render(self.iterations_done, n, self.buffer)
        self.iterations_done += n

    def save(self, path, samples=None):
        samples = self.iterations_done if samples is None else samples
        save_ppm(self.buffer, samples, path)

    # -- explicit forms -------------------------------------------------------------------
    def render(self, first_iteration, n_iterations, accum=None):
        if accum is None:
            accum = np.zeros((self.height, self.width, 3), dtype=np.float64)
        assert accum.dtype == np.float64 and accum.flags.c_contiguous and accum.size == self.width * self.height * 3
        _check(lib().pt_render(self._h, int(first_iteration), int(n_iterations), _p(accum)))
        return accum

    def render_device(self, first_iteration, n_iterations, accum_ptr, stream_ptr=0):
        """accum_ptr: device address of W*H*3 doubles (e.g. torch_tensor.data_ptr())."""
        _check(
            lib().pt_render_device(
                self._h, int(first_iteration), int(n_iterations), C.c_void_p(accum_ptr), C.c_void_p(stream_ptr)
            )
        )

    def render_device_noalloc(self, first_iteration, n_iterations, accum_ptr, stream_ptr=0):
        """pt_render_device_noalloc: same result, work buffers in the library's fixed device scratch."""
        _check(
            lib().pt_render_device_noalloc(
                self._h, int(first_iteration), int(n_iterations), C.c_void_p(accum_ptr), C.c_void_p(stream_ptr)
            )
        )

    def set_partition(self, rank, world, tile_w=32, tile_h=8):
        _check(lib().pt_set_partition(self._h, rank, world, tile_w, tile_h))

    def set_rows(self, start_row, row_count):
        """Render only destination rows [start_row, start_row + row_count) (pt_set_rows)."""
        _check(lib().pt_set_rows(self._h, int(start_row), int(row_count)))

    def ray_iterate(self, start_row, row_count, n=1):
        """Renderer::ray_iterate(start_row, row_count) (light_tracing.cpp:774-790): the rows' pixels get
        n more samples in self.buffer; the other rows are left alone.  The scene decides how a pixel is
        sampled (F_PIXEL_GRID: on the grid, like the reference's loop)."""
        self.set_rows(start_row, row_count)
        try:
            self.render(self.iterations_done, n, self.buffer)
        finally:
            self.set_rows(0, self.height)

    def tiles_per_rank(self, world):
        return _check(lib().pt_tiles_per_rank(self._h, int(world)))

    def tiles_pack(self, accum_ptr, packed_ptr, tiles_capacity, stream_ptr=0):
        _check(lib().pt_tiles_pack(self._h, C.c_void_p(accum_ptr), C.c_void_p(packed_ptr), int(tiles_capacity),
                                   C.c_void_p(stream_ptr)))

    def tiles_unpack(self, world, tiles_per_rank, gathered_ptr, accum_ptr, stream_ptr=0):
        _check(lib().pt_tiles_unpack(self._h, int(world), int(tiles_per_rank), C.c_void_p(gathered_ptr),
                                     C.c_void_p(accum_ptr), C.c_void_p(stream_ptr)))

    def set_pipeline(self, pipeline):
        _check(lib().pt_set_pipeline(self._h, int(pipeline)))

    def stats(self):
        st = Stats()
        _check(lib().pt_get_stats(self._h, C.byref(st)))
        return dict(paths=st.paths, kernel_ms=st.kernel_ms, launches=st.launches)

    def bodies(self):
        cap = self.desc.body_count
        out = (Body * cap)()
        n = _check(lib().pt_scene_bodies(self._h, out, cap))
        return out, n

    def brdf_upload(self, brdf_id, table):
        table = np.ascontiguousarray(table, dtype=np.float64)
        _check(lib().pt_brdf_upload(self._h, brdf_id, _p(table), C.c_size_t(table.size)))

    def brdf_load_file(self, brdf_id, path):
        _check(lib().pt_brdf_load_file(self._h, brdf_id, path.encode()))

    def brdf_lookup(self, brdf_id, angles4):
        angles4 = np.ascontiguousarray(angles4, dtype=np.float64)
        n = angles4.shape[0]
        rgb = np.zeros((n, 3))
        ind = np.zeros(n, dtype=np.int32)
        _check(lib().pt_brdf_lookup(self._h, brdf_id, _p(angles4), n, _p(rgb), _p(ind)))
        return rgb, ind

    def brdf_lookup_vectors(self, brdf_id, vec6):
        vec6 = np.ascontiguousarray(vec6, dtype=np.float64)
        n = vec6.shape[0]
        rgb = np.zeros((n, 3))
        ind = np.zeros(n, dtype=np.int32)
        _check(lib().pt_brdf_lookup_vectors(self._h, brdf_id, _p(vec6), n, _p(rgb), _p(ind)))
        return rgb, ind

    def intersect_batch(self, rays6, skip=-1):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64)
        n = rays6.shape[0]
        idx = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        _check(lib().pt_intersect_batch(self._h, _p(rays6), n, skip, _p(idx), _p(t)))
        return idx, t

    def intersect_all(self, rays6):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64)
        n = rays6.shape[0]
        t = np.zeros((n, self.desc.body_count))
        _check(lib().pt_intersect_all(self._h, _p(rays6), n, _p(t)))
        return t

    def trace_paths(self, it_idx, xy):
        it_idx = np.ascontiguousarray(it_idx, dtype=np.int32)
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        n = it_idx.shape[0]
        out = dict(
            rgb=np.zeros((n, 3)),
            nrec=np.zeros(n, dtype=np.int32),
            bodies=np.zeros((n, 5), dtype=np.int32),
            points=np.zeros((n, 5, 3)),
            visible=np.zeros(n, dtype=np.int32),
            draws=np.zeros(n, dtype=np.int32),
        )
        _check(
            lib().pt_trace_paths(
                self._h, _p(it_idx), _p(xy), n, _p(out["rgb"]), _p(out["nrec"]), _p(out["bodies"]), _p(out["points"]),
                _p(out["visible"]), _p(out["draws"]),
            )
        )
        return out

    def trace_brdf_indices(self, it_idx, xy):
        it_idx = np.ascontiguousarray(it_idx, dtype=np.int32)
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        n = it_idx.shape[0]
        ind = np.zeros((n, 5), dtype=np.int32)
        _check(lib().pt_trace_brdf_indices(self._h, _p(it_idx), _p(xy), n, _p(ind)))
        return ind

    def frame_layout(self, iteration):
        ys = np.zeros(self.height)
        xs = np.zeros((self.height, self.width))
        rl = np.zeros(self.height, dtype=np.int32)
        nrows = C.c_int32()
        _check(lib().pt_frame_layout(self._h, int(iteration), _p(ys), _p(xs), _p(rl), C.byref(nrows)))
        return nrows.value, ys, xs, rl


def row_split(height, workers):
    """[(start_row, row_count)] per worker thread, as main() of the light-tracing programs splits the
    frame (light_tracing.cpp:1441-1451)."""
    out = []
    for i in range(workers):
        a, b = C.c_int(), C.c_int()
        rc = lib().pt_row_split(int(height), int(workers), i, C.byref(a), C.byref(b))
        if rc != 0:
            raise PtError(f"pt_row_split failed ({rc})")
        out.append((a.value, b.value))
    return out


def rng_fill(seed, iteration, domain, index, n):
    k = np.zeros(n, dtype=np.int32)
    u = np.zeros(n)
    _check(lib().pt_rng_fill(C.c_uint64(seed), iteration, domain, index, n, _p(k), _p(u)))
    return k, u


def save_ppm(buffer, samples, path):
    buffer = np.ascontiguousarray(buffer, dtype=np.float64)
    h, w = buffer.shape[0], buffer.shape[1]
    rc = lib().pt_save_ppm(_p(buffer), int(samples), w, h, path.encode())
    if rc != 0:
        raise PtError(f"pt_save_ppm failed ({rc})")
