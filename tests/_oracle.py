"""ctypes access to the two parity oracles (test infrastructure only).

* ``port``  — oracle/_build/libptoracle.so, the plain-C restatement (oracle/cpu_ref.c).
* ``ref``   — oracle/_ref/libptref.so / libptref_hooked.so, the reference's own sources
  compiled by oracle/Makefile.  Only present where that build has been run.
"""
import ctypes as C
import os
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT_SO = os.path.join(ROOT, "oracle", "_build", "libptoracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libptref.so")
REF_HOOKED_SO = os.path.join(ROOT, "oracle", "_ref", "libptref_hooked.so")

MODE_SEQ, MODE_SCRIPT, MODE_PIXEL = 0, 1, 2
EST_PT, EST_NEE, EST_BIDIR = 0, 1, 2  # EST_BIDIR: bidirectional.cpp:374-497
SPHERE, PLANE, RECTANGLE, RECTANGLE_MT = 0, 1, 2, 3
DIFFUSE, CHROME, GLASS, MEASURED = 0, 1, 2, 3
RAND_MAX = 2147483647
BRDF_N = 90 * 90 * 180


_BODY_FIELDS_136 = [
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
    ("pad", C.c_int32),
]


class Body(C.Structure):
    _fields_ = _BODY_FIELDS_136 + [("corners", C.c_double * 12)]  # rectangles: p1..p4 (raytrace3.cpp:254)


class Body136(C.Structure):
    """The record without corner points, as stored in the golden files made before rectangles."""

    _fields_ = _BODY_FIELDS_136


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("topleft", C.c_double * 3),
        ("topright", C.c_double * 3),
        ("bottomleft", C.c_double * 3),
    ]


class PorScene(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("camera", Camera),
        ("bodies", C.POINTER(Body)),
        ("body_count", C.c_int32),
        ("estimator", C.c_int32),
        ("flags", C.c_uint32),
        ("pad", C.c_uint32),
        ("seed", C.c_uint64),
        ("brdf_tables", C.c_void_p * 8),
    ]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bodies_to_list(bodies, n):
    out = []
    for i in range(n):
        b = bodies[i]
        out.append(
            dict(
                shape=int(b.shape),
                material=int(b.material),
                center=[float(v).hex() for v in b.center],
                normal=[float(v).hex() for v in b.normal],
                radius=float(b.radius).hex(),
                color=[float(v).hex() for v in b.color],
                emission=[float(v).hex() for v in b.emission],
                ior=float(b.ior).hex(),
                reflection=float(b.reflection).hex(),
                brdf_id=int(b.brdf_id),
            )
        )
    return out


def bodies_from_list(lst):
    arr = (Body * len(lst))()
    for i, d in enumerate(lst):
        b = arr[i]
        b.shape = d["shape"]
        b.material = d["material"]
        for k in ("center", "normal", "color", "emission"):
            for j in range(3):
                getattr(b, k)[j] = float.fromhex(d[k][j])
        b.radius = float.fromhex(d["radius"])
        b.ior = float.fromhex(d["ior"])
        b.reflection = float.fromhex(d["reflection"])
        b.brdf_id = d.get("brdf_id", -1)
    return arr


def camera_from_list(lst):
    cam = Camera()
    for k, name in enumerate(("origin", "topleft", "topright", "bottomleft")):
        for j in range(3):
            getattr(cam, name)[j] = float.fromhex(lst[k][j])
    return cam


def camera_to_list(cam):
    return [[float(v).hex() for v in getattr(cam, n)] for n in ("origin", "topleft", "topright", "bottomleft")]


class Port:
    """The C restatement."""

    def __init__(self, path=PORT_SO):
        self.lib = C.CDLL(path)
        L = self.lib
        L.por_render.restype = C.c_longlong
        L.por_save_ppm.restype = C.c_long
        L.por_time_render.restype = C.c_double

    def scene(self, bodies, n, cam, width, height, estimator, seed=1, flags=0, tables=None):
        s = PorScene()
        s.width, s.height = width, height
        s.camera = cam
        s.bodies = C.cast(bodies, C.POINTER(Body))
        s.body_count = n
        s.estimator = estimator
        s.flags = flags
        s.seed = seed
        s._keep = [bodies]
        if tables:
            for i, t in enumerate(tables):
                s.brdf_tables[i] = t.ctypes.data
                s._keep.append(t)
        return s

    def render(self, s, mode, first_it, n_it, script=None, accum=None):
        if accum is None:
            accum = np.zeros((s.height, s.width, 3), dtype=np.float64)
        sc = np.ascontiguousarray(script, dtype=np.int32) if script is not None else None
        draws = self.lib.por_render(
            C.byref(s), mode, first_it, n_it, _p(sc) if sc is not None else None, len(sc) if sc is not None else 0, _p(accum)
        )
        return accum, draws

    def render_rows(self, s, start_row, row_count, first_it, n_it, accum=None):
        """ray_iterate(start_row, row_count) (light_tracing.cpp:774-790), n_it passes."""
        if accum is None:
            accum = np.zeros((s.height, s.width, 3), dtype=np.float64)
        self.lib.por_render_rows(C.byref(s), int(start_row), int(row_count), int(first_it), int(n_it), _p(accum))
        return accum

    def row_split(self, height, workers):
        out = []
        for i in range(workers):
            a, b = C.c_int(), C.c_int()
            self.lib.por_row_split(int(height), int(workers), i, C.byref(a), C.byref(b))
            out.append((a.value, b.value))
        return out

    def frame_layout(self, s, it):
        ys = np.full(s.height, np.nan)
        xs = np.full((s.height, s.width), np.nan)
        rl = np.zeros(s.height, dtype=np.int32)
        rows = self.lib.por_frame_layout(C.byref(s), it, _p(ys), _p(xs), _p(rl))
        return rows, ys, xs, rl

    def intersect(self, s, rays6, skip=-1):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64)
        n = rays6.shape[0]
        t_all = np.zeros((n, s.body_count))
        idx = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        self.lib.por_intersect(C.byref(s), _p(rays6), n, skip, _p(t_all), _p(idx), _p(t))
        return t_all, idx, t

    def get_rays(self, s, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        rays = np.zeros((xy.shape[0], 6))
        self.lib.por_get_rays(C.byref(s), _p(xy), xy.shape[0], _p(rays))
        return rays

    def plane_normal(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros(3)
        self.lib.por_plane_normal(_p(v), _p(out))
        return out

    def rect_normal(self, corners12):
        c = np.ascontiguousarray(corners12, dtype=np.float64)
        out = np.zeros(3)
        self.lib.por_rect_normal(_p(c), _p(out))
        return out

    def bounce(self, s, body, ray6, normal3, script):
        ray6 = np.ascontiguousarray(ray6, dtype=np.float64)
        normal3 = np.ascontiguousarray(normal3, dtype=np.float64)
        sc = np.ascontiguousarray(script, dtype=np.int32)
        out = np.zeros(3)
        d = self.lib.por_bounce(C.byref(s), body, _p(ray6), _p(normal3), _p(sc), len(sc), _p(out))
        return out, d

    def trace_paths(self, s, it_idx, xy):
        it_idx = np.ascontiguousarray(it_idx, dtype=np.int32)
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        n = it_idx.shape[0]
        rgb = np.zeros((n, 3))
        nrec = np.zeros(n, dtype=np.int32)
        bodies = np.zeros((n, 5), dtype=np.int32)
        points = np.zeros((n, 5, 3))
        vis = np.zeros(n, dtype=np.int32)
        used = np.zeros(n, dtype=np.int32)
        self.lib.por_trace_paths(C.byref(s), _p(it_idx), _p(xy), n, _p(rgb), _p(nrec), _p(bodies), _p(points), _p(vis), _p(used))
        return dict(rgb=rgb, nrec=nrec, bodies=bodies, points=points, visible=vis, draws=used)

    def trace_brdf_indices(self, s, it_idx, xy):
        it_idx = np.ascontiguousarray(it_idx, dtype=np.int32)
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        n = it_idx.shape[0]
        ind = np.zeros((n, 5), dtype=np.int32)
        self.lib.por_trace_brdf_indices(C.byref(s), _p(it_idx), _p(xy), n, _p(ind))
        return ind

    def rng_fill(self, seed, it, dom, idx, n):
        k = np.zeros(n, dtype=np.int32)
        u = np.zeros(n)
        self.lib.por_rng_fill(C.c_uint64(seed), it, dom, idx, n, _p(k), _p(u))
        return k, u

    def brdf_lookup(self, table, angles4):
        angles4 = np.ascontiguousarray(angles4, dtype=np.float64)
        n = angles4.shape[0]
        rgb = np.zeros((n, 3))
        ind = np.zeros(n, dtype=np.int32)
        self.lib.por_brdf_lookup(_p(table), _p(angles4), n, _p(rgb), _p(ind))
        return rgb, ind

    def vector_to_spherical(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros((v.shape[0], 2))
        self.lib.por_vector_to_spherical(_p(v), v.shape[0], _p(out))
        return out

    def save_ppm(self, buffer, samples, width, height):
        buffer = np.ascontiguousarray(buffer, dtype=np.float64)
        cap = 32 + width * height * 16
        out = C.create_string_buffer(cap)
        n = self.lib.por_save_ppm(_p(buffer), samples, width, height, out, cap)
        return out.raw[:n]

    def time_render(self, s, iters_per_thread, threads):
        chk = C.c_double()
        secs = self.lib.por_time_render(C.byref(s), iters_per_thread, threads, C.byref(chk))
        return secs, chk.value


class Ref:
    """The reference's own code (oracle/_ref)."""

    def __init__(self, hooked=False):
        path = REF_HOOKED_SO if hooked else REF_SO
        self.lib = C.CDLL(path)
        self.scratch = tempfile.mkdtemp(prefix="ptor_")
        self.lib.ptor_set_scratch(self.scratch.encode())
        self.lib.ptor_render.restype = C.c_longlong
        self.lib.ptor_time_iterate.restype = C.c_double

    @staticmethod
    def available(hooked=False):
        return os.path.exists(REF_HOOKED_SO if hooked else REF_SO)

    def scene_bodies(self, prog):
        bodies = (Body * 64)()
        cam = Camera()
        n = self.lib.ptor_scene_bodies(prog, bodies, 64, C.byref(cam))
        return bodies, n, cam

    def intersect_all(self, prog, rays6, count):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64)
        t = np.zeros((rays6.shape[0], count))
        self.lib.ptor_intersect_all(prog, _p(rays6), rays6.shape[0], _p(t))
        return t

    def nearest_nee(self, rays6):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64)
        n = rays6.shape[0]
        idx = np.zeros(n, dtype=np.int32)
        pts = np.zeros((n, 3))
        self.lib.ptor_nearest_nee(_p(rays6), n, _p(idx), _p(pts))
        return idx, pts

    def get_rays(self, prog, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        rays = np.zeros((xy.shape[0], 6))
        self.lib.ptor_get_rays(prog, _p(xy), xy.shape[0], _p(rays))
        return rays

    def bounce(self, prog, body, ray6, normal3, script):
        ray6 = np.ascontiguousarray(ray6, dtype=np.float64)
        normal3 = np.ascontiguousarray(normal3, dtype=np.float64)
        sc = np.ascontiguousarray(script, dtype=np.int32)
        out = np.zeros(3)
        d = self.lib.ptor_bounce(prog, body, _p(ray6), _p(normal3), _p(sc), len(sc), _p(out))
        return out, d

    def render(self, prog, W, H, first_it, n_it, mode, seed=1, script=None):
        buf = np.zeros((H, W, 3))
        sc = np.ascontiguousarray(script, dtype=np.int32) if script is not None else None
        draws = self.lib.ptor_render(
            prog, W, H, first_it, n_it, mode, C.c_uint64(seed), _p(sc) if sc is not None else None,
            len(sc) if sc is not None else 0, _p(buf),
        )
        return buf, draws

    def render_custom(self, prog, bodies, n, W, H, first_it, n_it, mode, seed=1):
        """Frames of the reference's iterate() on a caller-supplied body list (constructor arguments)."""
        buf = np.zeros((H, W, 3))
        self.lib.ptor_render_custom.restype = C.c_longlong
        draws = self.lib.ptor_render_custom(prog, bodies, n, W, H, first_it, n_it, mode, C.c_uint64(seed), _p(buf))
        assert draws >= 0
        return buf, draws

    def trace_pt(self, rays6, seed, iteration):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64)
        rgb = np.zeros((rays6.shape[0], 3))
        self.lib.ptor_trace_pt(_p(rays6), rays6.shape[0], C.c_uint64(seed), iteration, _p(rgb))
        return rgb

    def save(self, prog, buffer, samples, W, H):
        buffer = np.ascontiguousarray(buffer, dtype=np.float64)
        name = C.create_string_buffer(64)
        self.lib.ptor_save(prog, _p(buffer), samples, W, H, name, 64)
        with open(os.path.join(self.scratch, name.value.decode()), "rb") as f:
            return f.read(), name.value.decode()

    def brdf_lookup(self, path, angles4):
        angles4 = np.ascontiguousarray(angles4, dtype=np.float64)
        rgb = np.zeros((angles4.shape[0], 3))
        rc = self.lib.ptor_brdf_lookup(path.encode(), _p(angles4), angles4.shape[0], _p(rgb))
        assert rc == 0
        return rgb

    def time_iterate(self, prog, W, H, iters_per_thread, threads, seed=1):
        chk = C.c_double()
        secs = self.lib.ptor_time_iterate(prog, W, H, iters_per_thread, threads, C.c_uint64(seed), C.byref(chk))
        return secs, chk.value


def bodies_to_bytes(bodies, n):
    """Body array -> uint8 numpy array (the 232-byte records, exact)."""
    return np.frombuffer(bytes(bodies)[: n * C.sizeof(Body)], dtype=np.uint8).copy()


def bodies_from_bytes(raw, record=136):
    """record: bytes per body in `raw` (136: golden files without corner points)."""
    if record == C.sizeof(Body):
        n = raw.size // record
        return (Body * n).from_buffer_copy(raw.tobytes()), n
    assert record == C.sizeof(Body136)
    n = raw.size // record
    old = (Body136 * n).from_buffer_copy(raw.tobytes())
    arr = (Body * n)()
    for i in range(n):
        C.memmove(C.byref(arr[i]), C.byref(old[i]), record)
    return arr, n


def constructed(port, bodies, n):
    """What the reference's constructors store for these constructor arguments: Plane keeps
    normalize(normal) (main.cpp:176-179), everything else is kept as given."""
    out = (Body * n)()
    for i in range(n):
        out[i] = Body.from_buffer_copy(bodies[i])
        if out[i].shape == PLANE:
            nn = port.plane_normal(np.array(list(bodies[i].normal)))
            for k in range(3):
                out[i].normal[k] = nn[k]
        if out[i].shape in (RECTANGLE, RECTANGLE_MT):  # normal = normalize((p3-p1) x (p2-p1)), raytrace3.cpp:260, light_tracing.cpp:494
            nn = port.rect_normal(np.array(list(bodies[i].corners)))
            for k in range(3):
                out[i].normal[k] = nn[k]
    return out
