"""Scene variants used by both the GPU parity tests and the golden-vector generator."""
import path_tracer_amd as pt


def open_scene(name, W, H, seed, drop=(7, 8, 9, 10, 11, 12, 13)):
    """A reference scene with some walls removed, so that paths leave the scene and terminate."""
    d = pt.preset(name, W, H, seed=seed)
    keep = [i for i in range(d.body_count) if i not in drop]
    bodies = (pt.Body * len(keep))()
    for k, i in enumerate(keep):
        bodies[k] = d.bodies[i]
    d.bodies = bodies
    d.body_count = len(keep)
    d._keep = bodies
    return d


def scene_with_extra_planes(name, W, H, seed):
    """The reference scene plus axis-aligned planes that stress the slab form of the scan: a plane
    coincident with the left wall (a tie between bodies), a third x plane inside the box, a second
    emitting z plane above the top light and a duplicate of the back wall with the opposite normal."""
    d = pt.preset(name, W, H, seed=seed)
    n = d.body_count
    extra = [(8, None), (8, (0.5, 0.0, 0.0)), (10, (0.0, 0.0, 3.0)), (7, "flip")]  # 7 back, 8 left, 10 top light
    bodies = (pt.Body * (n + len(extra)))()
    for i in range(n):
        bodies[i] = d.bodies[i]
    for k, (src, change) in enumerate(extra):
        b = pt.Body.from_buffer_copy(d.bodies[src])
        if change == "flip":
            for c in range(3):
                b.normal[c] = -b.normal[c]
        elif change is not None:
            for c in range(3):
                b.center[c] = change[c]
        bodies[n + k] = b
    d.bodies = bodies
    d.body_count = n + len(extra)
    d._keep = bodies
    return d


def rectangle_scene(seed, W, H, n_rect=7, n_sphere=5, n_plane=2):
    """A seeded scene for raytrace3.cpp's path tracer: oblique and axis-aligned rectangles (some far,
    some nearer than t = 1, which the reference's Rectangle::intersect treats as a miss), spheres,
    planes, a few emitters; flag F_RAYTRACE3 (hit threshold 1e-7)."""
    import numpy as np

    rng = np.random.default_rng(4000 + seed)
    d = pt.preset("raytrace3", W, H, seed=seed)
    n = n_rect + n_sphere + n_plane
    bodies = (pt.Body * n)()

    def material(b):
        kind = int(rng.integers(0, 3))
        b.material = kind
        b.brdf_id = -1
        for c in range(3):
            b.color[c] = float(rng.uniform(0.1, 0.95))
            b.emission[c] = float(rng.uniform(0.5, 2.0)) if rng.uniform() < 0.3 else 0.0
        if kind == pt.GLASS:
            b.ior = float(rng.uniform(1.0, 1.8))
            b.reflection = float(rng.uniform(0.0, 0.5))

    k = 0
    for _ in range(n_rect):
        b = bodies[k]
        k += 1
        b.shape = pt.RECTANGLE
        centre = rng.uniform([-1.5, 1.0, -0.5], [1.5, 5.0, 2.5])
        if rng.uniform() < 0.5:  # axis-aligned
            axis = int(rng.integers(0, 3))
            u = np.zeros(3)
            v = np.zeros(3)
            u[(axis + 1) % 3] = rng.uniform(0.5, 3.0)
            v[(axis + 2) % 3] = rng.uniform(0.5, 3.0)
        else:
            u = rng.normal(size=3)
            u *= rng.uniform(0.5, 3.0) / np.linalg.norm(u)
            v = np.cross(u, rng.normal(size=3))
            v *= rng.uniform(0.5, 3.0) / np.linalg.norm(v)
        # p1 -> p2 along u, p2 -> p3 along v, p3 -> p4 back along u (the corner order of raytrace3.cpp:942)
        p1 = centre - 0.5 * u - 0.5 * v
        pts = [p1, p1 + u, p1 + u + v, p1 + v]
        for q in range(4):
            for c in range(3):
                b.corners[3 * q + c] = float(pts[q][c])
        material(b)
    for _ in range(n_sphere):
        b = bodies[k]
        k += 1
        b.shape = pt.SPHERE
        c = rng.uniform([-1.5, 1.0, -0.5], [1.5, 5.0, 2.5])
        for q in range(3):
            b.center[q] = float(c[q])
        b.radius = float(rng.uniform(0.1, 0.7))
        material(b)
    for _ in range(n_plane):
        b = bodies[k]
        k += 1
        b.shape = pt.PLANE
        axis = int(rng.integers(0, 3))
        sign = 1.0 if rng.uniform() < 0.5 else -1.0
        c = [0.0, 0.0, 0.0]
        c[axis] = -sign * float(rng.uniform(2.0, 5.0)) + (2.0 if axis == 1 else 0.0)
        nrm = [0.0, 0.0, 0.0]
        nrm[axis] = sign
        if rng.uniform() < 0.5:
            nrm = [float(x) for x in rng.normal(size=3)]
        for q in range(3):
            b.center[q] = c[q]
            b.normal[q] = nrm[q]
        material(b)
    d.bodies = bodies
    d.body_count = n
    d._keep = bodies
    return d


def mt_rectangle_scene(seed, W, H, n_rect=7, n_sphere=5, n_plane=2, grid=True, warp=0.0):
    """rectangle_scene with the rectangles as light_tracing.cpp's Moller-Trumbore kind, hit threshold
    t > 0, and (grid) samples on the pixel grid.  warp > 0 lifts every fourth corner off the plane of
    the other three, so that the two triangles of a rectangle are not coplanar."""
    import numpy as np

    d = rectangle_scene(seed, W, H, n_rect=n_rect, n_sphere=n_sphere, n_plane=n_plane)
    rng = np.random.default_rng(9000 + seed)
    for i in range(d.body_count):
        b = d.bodies[i]
        if b.shape == pt.RECTANGLE:
            b.shape = pt.RECTANGLE_MT
            if warp > 0.0:
                for c in range(3):
                    b.corners[9 + c] += float(rng.normal() * warp)
    d.flags = pt.F_PIXEL_GRID if grid else 0
    return d


def mt_edge_rays(d, rng, n_each=300):
    """Rays aimed at corners, edges and the shared diagonal of every Moller-Trumbore rectangle."""
    import numpy as np

    rays = []
    for i in range(d.body_count):
        b = d.bodies[i]
        if b.shape != pt.RECTANGLE_MT:
            continue
        p = np.array(list(b.corners)).reshape(4, 3)
        for _ in range(n_each):
            o = rng.uniform([-1.9, -2.5, -0.5], [1.9, 4.5, 2.5])
            kind = int(rng.integers(0, 4))
            if kind == 0:
                target = p[int(rng.integers(0, 4))]
            elif kind == 1:
                a = int(rng.integers(0, 4))
                target = p[a] + rng.uniform() * (p[(a + 1) % 4] - p[a])
            elif kind == 2:
                target = p[0] + rng.uniform() * (p[2] - p[0])  # the diagonal both triangles share
            else:
                target = p[0] + rng.uniform() * (p[1] - p[0]) + rng.uniform() * (p[3] - p[0])
            v = target - o
            if kind != 3 and rng.uniform() < 0.5:
                v /= np.linalg.norm(v)
            rays.append(np.concatenate([o, v]))
        # rays lying in the rectangle's plane (|det| tiny) and rays starting on it
        e1, e2 = p[1] - p[0], p[2] - p[0]
        rays.append(np.concatenate([p[0] - e1, e1 + 0.5 * e2]))
        rays.append(np.concatenate([p[0] + 0.25 * e1 + 0.25 * e2, np.cross(e1, e2)]))
    return np.array(rays)
