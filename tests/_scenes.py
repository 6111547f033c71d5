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
