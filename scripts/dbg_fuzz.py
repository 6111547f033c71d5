import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import path_tracer_amd as pt
import _oracle
from _oracle import MODE_PIXEL
import test_gpu_parity as T

seed = int(sys.argv[1])
W, H = 32, 20
port = _oracle.Port()
d = T.random_scene(seed, W, H)
print("estimator", d.estimator, "bodies", d.body_count)
for i in range(d.body_count):
    b = d.bodies[i]
    print(i, "shape", b.shape, "mat", b.material, "c", list(b.center), "n", list(b.normal), "r", b.radius, "col", list(b.color), "em", list(b.emission), "ior", b.ior)
r = pt.Renderer(d)
s = T.oracle_scene(port, r, W, H, seed)
for it in (1, 2):
    want, _ = port.render(s, MODE_PIXEL, it, 1)
    got = r.render(it, 1)
    bad = np.argwhere(got.view(np.uint64) != want.view(np.uint64))
    print("iteration", it, "mismatching entries", len(bad))
    for y, x, c in bad[:10]:
        print("  ", y, x, c, got[y, x, c], want[y, x, c])
    r.set_pipeline(pt.PIPELINE_WAVEFRONT)
    gw = r.render(it, 1)
    print("  wavefront mismatches", int((gw.view(np.uint64) != want.view(np.uint64)).sum()))
    r.set_pipeline(pt.PIPELINE_FUSED)
    if len(bad):
        y, x, _ = bad[0]
        idx = y * W + x
        it_idx = np.array([[it, idx]], dtype=np.int32)
        # x, y of that pixel from the frame layout
        nrows, ys, xs, rl = r.frame_layout(it)
        # find row/col of running index
        starts = np.concatenate([[0], np.cumsum(rl)])
        row = int(np.searchsorted(starts, idx, side="right") - 1)
        col = idx - starts[row]
        xy = np.array([[xs[row, col], ys[row]]])
        g = r.trace_paths(it_idx, xy)
        o = port.trace_paths(s, it_idx, xy)
        for k in ("nrec", "bodies", "draws", "visible", "rgb"):
            print("   ", k, g[k], o[k])
