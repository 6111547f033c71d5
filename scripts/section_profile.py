"""Per-section wave-cycle shares of the fused estimator kernel (development build).

  make -C path_tracer_amd/csrc prof
  PTGPU_LIB=path_tracer_amd/libptgpu_prof.so python scripts/section_profile.py [scene] [iterations]

Sections are timed with the wave cycle counter; nested sections are counted in their parents too
(scan_exact inside scan_screen; glass/diffuse/chrome/normal_sphere inside shade).
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pt

NAMES = ["total", "setup", "scan_planes", "scan_screen", "scan_exact", "shade", "glass", "diffuse", "chrome",
         "normal_sphere", "nee_prep", "nee_vertex", "records"]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "nee15"
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    r = pt.Renderer(pt.preset(scene, 1024, 768, seed=1))
    L = pt.lib()
    out = (C.c_ulonglong * len(NAMES))()
    acc = r.render(0, its)
    L.pt_prof_read(out, len(NAMES))  # reset after warm-up
    r.render(its, its, acc)
    n = L.pt_prof_read(out, len(NAMES))
    assert n == len(NAMES), n
    tot = out[0]
    print(f"scene {scene}, {its} iterations, total wave-cycles {tot:.4g}")
    for name, v in zip(NAMES, out):
        print(f"  {name:14s} {v:14d}  {100.0 * v / tot:6.2f} %")
    other = tot - out[1] - out[2] - out[3] - out[5] - out[10] - out[11] - out[12]
    print(f"  {'unattributed':14s} {other:14d}  {100.0 * other / tot:6.2f} %")


if __name__ == "__main__":
    main()
