"""Per-section wave-cycle shares of the fused estimator kernel (development build).

  make -C path_tracer_amd/csrc prof
  PTGPU_LIB=path_tracer_amd/libptgpu_prof.so python scripts/section_profile.py [scene] [iterations]

Sections are timed with the wave cycle counter; nested sections are counted in their parents too
(scan_exact inside scan_screen; glass/diffuse/chrome/normal_sphere inside shade).  Beside the cycles
of a section the build sums the cycles times the lanes that took part in it, which gives the lanes
active per section and the share of all lane-slots each section leaves idle; the hemisphere
rejection loop, which lanes leave one by one, is counted trip by trip.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pt

NAMES = ["total", "setup", "scan_planes", "scan_screen", "scan_exact", "shade", "glass", "diffuse", "chrome",
         "normal_sphere", "nee_prep", "nee_vertex", "records", "reject_trips"]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "nee15"
    its = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    r = pt.Renderer(pt.preset(scene, 1024, 768, seed=1))
    L = pt.lib()
    N = len(NAMES)
    out = (C.c_ulonglong * (2 * N))()
    acc = r.render(0, its)
    L.pt_prof_read(out, 2 * N)  # reset after warm-up
    r.render(its, its, acc)
    n = L.pt_prof_read(out, 2 * N)
    assert n == N, n
    cyc, lanes = list(out[:N]), list(out[N:])
    tot = cyc[0]
    print(f"scene {scene}, {its} iterations, total wave-cycles {tot:.4g}; lanes that take part in the kernel body: "
          f"{lanes[0] / tot:.1f} of 64")
    print(f"  {'section':14s} {'wave-cycles':>14s}  share   lanes   idle lane-slots (of all 64 x total)")
    for k, name in enumerate(NAMES[:-1]):
        a = lanes[k] / cyc[k] if cyc[k] else 0.0
        print(f"  {name:14s} {cyc[k]:14d}  {100.0 * cyc[k] / tot:6.2f} %  {a:5.1f}  {100.0 * cyc[k] * (64.0 - a) / (64.0 * tot):6.2f} %")
    other = tot - cyc[1] - cyc[2] - cyc[3] - cyc[5] - cyc[10] - cyc[11] - cyc[12]
    print(f"  {'unattributed':14s} {other:14d}  {100.0 * other / tot:6.2f} %")
    k = N - 1
    if cyc[k]:
        per_wave = cyc[k]
        a = lanes[k] / cyc[k]
        print(f"  hemisphere rejection loop: {per_wave} trips by waves, {lanes[k]} by lanes: {a:.1f} of 64 lanes active per trip; "
              f"with the loop taking the share of `diffuse` above, it leaves {100.0 * cyc[7] * (64.0 - a) / (64.0 * tot):.2f} % of all lane-slots idle")


if __name__ == "__main__":
    main()
