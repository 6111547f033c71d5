"""Development: per-wave counters of the grouped large-scene scan (a -DPT_SCAN_STATS build).
   PTGPU_LIB=scripts/ko/libptgpu_scanstats.so python scripts/scan_stats.py [iterations]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pt

its = int(sys.argv[1]) if len(sys.argv) > 1 else 4
r = pt.Renderer(pt.preset("synthetic10k", 1024, 768, seed=1))
L = pt.lib()
out = (C.c_ulonglong * 8)()
r.render(0, its)
assert L.pt_scan_stats(out) == 8
w = float(out[0])
print(f"wave-scans {out[0]}  per wave-scan: super pairs entered {out[1]/w:.1f} of 79, group pairs tested {out[2]/w:.1f}, "
      f"groups entered {out[3]/w:.1f}, member steps with a candidate {out[4]/w:.1f}, evaluation rounds {out[5]/w:.1f} "
      f"with {out[6]/max(1,out[5]):.1f} lanes each")
