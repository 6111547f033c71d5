"""Renders the headline workload several times in one process and compares the frames bit for bit
(the overlapped layout / estimator / accumulation streams must not change a single sample).
    python scripts/determinism_check.py [scene] [iterations] [repeats]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pt  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "nee15"
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 7200
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W, H = 1024, 768
torch.cuda.set_device(0)
pt.set_device(0)
r = pt.Renderer(pt.preset(scene, W, H, seed=1))
accum = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
frames = []
for rep in range(REP):
    accum.zero_()
    r.render_device(0, IT, accum.data_ptr(), stream)
    torch.cuda.synchronize()
    frames.append(accum.cpu().numpy().copy())
    print("repeat", rep, "mean", frames[-1].sum() / (W * H * 3 * IT),
          "entries differing from repeat 0", int((frames[-1].view(np.uint64) != frames[0].view(np.uint64)).sum()),
          flush=True)
# the same iterations rendered in 8 separate calls must add up to the same frame only up to FP64
# summation order, so that comparison is on the mean
accum.zero_()
for k in range(8):
    r.render_device(k * (IT // 8), IT // 8, accum.data_ptr(), stream)
torch.cuda.synchronize()
print("8 calls mean", accum.cpu().numpy().sum() / (W * H * 3 * (IT // 8 * 8)))
# one call per iteration: a call with a single batch never reuses a table set, so nothing overlaps
accum.zero_()
for k in range(IT):
    r.render_device(k, 1, accum.data_ptr(), stream)
torch.cuda.synchronize()
single = accum.cpu().numpy()
print("one call per iteration: mean", single.sum() / (W * H * 3 * IT),
      "largest relative difference from the one-call frame",
      float(np.max(np.abs(single - frames[0]) / np.maximum(np.abs(frames[0]), 1e-300))))
sys.exit(0 if all((f.view(np.uint64) == frames[0].view(np.uint64)).all() for f in frames) else 1)
