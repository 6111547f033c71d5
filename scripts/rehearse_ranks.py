"""Rehearsal of the multi-rank render on one GPU (gloo, ranks share the device): the reduced frame
must equal the single-partition frame bit for bit.
    PTGPU_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P scripts/rehearse_ranks.py [iterations]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pt  # noqa: E402
from path_tracer_amd import distributed as ptdist  # noqa: E402

IT = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H = 1024, 768
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
pt.set_device(0)
dist.init_process_group(backend="gloo")
r = pt.Renderer(pt.preset("nee15", W, H, seed=1))
accum = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    accum.zero_()
    ptdist.render_distributed(r, 0, IT, accum, rank, world, stream)
    torch.cuda.synchronize()
got = accum.cpu().numpy().reshape(H, W, 3)
if rank == 0:
    r.set_partition(0, 1)
    want = r.render(0, IT)
    bad = np.argwhere(got.view(np.uint64) != want.view(np.uint64))
    print("world", world, "iterations", IT, "mismatching entries", len(bad), "sum", got.sum(), want.sum(), flush=True)
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "first", bad[:5].tolist())
        y, x, c = bad[0]
        print("  values", got[y, x], want[y, x], "ratio", got[y, x] / want[y, x])
        owner = ptdist.tile_owner(W, H, world)
        print("  owners of mismatching pixels", np.unique(owner[ys, xs], return_counts=True))
    status = torch.tensor([1 if len(bad) else 0])
else:
    status = torch.tensor([0])
dist.all_reduce(status)
dist.destroy_process_group()
sys.exit(int(status.item()))
