"""Rehearsal of the multi-rank render on one GPU: the exchanged frame must equal the
single-partition frame bit for bit.  With PTGPU_BENCH_BACKEND=gloo several ranks share cuda:0 (the
collectives are staged through host memory); with PTGPU_BENCH_BACKEND=nccl and one rank the same
calls — all-gather of the owned tiles, then all-reduce of the FP64 accumulator — run through RCCL.
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
backend = os.environ.get("PTGPU_BENCH_BACKEND", "gloo")
if backend == "nccl":
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group(backend=backend)
r = pt.Renderer(pt.preset("nee15", W, H, seed=1))
accum = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
exchange = ptdist.Exchange(r, world, accum.device, always=True)
for rep in range(2):
    accum.zero_()
    ptdist.render_distributed(r, 0, IT, accum, rank, world, stream, exchange=exchange)
    torch.cuda.synchronize()
gathered = accum.clone()
# the older exchange, a sum of zero-padded accumulators, must give the same frame
accum.zero_()
r.set_partition(rank, world)
r.render_device(0, IT, accum.data_ptr(), stream)
if backend == "gloo":
    ptdist.all_reduce_accumulator(accum)
else:
    dist.all_reduce(accum, op=dist.ReduceOp.SUM)  # also with a single rank: the RCCL call itself
torch.cuda.synchronize()
same = torch.equal(accum.view(torch.int64), gathered.view(torch.int64))
got = gathered.cpu().numpy().reshape(H, W, 3)
if rank == 0:
    r.set_partition(0, 1)
    want = r.render(0, IT)
    bad = np.argwhere(got.view(np.uint64) != want.view(np.uint64))
    print("backend", dist.get_backend(), "world", world, "iterations", IT, "mismatching entries", len(bad), "sum", got.sum(),
          want.sum(), "gather equals reduce", same, flush=True)
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "first", bad[:5].tolist())
        y, x, c = bad[0]
        print("  values", got[y, x], want[y, x], "ratio", got[y, x] / want[y, x])
        owner = ptdist.tile_owner(W, H, world)
        print("  owners of mismatching pixels", np.unique(owner[ys, xs], return_counts=True))
    status = torch.tensor([1 if len(bad) or not same else 0])
else:
    status = torch.tensor([0 if same else 1])
if backend == "nccl":
    status = status.cuda()
dist.all_reduce(status)
dist.destroy_process_group()
sys.exit(int(status.item()))
