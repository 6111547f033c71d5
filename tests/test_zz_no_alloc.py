"""pt_render_device_noalloc (work buffers in the library's fixed device scratch, no allocation in the
call) against pt_render_device, through the C ABI, bit for bit: two calls in a row, the second on a
smaller frame, so the second call runs over what the first one left in the scratch."""
import numpy as np
import pytest

import path_tracer_amd as pt

pytestmark = pytest.mark.gpu


def _both(name, W, H, first, n_it, seed, partition=None):
    import torch

    r = pt.Renderer(pt.preset(name, W, H, seed=seed))
    if partition:
        r.set_partition(*partition)
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    got = torch.zeros_like(want)
    r.render_device(first, n_it, want.data_ptr(), stream)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    r.render_device_noalloc(first, n_it, got.data_ptr(), stream)
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    r.close()
    assert torch.equal(got.view(torch.int64), want.view(torch.int64)), (name, W, H)
    assert float(got.abs().sum()) > 0
    assert free_after >= free_before  # the call took no device memory
    return got


def test_noalloc_render_equals_render_device_twice_second_smaller():
    # 1024x768: 20.5 MB of tables and samples per iteration, three iterations per batch in the 64 MiB scratch
    _both("nee15", 1024, 768, 5, 7, seed=3)
    _both("nee15", 96, 40, 0, 300, seed=4)  # smaller frame, several batches of many iterations
    _both("main32", 50, 30, 2, 9, seed=5)
    _both("nee15_lit", 70, 33, 1, 5, seed=6, partition=(1, 3, 32, 8))


def test_noalloc_render_rejects_what_does_not_fit():
    import torch

    r = pt.Renderer(pt.preset("nee15", 2600, 1100, seed=1))  # 2.86 M pixels: one iteration exceeds the scratch
    a = torch.zeros(2600 * 1100 * 3, dtype=torch.float64, device="cuda")
    with pytest.raises(pt.PtError, match="scratch"):
        r.render_device_noalloc(0, 1, a.data_ptr(), 0)
    r.close()
    r = pt.Renderer(pt.preset("synthetic1k", 16, 8, seed=1))  # 1000 bodies: wavefront pipeline only
    b = torch.zeros(16 * 8 * 3, dtype=torch.float64, device="cuda")
    with pytest.raises(pt.PtError, match="fused"):
        r.render_device_noalloc(0, 1, b.data_ptr(), 0)
    r.close()


def test_noalloc_render_under_graph_capture():
    """The callers the entry point is for: the whole call is recorded into a HIP graph (capture mode
    "global", under which any hipMalloc, hipFree or synchronising call fails and invalidates the
    capture) and the replayed graph gives the bits of pt_render_device."""
    import torch

    W, H, first, n_it = 160, 90, 3, 11
    r = pt.Renderer(pt.preset("nee15", W, H, seed=8))
    want = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    r.render_device(first, n_it, want.data_ptr(), torch.cuda.current_stream().cuda_stream)
    warm = torch.zeros_like(want)
    r.render_device_noalloc(first, 1, warm.data_ptr(), torch.cuda.current_stream().cuda_stream)  # resolves the scratch address
    torch.cuda.synchronize()
    got = torch.zeros_like(want)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
        r.render_device_noalloc(first, n_it, got.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float(got.abs().sum()) == 0.0  # nothing ran during capture
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    got.zero_()
    graph.replay()  # the scratch keeps the first replay's contents; the second must not care
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    r.close()
