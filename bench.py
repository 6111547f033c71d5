"""Benchmark of the estimator on N GPUs of one node: paths/s, roofline of the dominant kernel, the
CPU partner timed beside it.

    python bench.py                                  # headline: --config C2, --gpus 1
    python bench.py --config C3|C4|C5                # the other configured workloads at their full sizes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Configured workloads (all seed 1, synthetic data):
    C2  next_event.cpp's 15-body scene, 1024x768, 7200 iterations (the path_tracing_1024x768x7200i run)
    C3  the same scene with bodies 0, 1, 6, 7 on measured-BRDF tables (four synthetic MERL-format
        tables: the .binary blobs are not available), 1024x768, 7200 iterations
    C4  the 15-body scene at 3840x2160, 4096 iterations
    C5  the 15-body room + 9985 spheres (10 000 bodies), 1024x768, 1024 iterations (wavefront pipeline)

A step is one complete render of the workload into a zeroed FP64 accumulator resident in HBM:
every rank renders its interleaved 32x8 destination tiles, then the owned tiles are exchanged with
one RCCL all-gather so that every rank holds the whole frame.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector FP64, FMA counted as 2 flop

CONFIGS = {
    "C2": dict(scene="nee15", width=1024, height=768, iterations=7200),
    "C3": dict(scene="nee15_measured", width=1024, height=768, iterations=7200),
    "C4": dict(scene="nee15", width=3840, height=2160, iterations=4096),
    "C5": dict(scene="synthetic10k", width=1024, height=768, iterations=1024),
}


def pmc_summary(scene):
    """Per-path figures of the estimator kernel from the committed counter passes
    (scripts/pmc.sh -> profiles/pmc_summary.json), or None when this scene has none."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(scene)


def host_cores():
    """Cores this process may use: the affinity mask, cut to the cgroup CPU quota and to
    OMP_NUM_THREADS when those are set (a GPU box hands each job a share of a larger machine)."""
    info = {"host_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    cores = info["affinity"]
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            info["cgroup_cpu_quota"] = float(quota) / float(period)
            cores = min(cores, max(1, int(info["cgroup_cpu_quota"])))
    except (OSError, ValueError):
        pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        info["omp_num_threads"] = int(os.environ["OMP_NUM_THREADS"])
        cores = min(cores, max(1, info["omp_num_threads"]))
    info["cores"] = max(1, cores)
    return info


def oracle_scene_of(port, desc, width, height, tables):
    """The CPU restatement's scene for a preset (bodies as the constructors store them)."""
    import ctypes

    import _oracle
    import path_tracer_amd as pt

    stored = (_oracle.Body * desc.body_count)()
    for i in range(desc.body_count):
        tmp = pt.Body.from_buffer_copy(desc.bodies[i])
        pt.lib().pt_body_construct(ctypes.byref(tmp))
        for f, _ in _oracle.Body._fields_:
            setattr(stored[i], f, getattr(tmp, "reserved" if f == "pad" else f))
    cam = _oracle.Camera()
    for f in ("origin", "topleft", "topright", "bottomleft"):
        for k in range(3):
            getattr(cam, f)[k] = getattr(desc.camera, f)[k]
    return port.scene(stored, desc.body_count, cam, width, height, desc.estimator, seed=1, flags=desc.flags,
                      tables=tables)


def cpu_baseline(desc, scene_name, width, height, tables, seconds_target):
    """The CPU partner on this host's cores, one independent single-thread worker per core (own
    random stream each, which is what the per-thread rand() of the reference's platform gives).
    The reference's own Renderer::iterate() (oracle/_ref) for its hard-coded 15-body scene when
    that build is present, else the C restatement (oracle/cpu_ref.c).  Bounded sample: the frame is
    shrunk (same scene, same camera) until one iteration per worker fits the time budget."""
    import _oracle

    info = host_cores()
    cores = info["cores"]
    use_ref = scene_name == "nee15" and _oracle.Ref.available()
    per_core = 0.65e6 if desc.body_count < 100 else 0.65e6 * 15.0 / desc.body_count  # paths/s, to size the sample
    w, h = width, height
    while w * h > 64 and w * h / per_core > seconds_target:
        w, h = max(8, w // 2), max(8, h // 2)
    iters = max(1, int(seconds_target * per_core / (w * h)))
    one_core = None
    if use_ref:
        ref = _oracle.Ref()
        secs, _ = ref.time_iterate(1, w, h, iters, cores, seed=1)
        kind = "reference"
        one_it = max(1, iters // 4)
        secs1, _ = ref.time_iterate(1, w, h, one_it, 1, seed=1)
        one_core = float(w * h) * one_it / secs1
    else:
        port = _oracle.Port()
        s = oracle_scene_of(port, desc, w, h, tables)
        secs, _ = port.time_render(s, iters, cores)
        kind = "port"
        one_it = max(1, iters // 4)
        secs1, _ = port.time_render(s, one_it, 1)
        one_core = float(w * h) * one_it / secs1
    paths = float(w * h) * iters * cores
    out = {
        "value": paths / secs,
        "unit": "paths/s",
        "cores": cores,
        "kind": kind,
        "one_core_value": one_core,
        "rng": "wrapped xoshiro128+ (the build's stream contract; cheaper than glibc rand(), which also takes a lock)",
        "sample": f"{scene_name} scene {w}x{h}, {iters} iteration(s) on each of {cores} independent "
                  f"single-thread workers ({paths:.3g} paths, {secs:.1f} s)",
    }
    literal = os.path.join(ROOT, "oracle", "_ref", "next_event_literal")
    if scene_name == "nee15" and os.path.exists(literal):
        # the literal run of the reference: `next_event <w> <h> <iterations> <thread count>` with thread count =
        # cores and the C library's rand(), whose one generator every worker thread locks (next_event.cpp:19-22,
        # 701-704); it squares the frame (height = argv[1], :596).  A small sample: it is several times slower
        # than one thread alone.
        import subprocess
        import tempfile

        lw, lit = 128, cores
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            try:
                subprocess.run([literal, str(lw), str(lw), str(lit), str(cores)], stdout=subprocess.DEVNULL, check=True,
                               cwd=tmp, timeout=120)
                lsecs = time.perf_counter() - t0
                out["literal_threads"] = {
                    "value": float(lw) * lw * lit / lsecs, "unit": "paths/s", "threads": cores, "rng": "glibc rand(), one locked generator shared by the threads",
                    "sample": f"next_event {lw} {lw} {lit} {cores}: {lw}x{lw} frame, {lit} iterations over {cores} threads ({lsecs:.1f} s)",
                }
            except (subprocess.SubprocessError, OSError) as e:
                out["literal_threads"] = {"error": str(e)}
    out.update({k: v for k, v in info.items() if k != "cores"})
    # what the whole machine would do if every core of it ran a worker at the one-core rate
    out["all_host_cores_extrapolated"] = one_core * info["host_cpu_count"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--iterations", type=int, default=0)
    ap.add_argument("--scene", default="")
    ap.add_argument("--pipeline", default="", choices=["", "fused", "wavefront"])
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--proxy-world", type=int, default=0,
                    help="single-GPU proxy of strong scaling: render only rank 0's tiles of a world of this many ranks "
                         "(full iteration count, no exchange) and report that rank's time")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.scene:
        cfg["scene"] = args.scene
    for k in ("width", "height", "iterations"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)
    as_configured = cfg == CONFIGS[args.config]

    import torch
    import torch.distributed as dist

    import path_tracer_amd as pt
    from path_tracer_amd import distributed as ptdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        # nccl is RCCL on ROCm; PTGPU_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse this path
        backend = os.environ.get("PTGPU_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(local_rank)
    pt.set_device(local_rank)

    W, H, IT, scene = cfg["width"], cfg["height"], cfg["iterations"], cfg["scene"]
    desc = pt.preset(scene, W, H, seed=1)
    r = pt.Renderer(desc)
    tables = []
    for k in sorted({desc.bodies[i].brdf_id for i in range(desc.body_count) if desc.bodies[i].material == pt.MEASURED}):
        while len(tables) <= k:
            tables.append(None)
        tables[k] = pt.synthetic_brdf_table(k)  # the MERL .binary blobs are not available: synthetic tables
        r.brdf_upload(k, tables[k])
    n_measured = sum(t is not None for t in tables)
    if args.pipeline:
        r.set_pipeline({"fused": pt.PIPELINE_FUSED, "wavefront": pt.PIPELINE_WAVEFRONT}[args.pipeline])
    # algorithmic work per path: 14 flop per plane test, 19 per sphere test, 56 per rectangle test
    # (raytrace3.cpp:283-308: the plane test, the hit point, two differences, two normalisations, two
    # dot products); 5 gather scans over all bodies (+ for NEE 5 shadow scans that leave out the
    # chosen light, the last emitter)
    planes = sum(1 for i in range(desc.body_count) if desc.bodies[i].shape == pt.PLANE)
    rectangles = sum(1 for i in range(desc.body_count) if desc.bodies[i].shape == pt.RECTANGLE)
    spheres = desc.body_count - planes - rectangles
    scan = 14.0 * planes + 19.0 * spheres + 56.0 * rectangles
    flop_per_path = 5.0 * scan
    rays_per_path = 5
    if desc.estimator == pt.EST_NEE:
        lights = [i for i in range(desc.body_count) if max(desc.bodies[i].emission) >= 1.0]
        flop_per_path += 5.0 * (scan - (19.0 if desc.bodies[lights[-1]].shape == pt.SPHERE else 14.0))
        rays_per_path = 10
    if desc.estimator == pt.EST_BIDIR:
        # bidirectional.cpp:394-437: five gather scans, five light-path scans, and one scan per (gather vertex,
        # light-path vertex) pair with the chosen light (a sphere in every draw of the shipped scene) left out
        flop_per_path = 10.0 * scan + 25.0 * (scan - 19.0)
        rays_per_path = 35
    wavefront = args.pipeline == "wavefront" or (args.pipeline != "fused" and desc.body_count > 127)
    kernel_name = "k_wf_* stages (wavefront pipeline)" if wavefront else (
        "k_trace<NEE>" if desc.estimator == pt.EST_NEE else "k_trace<BIDIR>" if desc.estimator == pt.EST_BIDIR else "k_trace<PT>")
    accum = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    exchange = ptdist.Exchange(r, world, accum.device) if args.exchange == "gather" else None

    proxy = args.proxy_world if world == 1 else 0

    proxy_rank = 0

    def step():
        accum.zero_()
        if proxy:
            r.set_partition(proxy_rank, proxy, ptdist.TILE_W, ptdist.TILE_H)
            r.render_device(0, IT, accum.data_ptr(), stream)
        else:
            ptdist.render_distributed(r, 0, IT, accum, rank, world, stream, exchange=exchange)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, launches, traced = 0.0, 0.0, 0.0
    for _ in range(args.steps):
        step()
        torch.cuda.synchronize()
        st = r.stats()  # HIP events recorded on the render stream around every estimator launch (this rank)
        kernel_ms += st["kernel_ms"]
        launches += st["launches"]
        traced += st["paths"]
    fence()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rank_seconds = [elapsed / args.steps]
    for proxy_rank in range(1, proxy):  # the other ranks of the proxy world, one after the other
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        rank_seconds.append((time.perf_counter() - t1) / args.steps)
    # N > 1 checks itself: a short render through the N-rank exchange against the same render by rank 0
    # alone (bit compare of the FP64 frames), the frame every rank ends up with, and every rank's times
    self_check = None
    if world > 1:
        check_it = min(8, IT)
        acc_d = torch.zeros_like(accum)
        ptdist.render_distributed(r, 0, check_it, acc_d, rank, world, stream, exchange=exchange)
        torch.cuda.synchronize()
        digest = int(acc_d.view(torch.int64).sum().item())  # wraps; equal frames give equal digests
        equal = None
        if rank == 0:
            acc_s = torch.zeros_like(accum)
            r.set_partition(0, 1, ptdist.TILE_W, ptdist.TILE_H)
            r.render_device(0, check_it, acc_s.data_ptr(), stream)
            torch.cuda.synchronize()
            equal = bool(torch.equal(acc_d.view(torch.int64), acc_s.view(torch.int64)))
            del acc_s
        mine = {"rank": rank, "device": torch.cuda.current_device(), "frame_digest": digest,
                "estimator_kernel_ms_per_step": kernel_ms / args.steps, "estimator_launches_per_step": launches / args.steps,
                "paths_per_step": traced / args.steps, "wall_ms_per_step": 1e3 * own_elapsed / args.steps}
        every = [None] * world
        dist.all_gather_object(every, mine)
        km = [e["estimator_kernel_ms_per_step"] for e in every]
        self_check = {
            "frame_equals_single_gpu": equal,
            "checked_iterations": check_it,
            "every_rank_holds_the_same_frame": len({e["frame_digest"] for e in every}) == 1,
            "rccl_ranks": dist.get_world_size(),
            "backend": dist.get_backend(),
            "devices": sorted({e["device"] for e in every}),
            "estimator_kernel_ms_per_step": {"max": max(km), "min": min(km)},
            "ranks": every,
        }
    paths = traced if proxy else float(W) * H * IT * args.steps
    checksum = float(accum.sum().item()) / (W * H * 3 * IT)
    if rank == 0:
        out = {
            "metric": "paths/sec (whole node), 1024x768 NEE, 1/2/4/8 GPUs; % of FP64 VALU peak",
            "value": paths / elapsed,
            "unit": "paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config if as_configured else 'custom'}: {scene} scene ({desc.body_count} bodies), "
                            f"{W}x{H}, {IT} iterations (spp), seed 1",
                "rays_per_path": rays_per_path,
                "rays_per_s": rays_per_path * paths / elapsed,
                "bodies": desc.body_count,
                "measured_brdf_tables": n_measured,
                "partition": f"interleaved 32x8 tiles over {world} GPU(s), RCCL "
                             + ("all-gather of the owned tiles" if exchange else "all-reduce of the FP64 accumulators"),
                "mean_radiance": checksum,
            },
        }
        if self_check:
            out.update({k: self_check[k] for k in ("frame_equals_single_gpu", "rccl_ranks")})
            out["self_check"] = self_check
        if proxy:
            # what one rank of `proxy` ranks does on its GPU, measured alone: value is that rank's path rate,
            # rank_seconds its time for the whole workload's share; T(1) / (N T(N)) is the efficiency
            out["proxy"] = {"world": proxy, "rank": 0, "rank_seconds_per_step": elapsed / args.steps,
                            "rank_paths_per_step": traced / args.steps, "exchange": "not run",
                            "every_rank_seconds_per_step": rank_seconds, "slowest_rank_seconds_per_step": max(rank_seconds)}
            out["config"]["partition"] = f"rank 0 of {proxy} (interleaved 32x8 tiles), alone on one GPU, no exchange"
        if launches > 0:
            paths_per_launch = traced / launches  # rank 0's own launches: a per-GPU figure when N > 1
            avg_ms = kernel_ms / launches
            achieved = paths_per_launch * flop_per_path / (avg_ms * 1e-3) / 1e12
            pmc = pmc_summary(scene)
            out["roofline"] = {
                "bound": "valu_fp64",
                "achieved": achieved,
                "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_VALU_PEAK_TFLOPS,
                "frac_of_no_fma_ceiling": achieved / (FP64_VALU_PEAK_TFLOPS / 2),
                "traffic": pmc["bytes_per_path"] * paths_per_launch if pmc and pmc.get("bytes_per_path") else None,
                "note": "algorithmic: every body test of every scan the reference makes, priced at its FP64 cost "
                        "(14 flop per plane, 19 per sphere); the kernels answer many of them with FP32 screens, "
                        "slabs, bounding groups or by proving a shadow scan dead, so this is a brute-force-equivalent "
                        "rate and may exceed what the FP64 units execute (see `executed`)",
                "hbm_view": {
                    "algorithmic_GBps": paths_per_launch * 24.0 / (avg_ms * 1e-3) / 1e9,
                    "peak_GBps": 8000.0,
                    "note": "24 B of FP64 sample per path; the kernel is FP64-VALU bound, not HBM bound",
                },
                "kernel": kernel_name,
                "avg_launch_ms": avg_ms,
                "launches": launches,
                "paths_per_launch": paths_per_launch,
                "flop_per_path": flop_per_path,
            }
            if wavefront:
                # The stages of the wavefront pipeline answer almost every body test of a 10 000-body scan in
                # float (bounding groups, member screens), so FP64 flops are the wrong yardstick.  The bound
                # claimed is VALU issue: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction.
                out["roofline"]["brute_force_equivalent_TFLOPs"] = achieved
                peak_issue = 1024 * 2.4e9 / 4.0 / 1e9  # G wave-instructions per second
                if pmc and pmc.get("valu_wave_insts_per_path"):
                    rate = (paths / elapsed) * pmc["valu_wave_insts_per_path"] / 1e9
                    out["roofline"].update({
                        "bound": "valu_issue", "achieved": rate, "peak": peak_issue, "unit": "G wave-instr/s",
                        "frac": rate / peak_issue, "frac_of_no_fma_ceiling": None,
                        "valu_wave_insts_per_path": pmc["valu_wave_insts_per_path"],
                        "active_lanes_of_64": pmc.get("active_lanes"),
                        "fp64_share_of_valu_issue": pmc.get("fp64_share_of_valu_issue"),
                        "source": pmc.get("source"),
                        "note": "VALU instructions issued per path (counter passes of the same workload) times the "
                                "measured whole-pipeline path rate, against the VALU issue peak; "
                                "brute_force_equivalent_TFLOPs prices every body test of every scan at its FP64 cost",
                    })
                else:
                    out["roofline"].update({"bound": "valu_issue", "achieved": None, "peak": peak_issue,
                                            "unit": "G wave-instr/s", "frac": None, "frac_of_no_fma_ceiling": None})
            if pmc and pmc.get("fp64_flop_per_path") and not wavefront:
                # what the FP64 units really executed (counter passes of the same kernel), at this run's rate
                rate = paths_per_launch / (avg_ms * 1e-3)
                ex = rate * pmc["fp64_flop_per_path"] / 1e12
                out["roofline"]["executed"] = {
                    "fp64_TFLOPs": ex,
                    "frac_of_peak": ex / FP64_VALU_PEAK_TFLOPS,
                    "fp64_share_of_valu_issue": pmc.get("fp64_share_of_valu_issue"),
                    "valu_instructions_per_path": pmc.get("valu_insts_per_path"),
                    "active_lanes_of_64": pmc.get("active_lanes"),
                    "source": pmc.get("source"),
                }
            if not wavefront:
                out["roofline"]["frac_kind"] = "algorithmic-equivalent (brute-force body tests at FP64 cost); executed.frac_of_peak is what the FP64 units did"
            if world > 1:
                out["roofline"]["note"] += "; rank 0's kernel launches, per-GPU rate"
            if world == 1 and not proxy and as_configured and args.config == "C2":
                # the same scene with the direct term live (light colour 1): every shadow scan and fake-BRDF
                # bounce is executed, nothing is proven dead — the rate of the kernel family without the
                # compact next-event form, measured on a short run beside the headline
                lit = pt.Renderer(pt.preset("nee15_lit", W, H, seed=1))
                lit_it = 240
                acc2 = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
                lit.render_device(0, 21, acc2.data_ptr(), stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                lit.render_device(0, lit_it, acc2.data_ptr(), stream)
                torch.cuda.synchronize()
                lit_rate = float(W) * H * lit_it / (time.perf_counter() - t1)
                lit.close()
                out["config"]["direct_term_live"] = {
                    "scene": "nee15_lit", "iterations": lit_it, "paths_per_s": lit_rate,
                    "algorithmic_frac": lit_rate * flop_per_path / 1e12 / FP64_VALU_PEAK_TFLOPS,
                }
            if world == 1 and not proxy and not args.no_cpu:
                cb = cpu_baseline(desc, scene, W, H, [t for t in tables] if tables else None, args.cpu_seconds)
                out["cpu_baseline"] = cb
                out["config"]["gpu_over_cpu"] = out["value"] / cb["value"]
                out["config"]["gpu_over_cpu_all_host_cores_extrapolated"] = out["value"] / cb["all_host_cores_extrapolated"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
