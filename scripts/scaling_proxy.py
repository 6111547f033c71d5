"""Single-GPU proxy of strong scaling: what rank 0 of N ranks does, measured alone on one GPU.

    python scripts/scaling_proxy.py <tag> [C2 C4 ...]        (on the GPU box, from the repository root)

For every configuration and N in 1, 2, 4, 8 it runs `bench.py --proxy-world N` (rank 0's interleaved
tiles of a world of N and then every other rank's, one after the other, the full iteration count,
no exchange; T(N) is the slowest rank's time) once plainly and once under
`rocprofv3 --kernel-trace --stats`, and writes gpurun_out/scaling_<tag>/<config>.json:
T(N), efficiency = T(1) / (N T(N)), and the per-kernel average durations of the profiled run.
The exchange is not part of the proxy: it moves W*H*3/N doubles per rank once per render.
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
configs = sys.argv[2:] or ["C2", "C4"]
out_dir = os.path.join(root, "gpurun_out", f"scaling_{tag}")
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
steps = {"C2": 3, "C4": 1}


def run(cmd, log):
    with open(log, "w") as err:
        p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=err, text=True, timeout=600)
    if p.returncode != 0:
        print("failed:", " ".join(cmd), "see", log, flush=True)
        sys.exit(1)
    return p.stdout


for cfg in configs:
    rows = []
    for n in (1, 2, 4, 8):
        base = ["bench.py", "--config", cfg, "--proxy-world", str(n), "--no-cpu", "--steps", str(steps.get(cfg, 1)), "--warmup", "1"]
        line = json.loads(run([sys.executable] + base, os.path.join(out_dir, f"{cfg}_n{n}.err")).strip().splitlines()[-1])
        tdir = os.path.join(out_dir, f"{cfg}_n{n}_trace")
        run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tdir, "-o", "trace", "--", sys.executable] + base,
            os.path.join(out_dir, f"{cfg}_n{n}_rocprof.err"))
        kernels = {}
        for f in glob.glob(os.path.join(tdir, "**", "trace_kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"k_\w+(<[^>]*>)?", r["Name"])
                if m:
                    kernels[m.group(0)] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                     "total_ms": float(r["TotalDurationNs"]) / 1e6}
        rows.append({"world": n, "rank_seconds": line["proxy"]["slowest_rank_seconds_per_step"],
                     "every_rank_seconds": line["proxy"]["every_rank_seconds_per_step"],
                     "rank_paths": line["proxy"]["rank_paths_per_step"], "rank_paths_per_s": line["value"],
                     "estimator_avg_launch_ms": line["roofline"]["avg_launch_ms"], "launches": line["roofline"]["launches"],
                     "kernels_profiled_run": kernels})
        print(cfg, "N", n, "T %.4f s" % rows[-1]["rank_seconds"], "%.4g paths/s" % line["value"], flush=True)
    t1 = rows[0]["rank_seconds"]
    for r in rows:
        r["efficiency"] = t1 / (r["world"] * r["rank_seconds"])
        r["speedup"] = t1 / r["rank_seconds"]
    res = {"config": cfg, "what": "every rank of N alone on one MI355X, one after the other, full iteration count, no exchange; "
                   "T(N) = the slowest rank's time; efficiency = T(1)/(N T(N)); kernel statistics: the profiled run of the same command",
           "rows": rows}
    json.dump(res, open(os.path.join(out_dir, f"{cfg}.json"), "w"), indent=1)
    print(cfg, "efficiency", {r["world"]: round(r["efficiency"], 3) for r in rows}, flush=True)
