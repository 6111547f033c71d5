"""Copy the results of scripts/profile_round.sh, bench_configs.sh and pmc.sh (run with one tag on the
GPU box) from gpurun_out/ into profiles/ under a round prefix, and refresh profiles/pmc_summary.json.
    python scripts/install_profiles.py <tag> <prefix>      e.g.  r02f r02_c
"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
for f in glob.glob(os.path.join(out, f"prof_{tag}", "bench_*.json")):
    shutil.copy(f, os.path.join(prof, f"{prefix}_{os.path.basename(f)}"))
shutil.copy(os.path.join(out, f"prof_{tag}", "trace", "trace_kernel_stats.csv"), os.path.join(prof, f"{prefix}_kernel_stats.csv"))
for c in ("C2", "C3", "C4", "C5"):
    shutil.copy(os.path.join(out, f"bench_{tag}", f"{c}.json"), os.path.join(prof, f"{prefix}_config_{c}.json"))
for sub, name in ((f"pmc_{tag}", "pmc_trace_kernel"), (f"pmc_{tag}_meas", "pmc_measured_brdf_kernel")):
    txt = subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_report.py"), os.path.join(out, sub)],
                         capture_output=True, text=True).stdout
    open(os.path.join(prof, f"{prefix}_{name}.txt"), "w").write(txt)
summary_path = os.path.join(prof, "pmc_summary.json")
summary = json.load(open(summary_path))
new = json.loads(subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_summary.py"),
                                 f"nee15={out}/pmc_{tag}", f"nee15_measured={out}/pmc_{tag}_meas"],
                                capture_output=True, text=True).stdout)
summary.update(new)
json.dump(summary, open(summary_path, "w"), indent=1)
for k in ("nee15", "nee15_measured"):
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in summary[k].items() if a != "source"})
for c in ("C2", "C3", "C4", "C5"):
    d = json.load(open(os.path.join(prof, f"{prefix}_config_{c}.json")))
    r, cb = d["roofline"], d.get("cpu_baseline", {})
    print(c, "%.4g" % d["value"], "frac", r["frac"], "of no-FMA", r.get("frac_of_no_fma_ceiling"),
          "achieved %.4g" % (r.get("brute_force_equivalent_TFLOPs") or r["achieved"]), "ms", r["avg_launch_ms"],
          "cpu %.3g" % cb.get("value", 0), "x%.0f" % d["config"].get("gpu_over_cpu", 0))
for f in sorted(glob.glob(os.path.join(prof, f"{prefix}_bench_*.json"))):
    d = json.load(open(f))
    r = d["roofline"]
    print(os.path.basename(f), "%.4g" % d["value"], "rays %.3g" % d["config"]["rays_per_s"], "frac", r.get("frac"),
          "achieved %.4g" % (r.get("brute_force_equivalent_TFLOPs") or r["achieved"] or 0), "ms %.4f" % r["avg_launch_ms"])
