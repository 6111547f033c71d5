"""Copy the results of scripts/c5_all.sh (run with one tag on the GPU box) into profiles/ under a round
prefix and refresh the `synthetic10k` entry of profiles/pmc_summary.json.
    python scripts/c5_install.py <tag> <prefix>"""
import json
import os
import re
import shutil
import subprocess
import sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
PATHS = 85 * 1024 * 768  # one batch: the counter passes render 85 iterations of 1024x768
shutil.copy(os.path.join(out, f"c5_{tag}", "trace", "t_kernel_stats.csv"), os.path.join(prof, f"{prefix}_kernel_stats_C5_wavefront.csv"))
shutil.copy(os.path.join(out, f"c5tr_{tag}.txt"), os.path.join(prof, f"{prefix}_traffic_C5_wavefront.txt"))
rep = subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_report.py"), os.path.join(out, f"c5_{tag}", "pmc"), "k_wf_"],
                     capture_output=True, text=True).stdout
open(os.path.join(prof, f"{prefix}_pmc_wavefront_c5.txt"), "w").write(rep)
new = json.loads(subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_summary.py"),
                                 f"synthetic10k={out}/c5_{tag}/pmc:k_wf_:{PATHS}"], capture_output=True, text=True).stdout)["synthetic10k"]
total = 0.0
for line in open(os.path.join(out, f"c5tr_{tag}.txt")):
    m = re.search(r"(\d+) calls .* read +(\d+) MB +written +(\d+) MB", line)
    if m and "at::native" not in line:
        total += int(m.group(1)) * (int(m.group(2)) + int(m.group(3))) * 1e6
new["bytes_per_path"] = total / PATHS
new["source"] = (f"rocprofv3 --pmc passes of scripts/c5_all.sh (c5_{tag}, c5tr_{tag}), all k_wf_* kernels of a batch of {PATHS} paths; "
                 "bytes: every kernel of the batch, the sorts included")
path = os.path.join(prof, "pmc_summary.json")
summary = json.load(open(path))
summary["synthetic10k"] = new
json.dump(summary, open(path, "w"), indent=1)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in new.items() if k != "source"})
