"""Per-path figures of the estimator kernel from the counter passes of scripts/pmc.sh.

    python scripts/pmc_summary.py <scene>=<gpurun_out/pmc_dir> [...]  > profiles/pmc_summary.json

bench.py reads the file and reports, next to the algorithmic roofline, what the FP64 units really
executed.  Counter values are chip totals per dispatch; one wave traces 64 paths.
FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes,
so it is doubled.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def totals(root, pat="k_trace"):
    tot = defaultdict(float)
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat not in row.get("Kernel_Name", ""):
                    continue
                c = row["Counter_Name"]
                tot[c] += float(row["Counter_Value"])
                disp[c].add((f, row.get("Dispatch_Id")))
    return {c: tot[c] / max(1, len(disp[c])) for c in tot}


def summary(root):
    t = totals(root)
    waves = t["SQ_WAVES"]
    paths = 64.0 * waves
    valu = t["SQ_INSTS_VALU"]
    lanes = t["SQ_THREAD_CYCLES_VALU"] / valu
    fp64 = t["SQ_INSTS_VALU_ADD_F64"] + t["SQ_INSTS_VALU_MUL_F64"] + t["SQ_INSTS_VALU_FMA_F64"] + t["SQ_INSTS_VALU_TRANS_F64"]
    flop_insts = fp64 + t["SQ_INSTS_VALU_FMA_F64"]  # FMA = 2 flop
    out = {
        "source": f"rocprofv3 --pmc passes of scripts/pmc.sh ({os.path.basename(root.rstrip('/'))}), per dispatch of {int(paths)} paths",
        "valu_insts_per_path": valu / waves,  # instructions a lane's wave executes for it
        "active_lanes": lanes,
        "fp64_share_of_valu_issue": fp64 / valu,
        "fp64_flop_per_path": flop_insts / waves * lanes / 64.0,
        "fp64_flop_per_path_all_lanes": flop_insts / waves,
        "int32_share": t.get("SQ_INSTS_VALU_INT32", 0.0) / valu,
        "cvt_share": t.get("SQ_INSTS_VALU_CVT", 0.0) / valu,
        "salu_per_valu": t.get("SQ_INSTS_SALU", 0.0) / valu,
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
        "valu_busy": 4.0 * t["SQ_ACTIVE_INST_VALU"] / (1024.0 * t["GRBM_GUI_ACTIVE"] / 8.0) if "GRBM_GUI_ACTIVE" in t else None,
        "cycles_per_valu_inst": (1024.0 * t["GRBM_GUI_ACTIVE"] / 8.0) / valu if "GRBM_GUI_ACTIVE" in t else None,
        "wait_inst_any_of_wave_cycles": t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"],
    }
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        out["fetch_kb_per_dispatch_x2"] = 2.0 * t["FETCH_SIZE"]
        out["write_kb_per_dispatch"] = t["WRITE_SIZE"]
        out["bytes_per_path"] = (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0 / paths
    if "TCC_HIT_sum" in t:
        out["l2_hit_rate"] = t["TCC_HIT_sum"] / max(1.0, t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
    return out


def sums(root, pat):
    """Counter totals over every dispatch whose kernel name contains `pat`."""
    tot = defaultdict(float)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row.get("Kernel_Name", ""):
                    tot[row["Counter_Name"]] += float(row["Counter_Value"])
    return tot


def summary_pipeline(root, pat, paths):
    """A pipeline of several kernels (the wavefront stages): totals over all of them, per path of the run."""
    t = sums(root, pat)
    valu = t["SQ_INSTS_VALU"]
    fp64 = t["SQ_INSTS_VALU_ADD_F64"] + t["SQ_INSTS_VALU_MUL_F64"] + t["SQ_INSTS_VALU_FMA_F64"] + t["SQ_INSTS_VALU_TRANS_F64"]
    flop_insts = fp64 + t["SQ_INSTS_VALU_FMA_F64"]
    lanes = t["SQ_THREAD_CYCLES_VALU"] / valu
    return {
        "source": f"rocprofv3 --pmc passes of scripts/pmc.sh ({os.path.basename(root.rstrip('/'))}), all {pat}* kernels of a run of {int(paths)} paths",
        "valu_wave_insts_per_path": valu / paths,  # wave instructions (64 lanes each) issued per path
        "active_lanes": lanes,
        "fp64_share_of_valu_issue": fp64 / valu,
        "fp64_flop_per_path": flop_insts * lanes / paths,
        "waves_per_path": t["SQ_WAVES"] / paths,
    }


def main():
    res = {}
    for arg in sys.argv[1:]:
        scene, root = arg.split("=", 1)
        if root.count(":") == 2:  # dir:kernel-pattern:paths
            d, pat, paths = root.split(":")
            res[scene] = summary_pipeline(d, pat, float(paths))
            continue
        res[scene] = summary(root)
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
