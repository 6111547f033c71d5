"""Development: per kernel, HBM bytes read and written per dispatch (FETCH_SIZE doubled as gfx950 needs,
WRITE_SIZE, both in KiB), the time GRBM_GUI_ACTIVE gives and the L2 hit rate, from rocprofv3 --pmc output.
    python scripts/traffic_report.py <dir>"""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k, c = row["Kernel_Name"][:48], row["Counter_Name"]
        tot[k][c] += float(row["Counter_Value"])
        n[k][c] += 1
for k, t in tot.items():
    calls = n[k]["FETCH_SIZE"]
    if not calls:
        continue
    ms = t["GRBM_GUI_ACTIVE"] / 8 / calls / 2.4e6
    rd, wr = 2 * t["FETCH_SIZE"] * 1024 / calls, t["WRITE_SIZE"] * 1024 / max(1, n[k]["WRITE_SIZE"])
    hit = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"] + 1)
    print(k.ljust(50), "%3d calls  ~%.3f ms  read %6.0f MB  written %6.0f MB  %.2f TB/s  L2 hit %.2f" % (calls, ms, rd / 1e6, wr / 1e6, (rd + wr) / ms / 1e9, hit))
