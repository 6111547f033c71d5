export TMPDIR=/tmp
mkdir -p gpurun_out/probe
for sc in nee15 main32; do
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/probe/$sc -o p -- python3 scripts/cost_probe.py $sc > gpurun_out/probe/$sc.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv,glob,collections
for sc in ("nee15","main32"):
    tot=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/probe/{sc}/**/*counter_collection.csv",recursive=True):
        for row in csv.DictReader(open(f)):
            tot[(row["Kernel_Name"][:60],row["Dispatch_Id"])][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,v in sorted(tot.items(), key=lambda kv:int(kv[0][1])):
        if v.get("SQ_WAVES"): print(sc,k[0],k[1],"valu/wave %.1f"%(v["SQ_INSTS_VALU"]/v["SQ_WAVES"]), "waves",v["SQ_WAVES"])
PY
