#!/bin/bash
# Development only: bench the duplicate-work builds under scripts/ko (`make -C path_tracer_amd/csrc dup`:
# each executes one part of the estimator twice and discards the duplicate, images unchanged) to see
# what that part costs in the real kernel.   scripts/ko_bench.sh <tag> [bench args]
set -u
out=gpurun_out/ko_$1; shift
mkdir -p $out
timeout -k 10 120 python3 bench.py --no-cpu --iterations 1200 "$@" > $out/base.json 2> $out/base.err || exit 1
for f in scripts/ko/libptgpu_ko_*.so; do
  v=$(basename $f .so); v=${v#libptgpu_ko_}
  PTGPU_LIB=$f timeout -k 10 120 python3 bench.py --no-cpu --iterations 1200 "$@" > $out/$v.json 2> $out/$v.err || { echo "$v failed"; tail -3 $out/$v.err; exit 1; }
done
python3 - $out <<'PY'
import json,sys,glob,os
d=sys.argv[1]
base=json.load(open(os.path.join(d,"base.json")))
b=base["roofline"]["avg_launch_ms"]
print("base avg_launch_ms %.4f  value %.4g" % (b, base["value"]))
for f in sorted(glob.glob(os.path.join(d,"*.json"))):
    if f.endswith("base.json"): continue
    j=json.load(open(f))
    t=j["roofline"]["avg_launch_ms"]
    print("%-8s avg_launch_ms %.4f  saves %5.1f %%  mean_radiance %.4f" % (os.path.basename(f)[:-5], t, 100*(b-t)/b, j["config"]["mean_radiance"]))
PY
