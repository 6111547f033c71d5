#!/bin/bash
# Bench lines of the path-tracing scenes for profiles/ (run on the GPU box from the repository root).
#   scripts/pt_lines.sh <tag>
set -u
out=gpurun_out/prof_$1
mkdir -p $out
timeout -k 10 200 python3 bench.py --no-cpu --scene main32 > $out/bench_main32.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene raytrace3 > $out/bench_raytrace3.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene light_room32 --iterations 720 > $out/bench_light_room32.json 2>> $out/bench.err || exit 1
for f in main32 raytrace3 light_room32; do python3 -c "import json; d=json.load(open('$out/bench_$f.json')); print('$f %.4g frac %.4f' % (d['value'], d['roofline']['frac']))"; done
