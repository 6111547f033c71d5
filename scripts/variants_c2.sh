#!/bin/bash
# Development: headline, direct-term-live and plain path-tracing rates of every library under scripts/ko.
set -u
for L in scripts/ko/*.so; do
  v=$(basename $L .so)
  PTGPU_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu --steps 2 --warmup 1 > gpurun_out/v_$v.json 2> gpurun_out/v_$v.err || { tail -3 gpurun_out/v_$v.err; exit 1; }
  PTGPU_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu --steps 1 --warmup 1 --scene main32 > gpurun_out/v_${v}_main32.json 2>> gpurun_out/v_$v.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/v_$v.json')); m=json.load(open('gpurun_out/v_${v}_main32.json'))
print('$v', 'C2 %.4g'%d['value'], 'launch %.4f ms'%d['roofline']['avg_launch_ms'], 'lit %.4g'%d['config']['direct_term_live']['paths_per_s'], 'main32 %.4g'%m['value'])"
done
