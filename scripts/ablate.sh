#!/bin/bash
# time the bench with alternative builds of the library: scripts/ablate.sh TAG1 TAG2 ...
cp path_tracer_amd/libptgpu.so /tmp/libptgpu_orig.so
for v in "$@"; do
  cp path_tracer_amd/libptgpu_$v.so path_tracer_amd/libptgpu.so
  echo -n "$v: "; timeout -k 10 200 python3 bench.py --iterations 420 --steps 2 --warmup 1 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['config']['mean_radiance'])"
done
cp /tmp/libptgpu_orig.so path_tracer_amd/libptgpu.so
