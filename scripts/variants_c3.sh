#!/bin/bash
# Development: bench the measured-material workload with every library under scripts/ko (variant builds).
#   scripts/variants_c3.sh [iterations]
set -u
it=${1:-720}
for L in scripts/ko/*.so; do
  v=$(basename $L .so)
  PTGPU_LIB=$L timeout -k 10 120 python3 bench.py --config C3 --iterations $it --steps 3 --warmup 1 --no-cpu > gpurun_out/y_$v.json 2> gpurun_out/y_$v.err || { echo "$v failed"; tail -3 gpurun_out/y_$v.err; exit 1; }
  python3 -c "import json;j=json.load(open('gpurun_out/y_$v.json'));print('$v', '%.4g' % j['value'], j['config'].get('mean_radiance'))"
done
