#!/bin/bash
# Development: bench the large-scene workload with every library under scripts/ko (variant builds).
#   scripts/variants_c5.sh [iterations]
set -u
it=${1:-256}
for L in scripts/ko/*.so; do
  v=$(basename $L .so)
  PTGPU_LIB=$L timeout -k 10 120 python3 bench.py --config C5 --iterations $it --steps 2 --warmup 1 --no-cpu > gpurun_out/x_$v.json 2> gpurun_out/x_$v.err || { echo "$v failed"; tail -3 gpurun_out/x_$v.err; exit 1; }
  python3 -c "import json;j=json.load(open('gpurun_out/x_$v.json'));print('$v', '%.4g' % j['value'], j['config'].get('mean_radiance'))"
done
