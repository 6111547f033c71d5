#!/bin/bash
# Development: kernel-trace statistics of the large-scene workload for every library under scripts/ko.
set -u
export TMPDIR=/tmp
for L in scripts/ko/*.so; do
  v=$(basename $L .so)
  out=gpurun_out/vt_$v
  mkdir -p $out
  PTGPU_LIB=$L timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --config C5 --iterations 64 --steps 1 --warmup 0 --no-cpu > $out/bench.json 2> $out/log || { tail -3 $out/log; exit 1; }
  echo "== $v"; cut -d, -f1-4 $out/t_kernel_stats.csv | grep "k_wf_bounce\|k_wf_shadow\|k_wf_nee\|k_wf_light\|ray_keys" | sed 's/(ptd[^"]*"/"/' | cut -c1-110
done
