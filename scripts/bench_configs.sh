#!/bin/bash
# One bench line per configured workload at its full size (run on the GPU box from the repository root).
#   scripts/bench_configs.sh <tag> [extra bench args...]
set -u
tag=$1; shift
out=gpurun_out/bench_$tag
mkdir -p $out
for c in C2 C3 C4 C5; do
  timeout -k 10 400 python3 bench.py --config $c "$@" > $out/$c.json 2> $out/$c.err || { echo "bench $c failed"; tail -5 $out/$c.err; exit 1; }
  echo "$c done"; cut -c1-400 $out/$c.json
done
