#!/bin/bash
# Everything the large-scene (wavefront) entries of profiles/ are refreshed from: kernel-trace statistics,
# the VALU / FP64 counter passes over all k_wf_* kernels of one batch, HBM bytes per kernel.
#   scripts/c5_all.sh <tag>        then   python scripts/c5_install.py <tag> <prefix>
set -u
tag=$1
export TMPDIR=/tmp
scripts/c5_prof.sh $tag || exit 1
out=gpurun_out/c5_$tag
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $out/pmc/f64 -o f -- python3 bench.py --config C5 --iterations 85 --steps 1 --warmup 0 --no-cpu > $out/pmc_f64.log 2>&1 || { tail -5 $out/pmc_f64.log; exit 1; }
scripts/c5_traffic.sh $tag > gpurun_out/c5tr_$tag.txt || exit 1
cat gpurun_out/c5tr_$tag.txt | cut -c1-140
