#!/bin/bash
# Collect hardware counters for the estimator kernel, one rocprofv3 --pmc pass per counter group
# (each pass is its own run; no tracing options are combined with --pmc).
#   scripts/pmc.sh <tag> [bench args...]
set -u
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --no-cpu "${BENCH_ARGS[@]}" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
BENCH_ARGS=("$@")
[ ${#BENCH_ARGS[@]} -eq 0 ] && BENCH_ARGS=(--iterations 40 --steps 1 --warmup 0)
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run p2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU &&
run p3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_IFETCH &&
run p4 GRBM_GUI_ACTIVE FETCH_SIZE &&
run p5 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
run p6 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum &&
python3 scripts/pmc_report.py $out
