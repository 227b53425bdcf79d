#!/bin/bash
# PMC passes of bench.py on the GPU box (run through gpurun).  Each pass is its own rocprofv3 run with --pmc only
# (no sys/hip/hsa tracing), as the pool requires.  Output: gpurun_out/pmc_<tag>/*counter_collection.csv
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 1 --warmup 1 --no-cpu-baseline --no-e2e"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# per-kernel counters want every kernel ALONE on the GPU: the coder serialised behind the filters on the main stream
export AV1MI_CODER_STREAMS=${AV1MI_CODER_STREAMS:-main}
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}/pass$i -- python3 $ROOT/bench.py $ARGS \
    > $ROOT/gpurun_out/pmc_${TAG}_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $ROOT/gpurun_out/pmc_${TAG}_pass$i.log; exit 1; }
  echo "pass $i ok"
done
