#!/bin/bash
# PMC passes over the closed-GOP workload (k_me_int, k_inter_pipe).  Output: gpurun_out/pmc_gop/pass*/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc_gop/pass$i -- python3 $ROOT/bench.py --workload 1080p8-gop --segments 4 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e \
    > $ROOT/gpurun_out/pmc_gop_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $ROOT/gpurun_out/pmc_gop_pass$i.log; exit 1; }
  echo "pass $i ok"
done
