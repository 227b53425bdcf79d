ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  if [ $v = old ]; then export AV1MI_LIB=$ROOT/tools/_diag/libav1mi_old.so; else unset AV1MI_LIB; fi
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $ROOT/gpurun_out/pmc_ab/$v -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_ab_$v.log 2>&1 || echo fail $v
done
