#!/bin/bash
# Regenerates everything under profiles/ that is measured (run through gpurun from the repo root, then copy the files named
# below from gpurun_out/ into profiles/ with the round prefix):
#   bench JSON (default workload = 4K 10-bit closed GOPs), rocprofv3 --kernel-trace --stats for three workloads, the per-row
#   stage table, PMC passes of the default bench (tools/prof_pmc.sh) — counters in their own runs, as the pool requires.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
python3 $ROOT/bench.py > $ROOT/gpurun_out/${TAG}_bench_4k10-gop.json 2> $ROOT/gpurun_out/${TAG}_bench.err || exit 1
python3 $ROOT/tools/bench_stages.py --json $ROOT/gpurun_out/${TAG}_stages.json > $ROOT/gpurun_out/${TAG}_stages.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for w in 4k10-gop 1080p8-gop 1080p8; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_kt_$w -- python3 $ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-e2e \
    > $ROOT/gpurun_out/${TAG}_bench_${w}_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}_kt_$w.err || exit 1
done
# the GOP session with the GPU tile entropy coder (the product path of RunTranscode): all kernels of an end-to-end batch
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_kt_session -- python3 $ROOT/tools/prof_session.py --segs 8 --frames 16 \
  > $ROOT/gpurun_out/${TAG}_session.log 2> $ROOT/gpurun_out/${TAG}_kt_session.err || exit 1
cd $ROOT && bash tools/prof_pmc.sh $TAG "--steps 1 --warmup 1 --no-cpu-baseline --no-e2e"
