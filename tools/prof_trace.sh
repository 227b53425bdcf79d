#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py (run through gpurun from the repo root).  Output: gpurun_out/<tag>_kt_<workload>/
# usage: tools/prof_trace.sh <tag> <workload> [extra bench args]
set -u
TAG=${1:-rXX}; W=${2:-4k10-gop}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_kt_$W -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" \
  > $ROOT/gpurun_out/${TAG}_bench_${W}_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}_kt_$W.err || { tail -5 $ROOT/gpurun_out/${TAG}_kt_$W.err; exit 1; }
ls -la $ROOT/gpurun_out/${TAG}_kt_$W/*/ | head
