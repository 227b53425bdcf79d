#!/bin/bash
# copies the files tools/prof_all.sh <tag> left under gpurun_out/ into profiles/ with the round prefix
# usage: tools/install_profiles.sh <tag> <round prefix, e.g. r01>
set -e
T=$1; R=$2
python3 tools/pmc_to_json.py $T 12 profiles/pmc_4k10_gop_latest.json > /dev/null
cp profiles/pmc_4k10_gop_latest.json profiles/${R}_pmc_4k10_gop.json
cp gpurun_out/${T}_bench_4k10-gop.json profiles/${R}_bench_4k10_gop.json
cp "$(ls -t gpurun_out/${T}_kt_4k10-gop/*/*_kernel_stats.csv | head -1)" profiles/${R}_bench_4k10_gop_kernel_stats.csv
cp gpurun_out/${T}_bench_4k10-gop_under_rocprof.json profiles/${R}_bench_4k10_gop_under_rocprof.json
cp "$(ls -t gpurun_out/${T}_kt_1080p8-gop/*/*_kernel_stats.csv | head -1)" profiles/${R}_bench_1080p8_gop_kernel_stats.csv
cp gpurun_out/${T}_bench_1080p8-gop_under_rocprof.json profiles/${R}_bench_1080p8_gop_under_rocprof.json
cp "$(ls -t gpurun_out/${T}_kt_1080p8/*/*_kernel_stats.csv | head -1)" profiles/${R}_bench_1080p8_intra_kernel_stats.csv
cp gpurun_out/${T}_bench_1080p8_under_rocprof.json profiles/${R}_bench_1080p8_intra_under_rocprof.json
cp gpurun_out/${T}_stages.json profiles/${R}_stages.json
cp "$(ls -t gpurun_out/${T}_kt_session/*/*_kernel_stats.csv | head -1)" profiles/${R}_session_gpu_entropy_kernel_stats.csv
cp gpurun_out/${T}_session.log profiles/${R}_session_gpu_entropy.log
