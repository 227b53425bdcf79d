set -e
tools/prof_trace.sh r03f 4k10-gop
AV1MI_CODER_STREAMS=main tools/prof_trace.sh r03fser 4k10-gop
tools/prof_trace.sh r03f 1080p8-gop
tools/prof_trace.sh r03f 1080p8
tools/prof_pmc.sh r03f
python bench.py --workload 1080p8 --no-cpu-baseline > gpurun_out/r03f_1080intra_k32.json 2> gpurun_out/r03f_1080intra_k32.err
python bench.py --workload 1080p8 --no-cpu-baseline --key-block-size 8 > gpurun_out/r03f_1080intra_k8.json 2> gpurun_out/r03f_1080intra_k8.err
echo all done
