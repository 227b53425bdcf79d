#!/bin/bash
# PMC passes over tools/ent_probe.py (K9 alone) on the GPU box.  Output: gpurun_out/pmc_ent/pass*/...
set -u
CASE=${1:-"48,64,128"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc_ent/pass$i -- python3 $ROOT/tools/ent_probe.py $CASE \
    > $ROOT/gpurun_out/pmc_ent_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $ROOT/gpurun_out/pmc_ent_pass$i.log; exit 1; }
  echo "pass $i ok"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$ROOT/gpurun_out/pmc_ent/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("av1mi::", "").replace("(anonymous namespace)::", "")
        if not k.startswith("k_ent"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-24s %14.0f per launch" % (c, v / cnt[(k, c)]))
PY
