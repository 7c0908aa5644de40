#!/bin/bash
# Utilisation counters of the render kernel on the bench scene (GPU box, repo root): tools/pmc_util.sh <tag>
# Each --pmc pass is a run of its own; summaries land in gpurun_out/pmcu_<tag>/summary.txt
tag=${1:-run}
export TMPDIR=/tmp
out=gpurun_out/pmcu_$tag
mkdir -p $out
i=0
for set in "VALUBusy VALUUtilization SALUBusy" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $out/p$i -o p --output-format csv -- python3 tools/kbench.py --spp 64 --reps 1 --scenes ${SCENE:-mesh_cornell} > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $out/p1 $out/p2 $out/p3 $out/p4 > $out/summary.txt
cat $out/summary.txt
