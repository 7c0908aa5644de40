#!/bin/bash
# Counter passes over one launch of a bench configuration (GPU box, repo root): tools/pmc_config.sh <tag> <config> <spp>
# Each --pmc set is a run of its own (rocprofv3 does not split counters over passes); the first render_kernel dispatch
# of a run is the timed launch, the second the counting launch. Environment variables such as PTAMD_NO_WIDE pass through.
tag=${1:-run}; cfg=${2:-3}; spp=${3:-64}
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy VALUUtilization SALUBusy" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set -d $out/p$i -o p --output-format csv -- python3 bench.py --config $cfg --no-cpu --steps 1 --warmup 0 --spp $spp > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $out/p1 $out/p2 $out/p3 $out/p4 $out/p5 $out/p6 > $out/summary.txt
cat $out/summary.txt
