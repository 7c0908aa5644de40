#!/bin/bash
# A/B counter collection: tools/pmc_ab.sh <tag> [PTAMD_LIB path]   (run on the GPU box, from the repo root)
set -e
tag=$1
export PTAMD_LIB=$2
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -k 10 120 python3 tools/kbench.py --spp 64 --reps 3 --scenes mesh_cornell > $out/time.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $out/A -o a --output-format csv -- python3 tools/kbench.py --spp 64 --reps 1 --scenes mesh_cornell > $out/a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -d $out/B -o b --output-format csv -- python3 tools/kbench.py --spp 64 --reps 1 --scenes mesh_cornell > $out/b.log 2>&1
python3 tools/pmc_summary.py $out/A $out/B > $out/summary.txt
grep mesh_cornell $out/time.log
