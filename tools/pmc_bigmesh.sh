#!/bin/bash
# Memory-side counters of the render kernel on the 1 M-triangle scene (GPU box, repo root).
export TMPDIR=/tmp
out=gpurun_out/pmc_bigmesh
mkdir -p $out
timeout -k 10 200 python3 tools/kbench.py --spp 16 --reps 3 --scenes bumpy_1m > $out/time.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy VALUUtilization"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set -d $out/p$i -o p --output-format csv -- python3 tools/kbench.py --spp 16 --reps 1 --scenes bumpy_1m > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $out/p1 $out/p2 $out/p3 $out/p4 > $out/summary.txt
cat $out/time.log | grep -E "per ray|Msamples"
cat $out/summary.txt
