#!/bin/bash
# Round profile of one bench configuration (run on the GPU box from the repo root): the bench line, the rocprofv3
# kernel-trace summary of the same command, and the memory-side byte counters of the render kernel in two separate
# counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass). Outputs land in gpurun_out/prof_<tag>/.
#   tools/profile_bench.sh <tag> [config] [more bench.py arguments]
# A scene whose last object is a mesh may measure its schedule (walk cut, deferred walk) with ten short launches of the
# render kernel before its first frame; the first bench run below reports what it chose, and the profiler runs are given
# that choice through PTAMD_WALK_BUDGET / PTAMD_WALK_DEFER so that their kernel statistics hold the frame launches only.
set -e
tag=${1:-run}
cfg=${2:-2}
shift; shift || true
extra="$*"
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py --config $cfg $extra > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
chosen=$(PTAMD_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg $extra --no-cpu --no-others --steps 1 --warmup 0 2>&1 >/dev/null | sed -n 's/.*ptamd: \(walk cut\|deferred walk\) measured.*-> \([0-9]*\)$/\1=\2/p' | head -1)
case "$chosen" in
  "walk cut="*) export PTAMD_WALK_BUDGET=${chosen#*=}; echo "walk cut measured by the library: ${chosen#*=}/64 (fixed for the profiler runs)";;
  "deferred walk="*) export PTAMD_WALK_DEFER=${chosen#*=}; echo "deferred walk measured by the library: ${chosen#*=} lanes (fixed for the profiler runs)";;
esac
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --config $cfg $extra --no-cpu --steps 2 --warmup 1 > $out/trace.log 2>&1
cp $out/trace/t_kernel_stats.csv $out/kernel_stats.csv
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- python3 bench.py --config $cfg $extra --no-cpu --steps 1 --warmup 0 > $out/fetch.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- python3 bench.py --config $cfg $extra --no-cpu --steps 1 --warmup 0 > $out/write.log 2>&1
python3 tools/pmc_summary.py $out/fetch $out/write > $out/hbm_counters.txt
cat $out/hbm_counters.txt
python3 tools/hbm_traffic.py $out $cfg > $out/hbm_traffic.json
cat $out/hbm_traffic.json
