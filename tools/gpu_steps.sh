#!/bin/bash
# Runs GPU steps one after the other on the GPU box, each under its own time limit, logging to gpurun_out/<name>.log.
# A step that fails in the ordinary way (exit 1..123) does not stop the sequence; a step that hits its time limit,
# aborts or crashes (exit >= 124) does: nothing more is started on the GPU after that.
#   tools/gpu_steps.sh name1 seconds1 'command 1' name2 seconds2 'command 2' ...
mkdir -p gpurun_out
export TMPDIR=/tmp
while [ $# -ge 3 ]; do
  name=$1; secs=$2; cmd=$3; shift 3
  echo "=== $name (limit $secs s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name exit $rc"; tail -n 12 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "=== stopping: $name ended with $rc"; exit $rc; fi
done
exit 0
