#!/bin/bash
# A/B of library builds on the bench configurations (GPU box, repo root): CONFIGS="3:64 4:32" tools/abcfg.sh lib1.so lib2.so ...
# Prints one line per build and configuration: Msamples/s, kernel ms, lanes busy in the node and triangle loops.
for lib in "$@"; do
  for cs in ${CONFIGS:-3:64 4:32}; do
    c=${cs%%:*}; spp=${cs##*:}
    PTAMD_LIB=$lib timeout -k 10 200 python3 bench.py --config $c --spp $spp --no-cpu --steps 3 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; u = r['lane_utilisation']
        print('%-40s config $c  %8.1f Msamples/s  %8.2f ms  nodes/ray %.2f tris/ray %.2f  busy node %.3f tri %.3f' % ('$lib', d['value'], d['kernel_ms_per_launch'], r['executed_per_ray']['nodes'], r['executed_per_ray']['tris'], u['node_loop'], u['triangle_loop']))
" || exit $?
  done
done
