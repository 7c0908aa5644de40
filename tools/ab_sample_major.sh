for nsm in "" 1; do
  for cs in 3:64 4:64 4:32 2:1024; do
    c=${cs%%:*}; spp=${cs##*:}
    if [ -n "$nsm" ]; then export PTAMD_NO_SAMPLE_MAJOR=1; else unset PTAMD_NO_SAMPLE_MAJOR; fi
    timeout -k 10 200 python3 bench.py --config $c --spp $spp --no-cpu --steps 3 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; u = r['lane_utilisation']
        print('nsm=$nsm config $c spp $spp %8.1f Msamples/s  %8.2f ms  nodes/ray %.2f tris/ray %.2f  busy node %.3f tri %.3f  %s' % (d['value'], d['kernel_ms_per_launch'], r['executed_per_ray']['nodes'], r['executed_per_ray']['tris'], u['node_loop'], u['triangle_loop'], r.get('kernel')))
" || exit $?
  done
done
