for b in 0 12 13 14 15; do
  for cs in 2:256 5:256; do
    c=${cs%%:*}; spp=${cs##*:}
    PTAMD_WALK_BUDGET=$b PTAMD_LIB=path-tracer_amd/dev/budget.so timeout -k 10 200 python3 bench.py --config $c --spp $spp --no-cpu --steps 3 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; u = r['lane_utilisation']
        print('budget $b config $c  %8.1f Msamples/s  %8.2f ms  nodes/ray %.2f tris/ray %.2f  busy node %.3f tri %.3f shading %.3f' % (d['value'], d['kernel_ms_per_launch'], r['executed_per_ray']['nodes'], r['executed_per_ray']['tris'], u['node_loop'], u['triangle_loop'], u['shading']))
" || exit $?
  done
done
