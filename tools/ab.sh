#!/bin/bash
# Times the render kernel of several library builds on the same scenes: tools/ab.sh lib1.so lib2.so ... (GPU box, repo root)
for lib in "$@"; do
  echo "== $lib"
  PTAMD_LIB=$lib PTAMD_DEBUG=1 timeout -k 10 200 python3 tools/kbench.py --spp 64 --reps 3 --scenes ${SCENES:-mesh_cornell,glass_mirror,blob_mixed} 2>&1 | grep -E "^ptamd|Msamples|per ray" | sort | uniq
done
