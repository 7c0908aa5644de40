#!/bin/bash
# Sweep of the builder's leaf size and SAH primitive cost on the headline scene (GPU box, repo root).
for lm in 1 2 3 4; do for ct in 0.5 0.8 1.2 2.0; do
echo "== leaf_max $lm cost_tri $ct"
PTAMD_LEAF_MAX=$lm PTAMD_COST_TRI=$ct timeout -k 10 200 python3 tools/kbench.py --spp 64 --reps 3 --scenes mesh_cornell 2>&1 | grep -E "Msamples|per ray" | cut -c1-120
done; done
