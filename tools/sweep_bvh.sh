for lm in 2 3 4; do for ct in 0.8 1.2 2.0 3.0; do
echo "== leaf_max $lm cost_tri $ct"
PTAMD_LEAF_MAX=$lm PTAMD_COST_TRI=$ct timeout -k 10 200 python3 tools/kbench.py --spp 64 --reps 2 --scenes mesh_cornell 2>&1 | grep -E "Msamples|per ray" | cut -c1-120
done; done
