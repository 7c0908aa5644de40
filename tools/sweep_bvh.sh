for lm in 4 8; do for ct in 0.3 0.6 1.2; do
echo "== leaf_max $lm cost_tri $ct"
PTAMD_LEAF_MAX=$lm PTAMD_COST_TRI=$ct timeout -k 10 200 python3 tools/kbench.py --spp 64 --reps 3 --scenes mesh_cornell,blob_mixed 2>&1 | grep -E "Msamples|per ray" | cut -c1-150
done; done
