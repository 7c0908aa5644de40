# Sweeps the BVH builder's leaf size and primitive cost on the kernel micro-benchmark (GPU box, repo root)
for lm in 1 2 4; do for ct in 0.6 1.2 2.0 4.0; do
echo "== leaf_max $lm cost_tri $ct"
PTAMD_LEAF_MAX=$lm PTAMD_COST_TRI=$ct timeout -k 10 200 python3 tools/kbench.py --spp 64 --reps 3 --scenes ${SCENES:-mesh_cornell,blob_mixed} 2>&1 | grep -E "Msamples|per ray" | cut -c1-150
done; done
