// Host BVH builder for one mesh. Replaces Mesh::BuildBVH / BuildNode / RangeBounds (Mesh.cpp:289-397):
// the reference splits at the object median with 2-triangle leaves; this builder uses a binned surface
// area heuristic, leaves of up to 4 triangles and a hard depth bound that matches the traversal stack of
// the kernels. The tree topology differs from the reference's; the closest hit does not depend on it.
#pragma once
#include <cstdint>
#include <vector>
#include "pt_types.h"

namespace ptamd {

struct BvhResult {
    std::vector<DNode> nodes;       // inner nodes only; children are encoded references (see DNode)
    std::vector<uint32_t> order;    // leaf order -> original face index
    int32_t root = 0;               // reference of the root (negative when the whole mesh is one leaf)
    int depth = 0;                  // number of inner-node levels (bounds the traversal stack)
};

// positions: num_faces * 9 floats (a, b, c). max_depth: inner-node levels allowed.
BvhResult build_bvh(const float* positions, uint32_t num_faces, int max_depth);

} // namespace ptamd
