// Host BVH builder for one mesh. Replaces Mesh::BuildBVH / BuildNode / RangeBounds (Mesh.cpp:289-397):
// the reference splits at the object median with 2-triangle leaves; this builder uses a binned surface
// area heuristic, leaves of up to 4 primitives and a hard depth bound that matches the traversal stack of
// the kernels. A primitive is a triangle, or two triangles that form a parallelogram (a "quad": one plane
// test with a square barycentric range covers both). The tree topology differs from the reference's; the
// closest hit does not depend on it.
#pragma once
#include <cstdint>
#include <vector>
#include "pt_types.h"

namespace ptamd {

struct BvhResult {
    std::vector<DNode> nodes;       // inner nodes only; children are encoded references (see DNode)
    std::vector<uint32_t> order;    // triangle slot (leaf order; the two triangles of a quad adjacent) -> original face index
    uint32_t num_quads = 0;
    std::vector<int> rotation;      // per face: which of its vertices (0..2) is the first vertex of its hot record (the
                                    // corner `a` of a quad; 0 for lone triangles and for the second triangle of a quad)
    int32_t root = 0;               // reference of the root (negative when the whole mesh is one leaf)
    int depth = 0;                  // number of inner-node levels (bounds the traversal stack)
    // The same tree collapsed to eight-wide nodes (build_wide): nodes in breadth-first blocks (the inner children of a
    // node are adjacent), node 0 the root; wide_order is the triangle slot order that goes with it (the leaf slots of a
    // node are adjacent), wide_depth the number of node levels. Empty when the mesh is a single leaf.
    std::vector<DWideNode> wide;
    std::vector<uint32_t> wide_order;
    int wide_depth = 0;
};

// positions: num_faces * 9 floats (a, b, c); group: material group per face (quads never mix groups), may be null.
// max_depth: inner-node levels allowed. Leaf reference: ~ref = (first_slot << 6) | (quad_mask << 2) | (prims - 1),
// bit p of quad_mask set when primitive p of the leaf is a quad (it then takes two slots).
BvhResult build_bvh(const float* positions, const int32_t* group, uint32_t num_faces, int max_depth);
// Fills res.wide / wide_order / wide_depth from res.nodes / res.order (the binary tree must be complete).
void build_wide(BvhResult& res);

} // namespace ptamd
