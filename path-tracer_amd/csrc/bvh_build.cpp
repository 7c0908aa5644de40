#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace ptamd {
namespace {

struct Box3 {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); } }
    void grow(const Box3& b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0.f) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16;
uint32_t kLeafMax = 4;
float kCostNode = 1.0f, kCostTri = 1.2f;

struct Builder {
    std::vector<Box3> fb;
    std::vector<float> fc; // 3 per face
    std::vector<uint32_t> idx;
    std::vector<DNode> nodes;
    int maxDepth = 32;
    int depthSeen = 0;
    float pad = 0.f;

    static int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }

    // Returns the child reference for faces idx[start, start+count); bounds -> out.
    int32_t build(uint32_t start, uint32_t count, int depth, Box3& out)
    {
        Box3 b; b.reset();
        Box3 cb; cb.reset();
        for (uint32_t i = start; i < start + count; ++i) {
            b.grow(fb[idx[i]]);
            cb.grow(&fc[3 * (size_t)idx[i]]);
        }
        out = b;
        const int remaining = maxDepth - depth; // inner levels still available below this one, this one included: remaining + 0
        bool forceMedian = ceil_log2(count) >= remaining; // keep enough levels for a balanced finish
        if (count == 1 || (count <= kLeafMax && (remaining <= 0 || forceMedian))) return leaf(start, count);

        int bestAxis = -1, bestBin = -1;
        float bestCost = std::numeric_limits<float>::max();
        if (!forceMedian) {
            for (int axis = 0; axis < 3; ++axis) {
                float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0.f)) continue;
                Box3 bins[kBins]; uint32_t cnt[kBins];
                for (int k = 0; k < kBins; ++k) { bins[k].reset(); cnt[k] = 0; }
                float scale = (float)kBins / ext;
                for (uint32_t i = start; i < start + count; ++i) {
                    uint32_t f = idx[i];
                    int k = std::min(kBins - 1, std::max(0, (int)((fc[3 * (size_t)f + axis] - lo) * scale)));
                    bins[k].grow(fb[f]); cnt[k]++;
                }
                float rightArea[kBins]; uint32_t rightCnt[kBins];
                Box3 acc; acc.reset(); uint32_t c = 0;
                for (int k = kBins - 1; k > 0; --k) { acc.grow(bins[k]); c += cnt[k]; rightArea[k] = acc.area(); rightCnt[k] = c; }
                acc.reset(); c = 0;
                for (int k = 1; k < kBins; ++k) {
                    acc.grow(bins[k - 1]); c += cnt[k - 1];
                    if (c == 0 || rightCnt[k] == 0) continue;
                    float cost = acc.area() * (float)c + rightArea[k] * (float)rightCnt[k];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = k; }
                }
            }
            if (bestAxis >= 0) {
                float parentArea = std::max(b.area(), 1e-30f);
                float splitCost = kCostNode + kCostTri * bestCost / parentArea;
                if (count <= kLeafMax && kCostTri * (float)count <= splitCost) return leaf(start, count);
            } else if (count <= kLeafMax) {
                return leaf(start, count);
            }
        }
        uint32_t mid;
        if (bestAxis >= 0) {
            float lo = cb.lo[bestAxis], scale = (float)kBins / (cb.hi[bestAxis] - cb.lo[bestAxis]);
            auto it = std::partition(idx.begin() + start, idx.begin() + start + count, [&](uint32_t f) {
                int k = std::min(kBins - 1, std::max(0, (int)((fc[3 * (size_t)f + bestAxis] - lo) * scale)));
                return k < bestBin;
            });
            mid = (uint32_t)(it - idx.begin());
        } else {
            mid = start; // fall through to the median split
        }
        if (mid == start || mid == start + count) {
            int axis = 0;
            float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
            if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez >= ey) axis = 2;
            mid = start + count / 2;
            std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + start + count,
                             [&](uint32_t a, uint32_t c2) { return fc[3 * (size_t)a + axis] < fc[3 * (size_t)c2 + axis]; });
        }
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        depthSeen = std::max(depthSeen, depth + 1);
        Box3 lb, rb;
        int32_t l = build(start, mid - start, depth + 1, lb);
        int32_t r = build(mid, start + count - mid, depth + 1, rb);
        DNode& n = nodes[(size_t)me];
        std::memset(&n, 0, sizeof n);
        pad_box(lb); pad_box(rb);
        n.lmin[0] = lb.lo[0]; n.lmin[1] = lb.lo[1]; n.lmin[2] = lb.lo[2]; n.lmax0 = lb.hi[0];
        n.lmax12[0] = lb.hi[1]; n.lmax12[1] = lb.hi[2]; n.rmin01[0] = rb.lo[0]; n.rmin01[1] = rb.lo[1];
        n.rmin2 = rb.lo[2]; n.rmax[0] = rb.hi[0]; n.rmax[1] = rb.hi[1]; n.rmax[2] = rb.hi[2];
        n.left = l; n.right = r;
        return me;
    }
    void pad_box(Box3& b) const
    {
        for (int k = 0; k < 3; ++k) { b.lo[k] -= pad; b.hi[k] += pad; }
    }
    int32_t leaf(uint32_t start, uint32_t count) const { return ~(int32_t)((start << 3) | (count - 1)); }
};

} // namespace

BvhResult build_bvh(const float* P, uint32_t n, int max_depth)
{
    BvhResult res;
    if (const char* e = std::getenv("PTAMD_LEAF_MAX")) kLeafMax = (uint32_t)std::max(1, std::min(8, std::atoi(e)));
    if (const char* e = std::getenv("PTAMD_COST_TRI")) kCostTri = (float)std::atof(e);
    Builder b;
    b.maxDepth = max_depth;
    b.fb.resize(n); b.fc.resize(3 * (size_t)n); b.idx.resize(n);
    Box3 all; all.reset();
    for (uint32_t i = 0; i < n; ++i) {
        const float* p = P + 9 * (size_t)i;
        Box3 t; t.reset();
        t.grow(p); t.grow(p + 3); t.grow(p + 6);
        b.fb[i] = t; all.grow(t);
        for (int k = 0; k < 3; ++k) b.fc[3 * (size_t)i + k] = (p[k] + p[3 + k] + p[6 + k]) * (1.0f / 3.0f);
        b.idx[i] = i;
    }
    float scale = 0.f;
    for (int k = 0; k < 3; ++k) scale = std::max(scale, std::max(std::fabs(all.lo[k]), std::fabs(all.hi[k])));
    // Child boxes are padded so that the fused-multiply-add slab test of the kernels never culls a
    // triangle an exact (b - o) / d test would keep (its rounding error is ~1e-7 of the coordinates).
    b.pad = 2e-6f * std::max(scale, 1e-20f);
    b.nodes.reserve(n);
    Box3 rootBounds;
    res.root = n ? b.build(0, n, 0, rootBounds) : ~0;
    res.nodes = std::move(b.nodes);
    res.order = std::move(b.idx);
    res.depth = b.depthSeen;
    return res;
}

} // namespace ptamd
