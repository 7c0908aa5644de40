#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <mutex>
#include <system_error>
#include <thread>
#include <unordered_map>

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

// Split search: exact sweeps for small ranges, binned above (against 16 bins everywhere the sweeps cut the wave's
// node trips on a 54-primitive room by 6 % and its render time by 4.6 %).
constexpr int kBins = 32;
constexpr uint32_t kSweepMax = 128;
// Leaves of at most two primitives (a primitive is a triangle or a quad): measured best once a node step costs
// less than a primitive test (4 % over leaves of four on a 108-triangle room, even on a 1 M-triangle mesh).
uint32_t kLeafMax = 2;
float kCostNode = 1.0f, kCostTri = 1.2f;

// What a subtree build writes: its own nodes and leaf slots (indices local to it until they are merged).
struct TreeOut {
    std::vector<uint32_t> slots;    // leaf order -> face
    std::vector<DNode> nodes;
    int depthSeen = 0;
    std::vector<uint32_t> order; std::vector<float> rightArea; // scratch of the exact sweeps
};

// Subtrees of at least this many primitives are handed to worker threads when the mesh is large: the top of the tree is
// built by the calling thread, which leaves a placeholder where such a subtree hangs.
constexpr uint32_t kParallelMin = 1u << 16;   // meshes below this size are built by one thread
constexpr uint32_t kTaskPrims = 1u << 13;     // subtree size at which a task is cut off

struct Builder {
    std::vector<Box3> fb;
    std::vector<float> fc; // 3 per face
    std::vector<uint32_t> idx;      // primitive indices, permuted by the build (a subtree only touches its own range)
    std::vector<uint32_t> primA, primB; // faces of each primitive (primB = kNone for a lone triangle)
    int maxDepth = 32;
    float pad = 0.f;
    struct Task { uint32_t start, count; int depth; int32_t parent; bool right; TreeOut out; int32_t ref; };
    std::vector<Task> tasks;        // deferred subtrees (filled only while `defer` is set)
    bool defer = false;

    static int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }

    // Returns the child reference for faces idx[start, start+count); bounds -> out. Nodes and leaf slots go to T.
    // parent / right: where the reference will be stored (used when the subtree is deferred to a task).
    int32_t build(TreeOut& T, uint32_t start, uint32_t count, int depth, Box3& out, int32_t parent = -1, bool right = false)
    {
        std::vector<uint32_t>& slots = T.slots; std::vector<DNode>& nodes = T.nodes; int& depthSeen = T.depthSeen;
        (void)slots;
        Box3 b; b.reset();
        Box3 cb; cb.reset();
        for (uint32_t i = start; i < start + count; ++i) {
            b.grow(fb[idx[i]]);
            cb.grow(&fc[3 * (size_t)idx[i]]);
        }
        out = b;
        const int remaining = maxDepth - depth; // inner levels still available below this one, this one included: remaining + 0
        bool forceMedian = ceil_log2(count) >= remaining; // keep enough levels for a balanced finish
        if (count == 1 || (count <= kLeafMax && (remaining <= 0 || forceMedian))) return leaf(T, start, count);
        if (defer && count <= kTaskPrims && parent >= 0) { // a worker thread builds this subtree; its box is known already
            tasks.push_back(Task{ start, count, depth, parent, right, TreeOut(), 0 });
            return 0x7ffffffe; // patched when the tasks are merged
        }

        // Split search. Ranges of up to kSweepMax primitives: exact sweep over the centroid order of each axis (every
        // split position is a candidate); larger ranges: kBins centroid bins per axis.
        int bestAxis = -1, bestBin = -1;
        uint32_t bestPos = 0; // sweep: primitives on the left side
        float bestCost = std::numeric_limits<float>::max();
        const bool sweep = count <= kSweepMax;
        if (!forceMedian && sweep) {
            std::vector<uint32_t>& order = T.order; std::vector<float>& rightArea = T.rightArea;
            if (order.size() < count) { order.resize(kSweepMax); rightArea.resize(kSweepMax); }
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cb.hi[axis] - cb.lo[axis] > 0.f)) continue;
                for (uint32_t i = 0; i < count; ++i) order[i] = idx[start + i];
                std::sort(order.begin(), order.begin() + count, [&](uint32_t a, uint32_t c2) {
                    const float ca = fc[3 * (size_t)a + axis], cc = fc[3 * (size_t)c2 + axis];
                    return ca < cc || (ca == cc && a < c2);
                });
                Box3 acc; acc.reset();
                for (uint32_t i = count - 1; i > 0; --i) { acc.grow(fb[order[i]]); rightArea[i] = acc.area(); }
                acc.reset();
                for (uint32_t i = 1; i < count; ++i) {
                    acc.grow(fb[order[i - 1]]);
                    const float cost = acc.area() * (float)i + rightArea[i] * (float)(count - i);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestPos = i; }
                }
            }
        } else if (!forceMedian) {
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
        }
        if (!forceMedian) {
            if (bestAxis >= 0) {
                float parentArea = std::max(b.area(), 1e-30f);
                float splitCost = kCostNode + kCostTri * bestCost / parentArea;
                if (count <= kLeafMax && kCostTri * (float)count <= splitCost) return leaf(T, start, count);
            } else if (count <= kLeafMax) {
                return leaf(T, start, count);
            }
        }
        uint32_t mid;
        if (bestAxis >= 0 && sweep) {
            std::sort(idx.begin() + start, idx.begin() + start + count, [&](uint32_t a, uint32_t c2) {
                const float ca = fc[3 * (size_t)a + bestAxis], cc = fc[3 * (size_t)c2 + bestAxis];
                return ca < cc || (ca == cc && a < c2);
            });
            mid = start + bestPos;
        } else if (bestAxis >= 0) {
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
        int32_t l = build(T, start, mid - start, depth + 1, lb, me, false);
        int32_t r = build(T, mid, start + count - mid, depth + 1, rb, me, true);
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
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    int32_t leaf(TreeOut& T, uint32_t start, uint32_t count)
    {
        std::vector<uint32_t>& slots = T.slots;
        uint32_t first = (uint32_t)slots.size(), mask = 0;
        for (uint32_t i = 0; i < count; ++i) {
            uint32_t p = idx[start + i];
            slots.push_back(primA[p]);
            if (primB[p] != kNone) { slots.push_back(primB[p]); mask |= 1u << i; }
        }
        return ~(int32_t)((first << 6) | (mask << 2) | (count - 1));
    }
};

} // namespace

namespace {
// Pairs triangles that share an edge and form a parallelogram: for A = (a, b, c) the partner must consist of b, c
// and d = b + c - a, in the same material group. Greedy, first match wins. The second triangle is then hit-tested
// through the first one's plane, so d may be off by no more than rounding: 2e-6 of the pair's own longest edge plus
// two units in the last place of its largest coordinate (a tolerance taken from the whole mesh's extent would, for a
// small feature far from the origin, approach the feature's size).
void pair_quads(const float* P, const int32_t* group, uint32_t n, std::vector<uint32_t>& primA, std::vector<uint32_t>& primB,
                std::vector<int>& rotation)
{
    // Edge table: one record per (face, edge) holding a 64-bit hash of the edge's two end points (unordered) and
    // face * 4 + edge, sorted by hash; the faces on an edge are the records of equal hash whose end points really match.
    struct Key { uint32_t v[6]; bool operator==(const Key& o) const { return !std::memcmp(v, o.v, sizeof v); } };
    auto key_of = [&](const float* p, const float* q) {
        Key k; uint32_t a[3], b[3];
        std::memcpy(a, p, 12); std::memcpy(b, q, 12);
        if (std::memcmp(a, b, 12) > 0) std::swap_ranges(a, a + 3, b);
        std::memcpy(k.v, a, 12); std::memcpy(k.v + 3, b, 12);
        return k;
    };
    auto hash_of = [](const Key& k) { uint64_t h = 1469598103934665603ull; for (uint32_t x : k.v) h = (h ^ x) * 1099511628211ull; return h ^ (h >> 29); };
    auto edge_key = [&](uint32_t f, int e) { return key_of(P + 9 * (size_t)f + 3 * e, P + 9 * (size_t)f + 3 * ((e + 1) % 3)); };
    struct Rec { uint64_t h; uint32_t id; };
    std::vector<Rec> recs((size_t)n * 3);
    for (uint32_t f = 0; f < n; ++f)
        for (int e = 0; e < 3; ++e) recs[(size_t)f * 3 + e] = Rec{ hash_of(edge_key(f, e)), f * 4u + (uint32_t)e };
    std::sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.h < y.h || (x.h == y.h && x.id < y.id); });
    std::vector<uint32_t> where((size_t)n * 3); // (face, edge) -> its place in the sorted table: the other faces on the edge sit next to it
    for (size_t i = 0; i < recs.size(); ++i) where[(size_t)(recs[i].id >> 2) * 3 + (recs[i].id & 3u)] = (uint32_t)i;
    std::vector<char> used(n, 0);
    rotation.assign(n, 0);
    for (uint32_t f = 0; f < n; ++f) {
        if (used[f]) continue;
        used[f] = 1;
        uint32_t partner = Builder::kNone; int rot = 0;
        for (int e = 0; e < 3 && partner == Builder::kNone; ++e) {
            // edge (b, c) = vertices (e, e+1); a = vertex e+2
            const float* b = P + 9 * (size_t)f + 3 * e; const float* c = P + 9 * (size_t)f + 3 * ((e + 1) % 3); const float* a = P + 9 * (size_t)f + 3 * ((e + 2) % 3);
            const Key bc = key_of(b, c);
            const uint64_t h = hash_of(bc);
            size_t first = where[(size_t)f * 3 + e];
            while (first > 0 && recs[first - 1].h == h) --first;
            for (auto it = recs.begin() + (long)first; it != recs.end() && it->h == h && partner == Builder::kNone; ++it) {
                const uint32_t g = it->id >> 2;
                if (g == f || used[g] || (group && group[g] != group[f]) || !(edge_key(g, (int)(it->id & 3u)) == bc)) continue;
                // the vertex of g that is neither b nor c
                const float* d = nullptr;
                for (int k = 0; k < 3; ++k) {
                    const float* v = P + 9 * (size_t)g + 3 * k;
                    if (std::memcmp(v, b, 12) && std::memcmp(v, c, 12)) d = v;
                }
                if (!d) continue;
                float edge = 0.f, coord = 0.f;
                for (int k = 0; k < 3; ++k) {
                    edge = std::max(edge, std::max(std::fabs(b[k] - a[k]), std::fabs(c[k] - a[k])));
                    coord = std::max(std::max(coord, std::fabs(a[k])), std::max(std::max(std::fabs(b[k]), std::fabs(c[k])), std::fabs(d[k])));
                }
                const float tol = 2e-6f * edge + 2.4e-7f * coord;
                bool ok = true;
                for (int k = 0; k < 3; ++k) ok = ok && std::fabs(d[k] - (b[k] + c[k] - a[k])) <= tol;
                if (!ok) continue;
                partner = g; rot = (e + 2) % 3; // A is used with `a` as its first vertex
            }
        }
        primA.push_back(f); primB.push_back(partner); rotation[f] = rot;
        if (partner != Builder::kNone) used[partner] = 1;
    }
}
} // namespace

BvhResult build_bvh(const float* P, const int32_t* group, uint32_t n, int max_depth)
{
    BvhResult res;
    if (const char* e = std::getenv("PTAMD_LEAF_MAX")) kLeafMax = (uint32_t)std::max(1, std::min(4, std::atoi(e)));
    if (const char* e = std::getenv("PTAMD_COST_TRI")) kCostTri = (float)std::atof(e);
    Builder b;
    b.maxDepth = max_depth;
    Box3 all; all.reset();
    // A triangle with a non-finite coordinate can never be hit (every comparison of Mesh::RayTriMT fails on NaN); it
    // stays in the tree as an inert primitive with an empty box, so that it cannot poison bounds, tolerances or the
    // ordering of the sorts.
    auto finite_tri = [&](uint32_t f) { for (int k = 0; k < 9; ++k) if (!std::isfinite(P[9 * (size_t)f + k])) return false; return true; };
    for (uint32_t i = 0; i < n; ++i) { if (!finite_tri(i)) continue; const float* p = P + 9 * (size_t)i; all.grow(p); all.grow(p + 3); all.grow(p + 6); }
    if (all.lo[0] > all.hi[0]) for (int k = 0; k < 3; ++k) all.lo[k] = all.hi[k] = 0.f; // no finite triangle at all
    float scale = 0.f;
    for (int k = 0; k < 3; ++k) scale = std::max(scale, std::max(std::fabs(all.lo[k]), std::fabs(all.hi[k])));
    const bool quads = !std::getenv("PTAMD_NO_QUADS");
    if (quads) pair_quads(P, group, n, b.primA, b.primB, res.rotation);
    else { for (uint32_t i = 0; i < n; ++i) { b.primA.push_back(i); b.primB.push_back(Builder::kNone); } res.rotation.assign(n, 0); }
    const uint32_t np = (uint32_t)b.primA.size();
    b.fb.resize(np); b.fc.resize(3 * (size_t)np); b.idx.resize(np);
    for (uint32_t i = 0; i < np; ++i) {
        Box3 t; t.reset();
        const float* p = P + 9 * (size_t)b.primA[i];
        if (!finite_tri(b.primA[i]) || (b.primB[i] != Builder::kNone && !finite_tri(b.primB[i]))) {
            if (b.primB[i] != Builder::kNone) ++res.num_quads;
            b.fb[i] = t; // empty
            for (int k = 0; k < 3; ++k) b.fc[3 * (size_t)i + k] = 0.f;
            b.idx[i] = i;
            continue;
        }
        t.grow(p); t.grow(p + 3); t.grow(p + 6);
        if (b.primB[i] != Builder::kNone) { const float* q = P + 9 * (size_t)b.primB[i]; t.grow(q); t.grow(q + 3); t.grow(q + 6); ++res.num_quads; }
        b.fb[i] = t;
        if (b.primB[i] == Builder::kNone) for (int k = 0; k < 3; ++k) b.fc[3 * (size_t)i + k] = (p[k] + p[3 + k] + p[6 + k]) * (1.0f / 3.0f);
        else { // centre of the parallelogram = midpoint of the shared edge
            const int r = res.rotation[b.primA[i]];
            const float* vb = p + 3 * ((r + 1) % 3); const float* vc = p + 3 * ((r + 2) % 3);
            for (int k = 0; k < 3; ++k) b.fc[3 * (size_t)i + k] = 0.5f * (vb[k] + vc[k]);
        }
        b.idx[i] = i;
    }
    // Child boxes are padded so that the fused-multiply-add slab test of the kernels never culls a
    // triangle an exact (b - o) / d test would keep (its rounding error is ~1e-7 of the coordinates).
    b.pad = 2e-6f * std::max(scale, 1e-20f);
    TreeOut top;
    top.nodes.reserve(np); top.slots.reserve(n);
    Box3 rootBounds;
    b.defer = np >= kParallelMin && !std::getenv("PTAMD_BVH_THREADS_OFF");
    res.root = np ? b.build(top, 0, np, 0, rootBounds) : ~0;
    if (!b.tasks.empty()) {
        // The deferred subtrees are independent (each permutes only its own range of idx): worker threads take them in
        // turn; then they are appended in task order, which keeps the result the same for any number of threads.
        b.defer = false;
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        std::atomic<size_t> next{ 0 };
        // An exception in a worker (out of memory) must not leave its thread through the top of its stack: the first one is
        // kept, the remaining tasks are skipped, and it is thrown again here once every thread has been joined.
        std::exception_ptr failure;
        std::mutex failureLock;
        auto work = [&]() {
            try {
                for (size_t i; (i = next.fetch_add(1)) < b.tasks.size();) {
                    Builder::Task& t = b.tasks[i];
                    Box3 box;
                    t.ref = b.build(t.out, t.start, t.count, t.depth, box);
                }
            } catch (...) {
                std::lock_guard<std::mutex> g(failureLock);
                if (!failure) failure = std::current_exception();
                next.store(b.tasks.size());
            }
        };
        try {
            for (unsigned i = 1; i < hw; ++i) pool.emplace_back(work);
        } catch (const std::system_error&) {
            // no more threads to be had (process or thread limit): the ones that started and this thread do the work
        }
        work();
        for (std::thread& th : pool) th.join();
        if (failure) std::rethrow_exception(failure);
        for (Builder::Task& t : b.tasks) {
            const int32_t nodeOff = (int32_t)top.nodes.size();
            const uint32_t slotOff = (uint32_t)top.slots.size();
            auto moved = [&](int32_t ref) { // a reference of the subtree, in the merged arrays
                if (ref >= 0) return ref + nodeOff;
                const uint32_t code = ~(uint32_t)ref;
                return ~(int32_t)((((code >> 6) + slotOff) << 6) | (code & 63u));
            };
            for (DNode& nd : t.out.nodes) { nd.left = moved(nd.left); nd.right = moved(nd.right); }
            top.nodes.insert(top.nodes.end(), t.out.nodes.begin(), t.out.nodes.end());
            top.slots.insert(top.slots.end(), t.out.slots.begin(), t.out.slots.end());
            (t.right ? top.nodes[(size_t)t.parent].right : top.nodes[(size_t)t.parent].left) = moved(t.ref);
            top.depthSeen = std::max(top.depthSeen, t.out.depthSeen);
        }
    }
    res.nodes = std::move(top.nodes);
    res.order = std::move(top.slots);
    res.depth = top.depthSeen;
    return res;
}

} // namespace ptamd

// ------------------------------------------------------------------------------------------------ eight-wide collapse
namespace ptamd {
namespace {

struct WideItem { int32_t ref; float lo[3], hi[3]; };

float item_area(const WideItem& it)
{
    const float dx = it.hi[0] - it.lo[0], dy = it.hi[1] - it.lo[1], dz = it.hi[2] - it.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
void children_of(const DNode& n, WideItem& l, WideItem& r)
{
    l.ref = n.left; r.ref = n.right;
    l.lo[0] = n.lmin[0]; l.lo[1] = n.lmin[1]; l.lo[2] = n.lmin[2]; l.hi[0] = n.lmax0; l.hi[1] = n.lmax12[0]; l.hi[2] = n.lmax12[1];
    r.lo[0] = n.rmin01[0]; r.lo[1] = n.rmin01[1]; r.lo[2] = n.rmin2; r.hi[0] = n.rmax[0]; r.hi[1] = n.rmax[1]; r.hi[2] = n.rmax[2];
}

// Which binary nodes become eight-wide nodes, chosen to minimise the summed surface area of the wide nodes (the
// expected number of node visits of a random ray), by dynamic programming over the binary tree (after Ylitie, Karras
// and Laine, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", 2017, section 3.2, with the
// leaves of the binary tree kept as they are):
//   cost[n][i], i = 1..7: least cost of turning the subtree of inner node n into at most i children of one wide node;
//   i = 1 makes n a wide node of its own: area(n) + (its two subtrees shared out over 8 slots);
//   i > 1: either the two subtrees share out i slots (split[n][i] of them to the left one) or fewer slots do as well.
// A leaf takes one slot and costs nothing here (its tests cost the same wherever it hangs).
struct WidePlan {
    std::vector<float> cost;    // [node][8], index 1..7 used
    std::vector<uint8_t> split; // [node][9], index 2..8: slots given to the left child when the children share i slots
    float c(int32_t ref, int i) const { return ref < 0 ? 0.f : cost[(size_t)ref * 8 + (size_t)i]; }
};

WidePlan plan_wide(const BvhResult& res)
{
    const size_t n = res.nodes.size();
    WidePlan pl; pl.cost.assign(n * 8, 0.f); pl.split.assign(n * 9, 1);
    // children before parents: an explicit post-order walk (the node array's order is the builder's business)
    std::vector<int32_t> order; order.reserve(n);
    {
        std::vector<int32_t> stack; stack.push_back(res.root);
        while (!stack.empty()) {
            const int32_t v = stack.back(); stack.pop_back();
            order.push_back(v);
            const DNode& nd = res.nodes[(size_t)v];
            if (nd.left >= 0) stack.push_back(nd.left);
            if (nd.right >= 0) stack.push_back(nd.right);
        }
    }
    for (size_t oi = order.size(); oi-- > 0;) {
        const int32_t v = order[oi];
        WideItem l, r; children_of(res.nodes[(size_t)v], l, r);
        WideItem self; self.ref = v;
        for (int k = 0; k < 3; ++k) { self.lo[k] = std::min(l.lo[k], r.lo[k]); self.hi[k] = std::max(l.hi[k], r.hi[k]); }
        float shared[9]; // least cost of the two subtrees sharing j slots
        for (int j = 2; j <= 8; ++j) {
            float best = std::numeric_limits<float>::max(); int bk = 1;
            for (int k = 1; k < j; ++k) {
                const float cst = pl.c(l.ref, std::min(k, 7)) + pl.c(r.ref, std::min(j - k, 7));
                if (cst < best) { best = cst; bk = k; }
            }
            shared[j] = best; pl.split[(size_t)v * 9 + (size_t)j] = (uint8_t)bk;
        }
        float* c = &pl.cost[(size_t)v * 8];
        const float area = std::max(item_area(self), 0.f);
        c[1] = area + shared[8];
        for (int i = 2; i <= 7; ++i) c[i] = std::min(shared[i], c[i - 1]);
    }
    return pl;
}

// The children (leaves and inner nodes that become wide nodes themselves) that `it` contributes to its wide parent when
// it may take up to `slots` of the parent's slots.
void gather_children(const BvhResult& res, const WidePlan& pl, const WideItem& it, int slots, WideItem* out, int& n)
{
    if (it.ref < 0 || slots <= 1) { out[n++] = it; return; }
    const float* c = &pl.cost[(size_t)it.ref * 8];
    int i = std::min(slots, 7);
    while (i > 1 && c[i] == c[i - 1]) --i; // fewer slots do as well
    if (i == 1) { out[n++] = it; return; }
    WideItem l, r; children_of(res.nodes[(size_t)it.ref], l, r);
    const int k = pl.split[(size_t)it.ref * 9 + (size_t)i];
    gather_children(res, pl, l, k, out, n);
    gather_children(res, pl, r, i - k, out, n);
}

} // namespace

void build_wide(BvhResult& res)
{
    res.wide.clear(); res.wide_order.clear(); res.wide_depth = 0;
    if (res.root < 0 || res.nodes.empty()) return;
    const bool greedy = std::getenv("PTAMD_WIDE_GREEDY") != nullptr; // the older collapse: open the largest child until eight
    WidePlan plan;
    if (!greedy) plan = plan_wide(res);
    struct Pending { int32_t binary; uint32_t wide; int level; };
    std::vector<Pending> queue;
    res.wide.emplace_back();
    queue.push_back({ res.root, 0u, 1 });
    res.wide_order.reserve(res.order.size());
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const Pending pe = queue[qi];
        res.wide_depth = std::max(res.wide_depth, pe.level);
        // Collapse: start from the two children of the binary node and keep opening the inner child with the largest
        // surface area until there are eight children or only leaves.
        WideItem items[8]; int n = 2;
        children_of(res.nodes[(size_t)pe.binary], items[0], items[1]);
        if (!greedy) {
            const WideItem l = items[0], r = items[1];
            const int k = plan.split[(size_t)pe.binary * 9 + 8];
            n = 0;
            gather_children(res, plan, l, k, items, n);
            gather_children(res, plan, r, 8 - k, items, n);
        }
        while (greedy && n < 8) {
            int best = -1; float bestArea = -1.f;
            for (int i = 0; i < n; ++i) if (items[i].ref >= 0 && item_area(items[i]) > bestArea) { bestArea = item_area(items[i]); best = i; }
            if (best < 0) break;
            WideItem l, r;
            children_of(res.nodes[(size_t)items[best].ref], l, r);
            items[best] = l; items[n++] = r;
        }
        // Node frame: origin = lower corner of all children, pitch 2^e per axis with 255 * 2^e >= extent.
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); }
        for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) {
            if (items[i].lo[k] <= items[i].hi[k]) { lo[k] = std::min(lo[k], items[i].lo[k]); hi[k] = std::max(hi[k], items[i].hi[k]); }
        }
        DWideNode w; std::memset(&w, 0, sizeof w);
        int e[3]; double pitch[3];
        for (int k = 0; k < 3; ++k) {
            if (!(lo[k] <= hi[k])) { lo[k] = hi[k] = 0.f; } // only empty boxes (inert triangles)
            w.origin[k] = lo[k];
            const double ext = (double)hi[k] - (double)lo[k];
            int ek = -100;
            if (ext > 0.0) { ek = (int)std::ceil(std::log2(ext / 255.0)); while (std::ldexp(255.0, ek) < ext) ++ek; }
            ek = std::max(-126, std::min(126, ek));
            e[k] = ek; pitch[k] = std::ldexp(1.0, ek);
        }
        w.ex = (uint8_t)(e[0] + 127); w.ey = (uint8_t)(e[1] + 127); w.ez = (uint8_t)(e[2] + 127);
        // Slot assignment: slot bits say on which side of the node centre a child should lie (bit k set = upper side of
        // axis k), so that slot ^ (octant of the ray direction) orders the children roughly front to back. Greedy: the
        // best (child, slot) pair first.
        int slotOf[8]; bool slotUsed[8] = { false }, childDone[8] = { false };
        float centre[3]; for (int k = 0; k < 3; ++k) centre[k] = 0.5f * (lo[k] + hi[k]);
        for (int round = 0; round < n; ++round) {
            int bc = -1, bs = -1; float bestScore = -std::numeric_limits<float>::max();
            for (int c = 0; c < n; ++c) {
                if (childDone[c]) continue;
                for (int s = 0; s < 8; ++s) {
                    if (slotUsed[s]) continue;
                    float score = 0.f;
                    for (int k = 0; k < 3; ++k) {
                        const float d = 0.5f * (items[c].lo[k] + items[c].hi[k]) - centre[k];
                        score += ((s >> k) & 1) ? d : -d;
                    }
                    if (score > bestScore) { bestScore = score; bc = c; bs = s; }
                }
            }
            slotOf[bc] = bs; slotUsed[bs] = true; childDone[bc] = true;
        }
        int childAt[8]; for (int s = 0; s < 8; ++s) childAt[s] = -1;
        for (int c = 0; c < n; ++c) childAt[slotOf[c]] = c;
        w.child_base = (uint32_t)res.wide.size();
        w.tri_base = (uint32_t)res.wide_order.size();
        for (int s = 0; s < 8; ++s) {
            uint8_t* q[6] = { &w.qlo_x[s], &w.qlo_y[s], &w.qlo_z[s], &w.qhi_x[s], &w.qhi_y[s], &w.qhi_z[s] };
            const int c = childAt[s];
            if (c < 0 || !(items[c].lo[0] <= items[c].hi[0])) { // empty slot, or a child that holds only inert triangles
                for (int k = 0; k < 3; ++k) { *q[k] = 255; *q[3 + k] = 0; }
                if (c < 0) continue;
            } else {
                for (int k = 0; k < 3; ++k) {
                    double a = std::floor(((double)items[c].lo[k] - (double)w.origin[k]) / pitch[k]);
                    double b = std::ceil(((double)items[c].hi[k] - (double)w.origin[k]) / pitch[k]);
                    a = std::max(0.0, std::min(255.0, a)); b = std::max(0.0, std::min(255.0, b));
                    // outward in the arithmetic of the kernels as well: origin + q * pitch evaluated in float
                    while (a > 0.0 && (float)((double)w.origin[k] + a * pitch[k]) > items[c].lo[k]) a -= 1.0;
                    while (b < 255.0 && (float)((double)w.origin[k] + b * pitch[k]) < items[c].hi[k]) b += 1.0;
                    *q[k] = (uint8_t)a; *q[3 + k] = (uint8_t)b;
                }
            }
            if (items[c].ref >= 0) {
                w.imask |= (uint8_t)(1u << s);
                queue.push_back({ items[c].ref, (uint32_t)res.wide.size(), pe.level + 1 });
                res.wide.emplace_back();
            } else {
                const uint32_t code = ~(uint32_t)items[c].ref;
                const uint32_t first = code >> 6, mask = (code >> 2) & 15u, prims = (code & 3u) + 1u;
                if (prims > 2u) { res.wide.clear(); res.wide_order.clear(); res.wide_depth = 0; return; } // the slot byte holds leaves of 1 or 2 primitives
                uint32_t slots = 0;
                for (uint32_t i = 0; i < prims; ++i) slots += ((mask >> i) & 1u) ? 2u : 1u;
                const uint32_t rel = (uint32_t)res.wide_order.size() - w.tri_base;
                // at most 8 leaf slots of at most 2 primitives (4 triangle slots) each: rel <= 28
                w.meta[s] = (uint8_t)(rel | ((prims - 1u) << 5) | ((mask & 3u) << 6));
                for (uint32_t i = 0; i < slots; ++i) res.wide_order.push_back(res.order[first + i]);
            }
        }
        res.wide[pe.wide] = w;
    }
}

} // namespace ptamd
