// CPU model of the render kernel's wave-level traversal cost on an LDS-resident mesh scene, used to decide whether
// regrouping the rays of a workgroup between bounces (so that the 64 rays of a wave take similar walks) is worth a
// kernel change. It builds the product's own tree (bvh_build.cpp), traces diffuse paths with per-lane path
// regeneration exactly as the kernel schedules them (one ray per lane and round, 8x8 pixel tiles, 4 tiles per
// workgroup), records every ray's walk as segments (inner-node steps, then primitives of one leaf), and prices a
// wave as the while-while loop does: per segment the maximum over its lanes.
//
//   g++ -O2 -std=c++17 tools/coherence_sim.cpp path-tracer_amd/csrc/bvh_build.cpp path-tracer_amd/csrc/gltf_loader.cpp path-tracer_amd/csrc/jpeg_decode.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -lz -o /tmp/coherence_sim
//   /tmp/coherence_sim tests/golden/cornell_mesh.glb [blocks] [spp] [block_lanes]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../path-tracer_amd/csrc/bvh_build.h"
#include "../path-tracer_amd/csrc/gltf_loader.h"

using namespace ptamd;

// the loader's only dependency on the C ABI (PBRMaterial{} defaults, ModelLoader.h:60-84)
extern "C" void ptamd_pbr_default(ptamd_pbr* p)
{
    p->base_color_tex = p->normal_tex = p->metallic_roughness_tex = p->emissive_tex = p->transmission_tex = -1;
    for (int k = 0; k < 4; ++k) p->base_color_factor[k] = 1.0f;
    p->metallic_factor = 0.0f; p->roughness_factor = 1.0f; p->alpha_mode = 0; p->alpha_cutoff = 0.5f; p->normal_scale = 1.0f;
    p->emissive_factor[0] = p->emissive_factor[1] = p->emissive_factor[2] = 0.0f;
    p->transmission_factor = 0.0f; p->ior = 1.5f;
}

struct V3 { float x, y, z; };
static V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static V3 norm(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct Seg { uint16_t nodes, prims; };
struct Walk { std::vector<Seg> segs; int nodes = 0, prims = 0; float t = 1e30f; int tri = -1; };

struct Scene {
    BvhResult bvh;
    std::vector<V3> v; // 3 per slot, leaf order
    Walk trace(V3 o, V3 d) const
    {
        Walk w;
        V3 inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
        int stack[64]; int sp = 0;
        int cur = bvh.root;
        const int kDone = 0x7fffffff;
        float closest = 1e30f;
        while (cur != kDone) {
            Seg s = { 0, 0 };
            while (cur >= 0 && cur != kDone) {
                ++s.nodes;
                const DNode& n = bvh.nodes[(size_t)cur];
                const float lmn[3] = { n.lmin[0], n.lmin[1], n.lmin[2] }, lmx[3] = { n.lmax0, n.lmax12[0], n.lmax12[1] };
                const float rmn[3] = { n.rmin01[0], n.rmin01[1], n.rmin2 }, rmx[3] = { n.rmax[0], n.rmax[1], n.rmax[2] };
                const float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
                float l0 = 0.f, l1 = closest, r0 = 0.f, r1 = closest;
                for (int k = 0; k < 3; ++k) {
                    float a = (lmn[k] - oo[k]) * ii[k], b = (lmx[k] - oo[k]) * ii[k];
                    l0 = std::max(l0, std::min(a, b)); l1 = std::min(l1, std::max(a, b));
                    a = (rmn[k] - oo[k]) * ii[k]; b = (rmx[k] - oo[k]) * ii[k];
                    r0 = std::max(r0, std::min(a, b)); r1 = std::min(r1, std::max(a, b));
                }
                const bool hl = l1 >= l0, hr = r1 >= r0;
                if (hl && hr) {
                    const bool ln = l0 < r0;
                    stack[sp++] = ln ? n.right : n.left;
                    cur = ln ? n.left : n.right;
                } else if (hl || hr) cur = hl ? n.left : n.right;
                else cur = sp ? stack[--sp] : kDone;
            }
            if (cur < 0) {
                uint32_t code = ~(uint32_t)cur;
                uint32_t slot = code >> 6, mask = (code >> 2) & 15u, prims = (code & 3u) + 1u;
                for (uint32_t i = 0; i < prims; ++i, mask >>= 1) {
                    ++s.prims;
                    const int ntri = (mask & 1u) ? 2 : 1;
                    for (int k = 0; k < ntri; ++k) {
                        const V3 a = v[3 * (size_t)(slot + k)], b = v[3 * (size_t)(slot + k) + 1], c = v[3 * (size_t)(slot + k) + 2];
                        const V3 e1 = b - a, e2 = c - a, p = cross(d, e2);
                        const float det = dot(e1, p);
                        if (std::fabs(det) < 1e-8f) continue;
                        const float id = 1.0f / det;
                        const V3 tv = o - a;
                        const float u = dot(tv, p) * id;
                        if (u < 0.f || u > 1.f) continue;
                        const V3 q = cross(tv, e1);
                        const float vv = dot(d, q) * id;
                        if (vv < 0.f || u + vv > 1.f) continue;
                        const float t = dot(e2, q) * id;
                        if (t > 1e-4f && t < closest) { closest = t; w.tri = (int)(slot + k); }
                    }
                    slot += ntri;
                }
                cur = sp ? stack[--sp] : kDone;
            }
            w.segs.push_back(s);
            w.nodes += s.nodes; w.prims += s.prims;
        }
        w.t = closest;
        return w;
    }
};

struct Lane {
    V3 o, d; int bounce; int originTri; uint32_t samplesLeft;
};

// Instruction prices of one wave-level loop trip (vector instructions, measured on the kernel)
static const double kNodeTrip = 38.0, kPrimTrip = 42.0; // loop bookkeeping included

static double wave_cost(const std::vector<const Walk*>& lanes, double* nodeTrips = nullptr, double* primTrips = nullptr)
{
    size_t maxSeg = 0;
    for (auto* w : lanes) maxSeg = std::max(maxSeg, w->segs.size());
    double nt = 0, pt = 0;
    for (size_t k = 0; k < maxSeg; ++k) {
        int mn = 0, mp = 0;
        for (auto* w : lanes) if (k < w->segs.size()) { mn = std::max(mn, (int)w->segs[k].nodes); mp = std::max(mp, (int)w->segs[k].prims); }
        nt += mn; pt += mp;
    }
    if (nodeTrips) *nodeTrips += nt;
    if (primTrips) *primTrips += pt;
    return nt * kNodeTrip + pt * kPrimTrip;
}

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/cornell_mesh.glb";
    const int blocks = argc > 2 ? atoi(argv[2]) : 100;
    const int spp = argc > 3 ? atoi(argv[3]) : 8;
    const int BL = argc > 4 ? atoi(argv[4]) : 256; // lanes per workgroup
    const int depth = 8, W = 1920, H = 1080;
    Model m; load_gltf(path, m);
    Scene sc;
    sc.bvh = build_bvh(m.positions.data(), m.group.data(), m.num_faces(), 32);
    for (uint32_t s = 0; s < m.num_faces(); ++s) {
        const uint32_t f = sc.bvh.order[s];
        for (int k = 0; k < 3; ++k) sc.v.push_back({ m.positions[9 * (size_t)f + 3 * k], m.positions[9 * (size_t)f + 3 * k + 1], m.positions[9 * (size_t)f + 3 * k + 2] });
    }
    printf("faces %u nodes %zu depth %d quads %u\n", m.num_faces(), sc.bvh.nodes.size(), sc.bvh.depth, sc.bvh.num_quads);
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const V3 camPos = { 2.78f, 2.73f, -8.0f };
    const float tanHalf = std::tan(45.0f * 0.5f * 3.14159265f / 180.0f), aspect = (float)W / H;
    auto primary = [&](int px, int py) {
        float nx = ((px + 0.5f) / W) * 2 - 1, ny = ((py + 0.5f) / H) * 2 - 1;
        // camera rotated 180 degrees about y: looks down +z, x flipped
        V3 dc = norm(V3{ nx * aspect * tanHalf, ny * tanHalf, -1.0f });
        return V3{ -dc.x, dc.y, -dc.z };
    };
    const int wavesPerBlock = BL / 64;
    const int tilesX = W / 8, tilesY = H / 8;
    enum { ID, OCT, TRI_OCT, CELL_OCT, NODES, NKEYS };
    const char* names[NKEYS] = { "identity", "octant", "origin tri + octant", "origin cell(4^3) + octant", "by node count (bound)" };
    double cost[NKEYS] = { 0 }, ntr[NKEYS] = { 0 }, ptr_[NKEYS] = { 0 };
    double laneNodes = 0, lanePrims = 0, rays = 0, rounds = 0;
    // several rays per lane and traversal phase (k rounds of the same wave priced together): static = lane l walks its k
    // rays one after the other; dynamic = a lane that runs out takes the next unstarted ray of the wave's 64 k
    const int KS[4] = { 1, 2, 4, 8 };
    double statCost[4] = { 0 }, dynCost[4] = { 0 }, statRays[4] = { 0 };
    const double kFetch = 60.0; // instructions of a "next ray" body, paid by every outer iteration in which some lane needs it
    std::vector<std::vector<std::vector<Walk>>> hist; // [round][wave][lane]
    for (int b = 0; b < blocks; ++b) {
        const int block = (int)(U(rng) * (tilesX * tilesY / wavesPerBlock));
        std::vector<Lane> lanes(BL);
        std::vector<int> px(BL), py(BL);
        for (int l = 0; l < BL; ++l) {
            const int tile = block * wavesPerBlock + l / 64, lane = l & 63;
            px[l] = (tile % tilesX) * 8 + (lane & 7); py[l] = (tile / tilesX) * 8 + (lane >> 3);
            lanes[l] = { camPos, primary(px[l], py[l]), 0, -1, (uint32_t)spp };
        }
        for (;;) {
            std::vector<int> active;
            for (int l = 0; l < BL; ++l) if (lanes[l].samplesLeft) active.push_back(l);
            if ((int)active.size() < BL) break; // stop at the tail: the comparison is about the steady state
            std::vector<Walk> walks(BL);
            for (int l = 0; l < BL; ++l) { walks[l] = sc.trace(lanes[l].o, lanes[l].d); laneNodes += walks[l].nodes; lanePrims += walks[l].prims; }
            rays += BL; rounds += 1;
            {
                std::vector<std::vector<Walk>> perWave(wavesPerBlock);
                for (int l = 0; l < BL; ++l) perWave[l / 64].push_back(walks[l]);
                hist.push_back(std::move(perWave));
            }
            for (int key = 0; key < NKEYS; ++key) {
                std::vector<int> order(BL);
                for (int l = 0; l < BL; ++l) order[l] = l;
                auto oct = [&](int l) { return (lanes[l].d.x < 0) | ((lanes[l].d.y < 0) << 1) | ((lanes[l].d.z < 0) << 2); };
                auto cell = [&](int l) {
                    if (lanes[l].bounce == 0) return 64;
                    auto q = [](float x) { return std::min(3, std::max(0, (int)(x / 5.55f * 4))); };
                    return q(lanes[l].o.x) | (q(lanes[l].o.y) << 2) | (q(lanes[l].o.z) << 4);
                };
                if (key == OCT) std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return oct(a) < oct(c); });
                if (key == TRI_OCT) std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return (lanes[a].originTri / 2) * 8 + oct(a) < (lanes[c].originTri / 2) * 8 + oct(c); });
                if (key == CELL_OCT) std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return cell(a) * 8 + oct(a) < cell(c) * 8 + oct(c); });
                if (key == NODES) std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return walks[a].nodes < walks[c].nodes; });
                for (int w = 0; w < wavesPerBlock; ++w) {
                    std::vector<const Walk*> wl;
                    for (int i = 0; i < 64; ++i) wl.push_back(&walks[order[w * 64 + i]]);
                    cost[key] += wave_cost(wl, &ntr[key], &ptr_[key]);
                }
            }
            // shade + next ray
            for (int l = 0; l < BL; ++l) {
                Lane& L = lanes[l];
                const Walk& w = walks[l];
                bool done = w.tri < 0;
                if (!done) {
                    const V3 a = sc.v[3 * (size_t)w.tri], bb = sc.v[3 * (size_t)w.tri + 1], c = sc.v[3 * (size_t)w.tri + 2];
                    V3 n = norm(cross(bb - a, c - a));
                    if (dot(n, L.d) > 0) n = n * -1.0f;
                    const V3 p = L.o + L.d * w.t;
                    if (L.bounce + 1 >= depth) done = true;
                    else {
                        const float u1 = U(rng), u2 = U(rng);
                        const float r = std::sqrt(u1), phi = 6.2831853f * u2;
                        V3 t = std::fabs(n.z) < 0.999f ? norm(V3{ -n.y, n.x, 0 }) : norm(V3{ 0, -n.z, n.y });
                        V3 bt = cross(n, t);
                        V3 wi = norm(t * (r * std::cos(phi)) + bt * (r * std::sin(phi)) + n * std::sqrt(std::max(0.f, 1 - u1)));
                        L.o = p + n * 1e-4f; L.d = wi; L.bounce++; L.originTri = w.tri;
                    }
                }
                if (done) { L.samplesLeft--; L.o = camPos; L.d = primary(px[l], py[l]); L.bounce = 0; L.originTri = -1; }
            }
        }
    }
    for (int ki = 0; ki < 4; ++ki) {
        const int k = KS[ki];
        for (size_t r0 = 0; r0 + k <= hist.size(); r0 += k) {
            bool sameBlock = true; // rounds of one block are consecutive in hist; a block change shows as a shorter tail: ignore
            for (int w = 0; w < wavesPerBlock && sameBlock; ++w) {
                // static
                {
                    size_t it = 0; double c = 0;
                    for (;; ++it) {
                        int mn = 0, mp = 0; bool any = false, fetch = false;
                        for (int l = 0; l < 64; ++l) {
                            // lane's concatenated segment list
                            size_t pos = it; int j = 0;
                            while (j < k && pos >= hist[r0 + j][w][l].segs.size()) { pos -= hist[r0 + j][w][l].segs.size(); ++j; }
                            if (j >= k) continue;
                            any = true;
                            if (pos == 0 && j > 0) fetch = true;
                            mn = std::max(mn, (int)hist[r0 + j][w][l].segs[pos].nodes); mp = std::max(mp, (int)hist[r0 + j][w][l].segs[pos].prims);
                        }
                        if (!any) break;
                        c += mn * kNodeTrip + mp * kPrimTrip + (fetch ? kFetch : 0.0);
                    }
                    statCost[ki] += c; statRays[ki] += 64.0 * k;
                }
                // dynamic
                {
                    std::vector<const Walk*> queue;
                    for (int j = 0; j < k; ++j) for (int l = 0; l < 64; ++l) queue.push_back(&hist[r0 + j][w][l]);
                    size_t next = 0; const Walk* cur[64]; size_t pos[64];
                    for (int l = 0; l < 64; ++l) { cur[l] = queue[next++]; pos[l] = 0; }
                    double c = 0;
                    for (;;) {
                        int mn = 0, mp = 0; bool any = false, fetch = false;
                        for (int l = 0; l < 64; ++l) {
                            if (cur[l] && pos[l] >= cur[l]->segs.size()) { cur[l] = next < queue.size() ? queue[next++] : nullptr; pos[l] = 0; if (cur[l]) fetch = true; }
                            if (!cur[l]) continue;
                            any = true;
                            mn = std::max(mn, (int)cur[l]->segs[pos[l]].nodes); mp = std::max(mp, (int)cur[l]->segs[pos[l]].prims);
                            ++pos[l];
                        }
                        if (!any) break;
                        c += mn * kNodeTrip + mp * kPrimTrip + (fetch ? kFetch : 0.0);
                    }
                    dynCost[ki] += c;
                }
            }
        }
        printf("k = %d rays per lane and phase: static %.0f instr/wave-ray, dynamic %.0f\n", k, statCost[ki] / statRays[ki] * 64.0, dynCost[ki] / statRays[ki] * 64.0);
    }
    // Other schedules of the same walks, simulated lane by lane: a wave owns a queue of 64 k rays; a lane that has
    // finished its ray takes the next one when a refill happens (cost kFetch, when at least `refillAt` lanes are idle
    // or nothing else can run). "majority": a trip is a node step for all lanes at inner nodes or a primitive test for
    // all lanes at leaves, whichever group is larger. "both": every trip issues both bodies.
    {
        const int ks[4] = { 1, 2, 4, 8 };
        for (int mode = 0; mode < 2; ++mode)
            for (int ki = 0; ki < 4; ++ki)
                for (int refillAt : { 16, 32 }) {
                    const int k = ks[ki];
                    double cost = 0, nrays = 0;
                    for (size_t r0 = 0; r0 + k <= hist.size() && r0 < 4000; r0 += k)
                        for (int w = 0; w < wavesPerBlock; ++w) {
                            std::vector<const Walk*> queue;
                            for (int j = 0; j < k; ++j) for (int l = 0; l < 64; ++l) queue.push_back(&hist[r0 + j][w][l]);
                            size_t next = 0;
                            const Walk* cur[64]; size_t seg[64]; int nLeft[64], pLeft[64];
                            const Walk* const kWaiting = (const Walk*)1; // finished, waits for a refill
                            auto load = [&](int l) {
                                cur[l] = next < queue.size() ? queue[next++] : nullptr; seg[l] = 0;
                                if (cur[l]) { nLeft[l] = cur[l]->segs[0].nodes; pLeft[l] = cur[l]->segs[0].prims; }
                            };
                            auto advance = [&](int l) { // after a step: move to the next segment / finish
                                while (cur[l] && nLeft[l] == 0 && pLeft[l] == 0) {
                                    if (++seg[l] >= cur[l]->segs.size()) { cur[l] = kWaiting; break; }
                                    nLeft[l] = cur[l]->segs[seg[l]].nodes; pLeft[l] = cur[l]->segs[seg[l]].prims;
                                }
                            };
                            for (int l = 0; l < 64; ++l) { load(l); advance(l); }
                            for (;;) {
                                int idle = 0, atNode = 0, atLeaf = 0;
                                for (int l = 0; l < 64; ++l) {
                                    if (cur[l] == kWaiting || !cur[l]) ++idle;
                                    else if (nLeft[l] > 0) ++atNode; else ++atLeaf;
                                }
                                const bool canRefill = next < queue.size() && idle > 0;
                                if (atNode + atLeaf == 0 && !canRefill) break;
                                if (canRefill && (idle >= refillAt || atNode + atLeaf == 0)) {
                                    for (int l = 0; l < 64; ++l) if (cur[l] == kWaiting || !cur[l]) { load(l); if (cur[l]) advance(l); }
                                    cost += kFetch;
                                    continue;
                                }
                                if (mode == 0) {
                                    if (atNode >= atLeaf) { cost += kNodeTrip; for (int l = 0; l < 64; ++l) if (cur[l] > kWaiting && nLeft[l] > 0) { --nLeft[l]; advance(l); } }
                                    else { cost += kPrimTrip; for (int l = 0; l < 64; ++l) if (cur[l] > kWaiting && nLeft[l] == 0 && pLeft[l] > 0) { --pLeft[l]; advance(l); } }
                                } else {
                                    cost += (atNode ? kNodeTrip : 0) + (atLeaf ? kPrimTrip : 0);
                                    for (int l = 0; l < 64; ++l) if (cur[l] > kWaiting) { if (nLeft[l] > 0) --nLeft[l]; else --pLeft[l]; advance(l); }
                                }
                            }
                            nrays += 64.0 * k;
                        }
                    printf("%-8s k = %2d refill at %2d idle: %.0f instr/wave-ray\n", mode == 0 ? "majority" : "both", k, refillAt, cost / nrays * 64.0);
                }
    }
    // Budgeted rounds: a search ends after K node trips of the wave; lanes whose walk is not finished keep their traversal
    // state and go on in the wave's next search, beside the new rays of the lanes that finished (those are shaded and get
    // their next ray at the end of the round). Priced in cycles per wave (node trip, primitive trip, the rest of a bounce-loop
    // trip, and what parking / resuming costs when some lane is unfinished); K = 0 means no budget (today's kernel).
    {
        const double cNode = 115.0, cPrim = 130.0, cRest = 1600.0, cPark = 60.0;
        for (int perLane = 0; perLane < 3; ++perLane)
        for (int K : { 0, 32, 24, 20, 16, 14, 12, 10, 8, 6, 5, 4, 3, 2, 1 }) {
            if (perLane < 2 && K < 6 && K != 0) continue;
            if (perLane == 2 && K > 6) continue;
            const bool phases = perLane == 2; // the budget counts node phases (outer iterations), not node trips
            double cycles = 0, done = 0, nodeTrips = 0, rounds2 = 0, shadeLanes = 0;
            for (int w = 0; w < wavesPerBlock; ++w) {
                std::vector<const Walk*> queue;
                for (size_t r = 0; r < hist.size() && r < 6000; ++r) for (int l = 0; l < 64; ++l) queue.push_back(&hist[r][w][l]);
                size_t next = 0;
                const Walk* cur[64]; size_t seg[64]; int nLeft[64], pLeft[64];
                auto load = [&](int l) { cur[l] = queue[next++]; seg[l] = 0; nLeft[l] = cur[l]->segs.empty() ? 0 : cur[l]->segs[0].nodes; pLeft[l] = cur[l]->segs.empty() ? 0 : cur[l]->segs[0].prims; };
                auto finished = [&](int l) { return seg[l] >= cur[l]->segs.size(); };
                for (int l = 0; l < 64; ++l) load(l);
                while (next + 64 <= queue.size()) {
                    int trips = 0, nph = 0;
                    int used[64] = { 0 };
                    for (;;) {
                        ++nph;
                        bool anyNode = false;
                        auto can = [&](int l) { return !finished(l) && nLeft[l] > 0 && (perLane != 1 || K == 0 || used[l] < K); };
                        for (int l = 0; l < 64; ++l) if (can(l)) anyNode = true;
                        while (anyNode && (perLane || K == 0 || trips < K)) {
                            ++trips; cycles += cNode; anyNode = false;
                            for (int l = 0; l < 64; ++l) if (can(l)) { --nLeft[l]; ++used[l]; if (can(l)) anyNode = true; }
                        }
                        int mp = 0;
                        for (int l = 0; l < 64; ++l) if (!finished(l) && nLeft[l] == 0) mp = std::max(mp, pLeft[l]);
                        cycles += mp * cPrim;
                        bool more = false;
                        for (int l = 0; l < 64; ++l) if (!finished(l) && nLeft[l] == 0) {
                            pLeft[l] = 0;
                            if (++seg[l] < cur[l]->segs.size()) { nLeft[l] = cur[l]->segs[seg[l]].nodes; pLeft[l] = cur[l]->segs[seg[l]].prims; }
                        }
                        for (int l = 0; l < 64; ++l) if (!finished(l) && (perLane != 1 || K == 0 || used[l] < K)) more = true;
                        if (!more || (!perLane && K != 0 && trips >= K) || (phases && K != 0 && nph >= K)) break;
                    }
                    nodeTrips += trips; rounds2 += 1;
                    bool parked = false;
                    for (int l = 0; l < 64; ++l) { if (finished(l)) { done += 1; shadeLanes += 1; load(l); } else parked = true; }
                    cycles += cRest + (parked ? cPark : 0.0);
                }
            }
            printf("%s budget %2d node trips per search: %.0f cycles per finished ray x 64 (node trips per round %.1f, lanes shaded per round %.1f)  %.3f of no budget\n",
                   perLane == 2 ? "phases  " : perLane ? "per-lane" : "per-wave", K, cycles / done * 64.0, nodeTrips / rounds2, shadeLanes / rounds2, 0.0);
        }
    }
    printf("rays %.0f, per lane: %.2f nodes, %.2f prims per ray\n", rays, laneNodes / rays, lanePrims / rays);
    const double wavesTotal = rounds * wavesPerBlock;
    for (int key = 0; key < NKEYS; ++key)
        printf("%-28s node trips %.2f  prim trips %.2f  instr/wave-ray %.0f  (%.3f of identity)\n", names[key], ntr[key] / wavesTotal, ptr_[key] / wavesTotal,
               cost[key] / wavesTotal, cost[key] / cost[ID]);
    return 0;
}
