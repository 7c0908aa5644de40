// HIP kernels of the path-tracing core (gfx950 / CDNA4, wave64).
//
// render_kernel: one lane per pixel, one wave per 8x8-pixel tile. Each lane runs every sample of its
// pixel for this launch with *path regeneration*: the loop body is one ray segment (closest hit +
// shade); a lane whose path ends adds the sample to its film sum and immediately starts the next
// sample, so lanes stay busy although path lengths differ. The film is read once and written once
// per launch (16 B per pixel each way), in the reference's sample order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "pt_device.h"

namespace ptamd {

constexpr int kBlock = 256;      // 4 waves, each owning one 8x8 tile
constexpr int kStackDepth = 32;  // deepest BVH (inner levels) the kernels can traverse: per-lane stack entries in LDS

// Copies the scene's nodes and hot triangles into LDS (when they fit) and returns where traversal reads them.
template <bool LDS_GEOM>
PT_DEV Geom stage_geometry(const DScene& sc, float4* lds_geom)
{
    Geom g;
    if (LDS_GEOM) {
        // the host's LDS-format nodes (see kLdsNodeBytes) and the hot triangles, records contiguous: the reads of one
        // record then differ only in their immediate offset
        const uint32_t n4 = sc.lds_node_f4, t4 = sc.num_tris * 3u;
        for (uint32_t i = threadIdx.x; i < n4; i += kBlock) lds_geom[i] = sc.lds_nodes[i];
        for (uint32_t i = threadIdx.x; i < t4; i += kBlock) lds_geom[n4 + i] = sc.tris[i];
        __syncthreads();
        g.nodes = lds_geom; g.tris = lds_geom + n4;
    } else {
        g.nodes = sc.nodes; g.tris = sc.tris;
    }
    return g;
}

// Film::AddSample (Film.cpp:34-46) on a register accumulator.
PT_DEV void add_sample(f3& acc, f3 c)
{
    float lum = dot(c, mk3(0.2126f, 0.7152f, 0.0722f));
    if (lum > 12.0f) c = c * (12.0f / lum);
    acc = acc + c;
}

// Dynamic LDS layout: [rp.stack_depth][kBlock] traversal stack entries (the depth of the scene's deepest tree),
// [kLaneFloats][kBlock] per-lane floats, then either the staged nodes and triangles (LDS_GEOM) or the work-sharing
// words of the cooperative traversal.
// Occupancy: throughput grows with resident waves up to 5 per SIMD, where the vector ALU saturates; asking for 5
// holds the register allocator to 96 VGPRs, which the LDS_GEOM builds meet without spilling. 6 (80 VGPRs) and
// 8 (64) spill into the loop and lose.
// COUNT: also tally rays, BVH node visits and triangle tests (PTAMD_FLAG_COUNT); kept out of the timed build.
// EXT: the two options the reference does not have, Russian roulette and the cubemap sky; a build of its own so
// that the default kernel carries neither their registers nor their branches.
constexpr uint32_t kFlagRoulette = 2u; // PTAMD_FLAG_RUSSIAN_ROULETTE
constexpr int kNumCounters = 10;        // fields of ptamd_stats, in order
#ifndef PT_PRIMARY_CACHE
#define PT_PRIMARY_CACHE 1 // work-sharing builds keep the camera ray's hit per pixel after the first sample of a launch
#endif
#ifndef PT_WAVES
#define PT_WAVES 5
#endif
#ifndef PT_WAVES_SM
#define PT_WAVES_SM PT_WAVES
#endif
#ifndef PT_WAVES_COOP
#define PT_WAVES_COOP 4 // the cooperative traversal needs ~112 VGPRs; spilling them costs more than the fifth wave brings
#endif
constexpr int kLaneFloats = 4;          // per-lane words parked in LDS: the film sum of the lane's pixel, the upper half of its path key
template <bool LDS_GEOM, bool COUNT, bool EXT, bool TILED>
__global__ __launch_bounds__(kBlock, LDS_GEOM ? PT_WAVES : PT_WAVES_COOP) void render_kernel(DScene sc, DRenderParams rp)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    StackEntry* stack = reinterpret_cast<StackEntry*>(lds_raw);
    const uint32_t stackBytes = (uint32_t)rp.stack_depth * kBlock * (uint32_t)sizeof(StackEntry);
    float* lds_acc = reinterpret_cast<float*>(lds_raw + stackBytes); // [kLaneFloats][kBlock]
    // Either the staged geometry (small scenes, lean traversal) or the work-sharing words of the cooperative
    // traversal (8 + 4 bytes per lane, and 20 for the kept camera-ray hit; meshes read from global memory) follow the per-lane floats.
    constexpr bool COOP = !LDS_GEOM;
    Coop coop;
    coop.res = reinterpret_cast<unsigned long long*>(lds_raw + stackBytes + 4 * kLaneFloats * kBlock);
    coop.mail = reinterpret_cast<uint32_t*>(lds_raw + stackBytes + (4 * kLaneFloats + 8) * kBlock);
    float4* lds_geom = reinterpret_cast<float4*>(lds_raw + stackBytes + 4 * kLaneFloats * kBlock);
    const Geom geom = stage_geometry<LDS_GEOM>(sc, lds_geom);
    const int slot = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); // this wave's 8x8 tile, in launch order
    int px, py; bool inside;
    if (TILED) {
        // Interleaved square tiles of one rank (ptamd_render_tiles): wave-tiles are numbered through the owned tiles one
        // after the other. A build of its own: the same lines in the rect kernel's prologue moved its loops by 260
        // bytes and cost it 2 % (identical loop code, different placement).
        const int owned = tile / rp.tile_waves, sub = tile % rp.tile_waves;
        const int big = rp.tile_phase + owned * rp.tile_stride;
        const int bx = big % rp.tile_cols, by = big / rp.tile_cols;
        px = bx * rp.tile_size + (sub % rp.tile_side) * 8 + (lane & 7); py = by * rp.tile_size + (sub / rp.tile_side) * 8 + (lane >> 3);
        inside = (owned < rp.tiles_owned) && (px < rp.x1) && (py < rp.y1);
    } else {
        const int tx = tile % rp.tiles_x, ty = tile / rp.tiles_x;
        px = rp.x0 + tx * 8 + (lane & 7); py = rp.y0 + ty * 8 + (lane >> 3);
        inside = (ty < rp.tiles_y) && (px < rp.x1) && (py < rp.y1);
    }
    const int W = rp.cam.width;
    const uint32_t pixel = (uint32_t)(py * W + px);

    // depth <= 0: TraceRay returns black at once (PathTracer.cpp:35-36); the samples are still counted.
    uint32_t s = rp.spp_begin;
    const uint32_t s_end = (inside && rp.depth > 0) ? rp.spp_begin + rp.spp_count : rp.spp_begin;
    const uint32_t seed_hash = pcg_hash(rp.seed);

    // The running film sum of this lane's pixel lives in LDS, not in registers: it is touched once per sample,
    // and three more live VGPRs through the whole bounce loop would cost occupancy.
    {
        f3 a0 = mk3(0.f, 0.f, 0.f);
        if (inside) a0 = mk3(rp.accum[3 * (size_t)pixel], rp.accum[3 * (size_t)pixel + 1], rp.accum[3 * (size_t)pixel + 2]);
        lds_acc[slot] = a0.x; lds_acc[kBlock + slot] = a0.y; lds_acc[2 * kBlock + slot] = a0.z;
    }

    // Path state: throughput T, radiance L, medium index, bounce, random key. The primary ray is written as
    // recomputed for every sample (no jitter: Camera.cpp:47-48); the compiler keeps the loop-invariant direction
    // in registers, and parking it in LDS instead measured 1.5 % slower.
    Ray ray = camera_ray(rp.cam, px, py);
    float ior = 1.0f;
    f3 T = mk3(1.f, 1.f, 1.f), L = mk3(0.f, 0.f, 0.f);
    int bounce = 0;
    uint32_t key = path_key(seed_hash, pixel, s);
    uint32_t* lds_key_hi = reinterpret_cast<uint32_t*>(lds_acc) + 3 * kBlock; // read once per bounce: not worth a register through the search
    lds_key_hi[slot] = path_key_hi(seed_hash, pixel, s);
    uint32_t nrays = 0;
    TraversalCounts tc_total; tc_total.nodes = 0; tc_total.tris = 0; tc_total.node_trips = 0; tc_total.tri_trips = 0; tc_total.steal_rounds = 0;
    uint32_t ray_trips = 0, shade_lanes = 0, shade_trips = 0; // wave-level trips of the bounce loop and of its shading block

    // The camera ray of a pixel is the same for every sample (no jitter: Camera.cpp:47-48), and so is its closest hit.
    // Where idle lanes can take over other lanes' subtrees (COOP) a lane keeps that hit after its first sample and
    // joins the search as a helper while its path is at bounce 0. (Without work sharing the lane would only wait for
    // the other 63: measured, no gain.)
    // (it waits in LDS: five registers fewer through the bounce loop)
    float* primary = reinterpret_cast<float*>(lds_raw + stackBytes + (4 * kLaneFloats + 12) * kBlock); // [5][kBlock], COOP only
    bool havePrimary = false;

    while (s < s_end) {
        Best best;
        const bool cached = COOP && PT_PRIMARY_CACHE && havePrimary && bounce == 0;
        if (COUNT) {
            TraversalCounts tc;
            best = closest_hit<kBlock, true, COOP>(sc, geom, ray, stack, coop, slot, &tc, cached);
            ++nrays; // path segments answered, those from the kept camera-ray hit included
            tc_total.nodes += tc.nodes; tc_total.tris += tc.tris;
            tc_total.node_trips += tc.node_trips; tc_total.tri_trips += tc.tri_trips; tc_total.steal_rounds += tc.steal_rounds;
            if (wave_leader()) ++ray_trips;
        } else {
            best = closest_hit<kBlock, false, COOP>(sc, geom, ray, stack, coop, slot, nullptr, cached);
        }
        if (COOP && PT_PRIMARY_CACHE && bounce == 0) {
            if (cached) {
                best.t = primary[slot]; best.obj = __float_as_int(primary[kBlock + slot]); best.tri = __float_as_int(primary[2 * kBlock + slot]);
                best.u = primary[3 * kBlock + slot]; best.v = primary[4 * kBlock + slot];
            } else {
                primary[slot] = best.t; primary[kBlock + slot] = __int_as_float(best.obj); primary[2 * kBlock + slot] = __int_as_float(best.tri);
                primary[3 * kBlock + slot] = best.u; primary[4 * kBlock + slot] = best.v;
                havePrimary = true;
            }
        }
        bool done;
        if (best.obj < 0) {
            const f3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
            L = L + T * ((EXT && sc.sky_size > 0) ? sky_lookup(sc, ray.d) : bg); // PathTracer.cpp:59-60
            done = true;
        } else {
            HitInfo h;
            if (COUNT) { ++shade_lanes; if (wave_leader()) ++shade_trips; }
            fill_hit(sc, ray, best, h);
            if (rp.albedo_only && bounce == 0) {  // PathTracer.cpp:62-71 (recursive calls never pass albedoOnly)
                float dist = clampf(best.t / 20.f, 0.0f, 0.8f);
                L = h.albedo * (1.0f - dist);
                done = true;
            } else {
                L = L + T * h.emission;           // PathTracer.cpp:76
                if (bounce + 1 >= rp.depth) {
                    done = true;                  // the next TraceRay would return 0 (PathTracer.cpp:35-36)
                } else {
                    f3 w;
                    done = !sample_bsdf(h, ray, ior, w, key, lds_key_hi[slot], (uint32_t)bounce);
                    T = T * w;
                    if (EXT && (rp.flags & kFlagRoulette) && bounce + 1 >= 3 && !done) {
                        // survive with probability q = clamp(max T, 0.05, 1), drawn from slot 7 of this bounce
                        float q = clampf(fmaxf(fmaxf(T.x, T.y), T.z), 0.05f, 1.0f);
                        if (!(rand01(key, lds_key_hi[slot], (uint32_t)bounce, 7u) < q)) done = true; // read again: not held through the BSDF step
                        else T = T * __fdiv_rn(1.0f, q);
                    }
                    ++bounce;
                }
            }
        }
        if (done) {
            f3 acc = mk3(lds_acc[slot], lds_acc[kBlock + slot], lds_acc[2 * kBlock + slot]);
            add_sample(acc, L);
            lds_acc[slot] = acc.x; lds_acc[kBlock + slot] = acc.y; lds_acc[2 * kBlock + slot] = acc.z;
            ++s;
            ray = camera_ray(rp.cam, px, py); ior = 1.0f;
            T = mk3(1.f, 1.f, 1.f); L = mk3(0.f, 0.f, 0.f);
            bounce = 0;
            key = path_key(seed_hash, pixel, s);
            lds_key_hi[slot] = path_key_hi(seed_hash, pixel, s);
        }
    }

    if (inside) {
        rp.accum[3 * (size_t)pixel] = lds_acc[slot]; rp.accum[3 * (size_t)pixel + 1] = lds_acc[kBlock + slot]; rp.accum[3 * (size_t)pixel + 2] = lds_acc[2 * kBlock + slot];
        rp.samples[pixel] += rp.spp_count;
    }
    if (COUNT && rp.counters) {
        // wave-level reduction, then one atomic per wave and counter
        unsigned long long c[kNumCounters] = { inside ? rp.spp_count : 0u, nrays, tc_total.nodes, tc_total.tris,
                                               ray_trips, tc_total.node_trips, tc_total.tri_trips, shade_lanes, shade_trips, tc_total.steal_rounds };
#pragma unroll
        for (int k = 0; k < kNumCounters; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c[k] += __shfl_xor(c[k], off, 64);
            if (lane == 0) atomicAdd(&rp.counters[k], c[k]);
        }
    }
}

// ---------------------------------------------------------------- sample-major render (LDS-resident meshes, many samples per launch)
//
// The camera has no jitter (Camera.cpp:47-48): every sample of a pixel starts with the same ray, the same closest hit and
// the same material. render_samples_kernel therefore gives a wave a 4x4-pixel tile instead of 64 pixels. The wave finds the
// camera-ray hits of its 16 pixels once and keeps the rebuilt hit records in LDS (80 bytes per pixel). Its lanes then draw
// (pixel, sample) tickets from a counter of the tile, samples in ascending order, and a lane whose path ends starts its next
// ticket from the pixel's record in the same trip of the bounce loop: its next closest-hit search is already the first
// bounce, so one search in 6.6 (mesh Cornell) is never made, and no lane waits for another's longer run of samples.
// (Keeping the hit per LANE in the pixel-major kernel would need 17 KB of LDS per workgroup and cost a resident workgroup.)
//
// Film: the samples of a pixel come from many lanes, so each finished sample is added to a per-pixel 64-bit fixed-point sum
// in LDS (2^-35 steps; LDS integer atomics). Integer addition is associative: the launch's sum does not depend on which
// lane ran which sample or in what order (the random stream is keyed by pixel and sample, not by lane), and it is the
// exactly rounded sum of the clamped samples (Film::AddSample, Film.cpp:34-46) where sequential float additions round after
// every sample. It is added to the film in one float addition per launch, so splitting a render into launches changes
// the film by float rounding only (the pixel-major kernel reproduces the reference's order of additions instead and is
// bit-identical under any split). A sample that is not finite makes the pixel NaN, as it would in a float sum.
constexpr float kFxScale = 34359738368.0f; // 2^35: |sample| <= 12 / 0.0722 < 2^8 after the luminance clamp, 2^20 samples per launch
constexpr double kFxInv = 1.0 / 34359738368.0;
constexpr uint32_t kSampleMajorMaxSpp = 1u << 20;
struct SampleTile {                  // per wave, in LDS
    unsigned long long acc[16][4];   // r, g, b sums and the count of non-finite samples of each pixel
    uint32_t list[16];               // the pixels (0..15) that are path traced: inside the rectangle, camera ray hits something
    uint32_t next;                   // ticket counter of the tile
    uint32_t pad[3];
};
static_assert(4 * kLaneFloats * kBlock == kLdsFixedBytes, "the host's residency decision counts these bytes");
static_assert(sizeof(SampleTile) % 16 == 0, "the hit records behind the tiles are read 16 bytes at a time");
constexpr uint32_t kSamplesTileBytes = 4 * sizeof(SampleTile);
constexpr uint32_t kSamplesCacheBytes = 4 * 16 * 5 * 16; // [wave][pixel][5 float4]
constexpr uint32_t kSamplesKeyBytes = 4 * kBlock;        // the upper half of each lane's path key
constexpr uint32_t kSamplesParkBytes = 3 * 4 * kBlock;   // BUDGET: the parked walk of each lane (WalkBudget)
PT_DEV unsigned long long to_fixed(float c)
{
    if (c < 0.0f) return (unsigned long long)__float2ll_rn(c * kFxScale); // not a radiance this renderer produces; kept exact anyway
    // c * 2^35 < 2^43 in two conversions that each fit 32 bits: the upper 19 bits from c * 2^11, the lower 24 from its fraction
    const float hiF = c * 2048.0f;
    const uint32_t hi = (uint32_t)hiF;
    const uint32_t lo = (uint32_t)(__builtin_amdgcn_fractf(hiF) * 16777216.0f);
    return ((unsigned long long)hi << 24) + lo;
}
PT_DEV void add_sample_fx(unsigned long long* acc, f3 c, unsigned long long times = 1ull)
{
    float lum = dot(c, mk3(0.2126f, 0.7152f, 0.0722f));
    if (lum > 12.0f) c = c * (12.0f / lum);
    const float sum = fabsf(c.x) + fabsf(c.y) + fabsf(c.z);
    if (!(sum < 768.0f)) { // NaN, infinite, or a magnitude that the clamp of non-negative radiance cannot leave
        if (!(sum < 3.0e38f)) { atomicAdd(&acc[3], 1ull); return; }
        atomicAdd(&acc[0], (unsigned long long)__float2ll_rn(c.x * kFxScale) * times);
        atomicAdd(&acc[1], (unsigned long long)__float2ll_rn(c.y * kFxScale) * times);
        atomicAdd(&acc[2], (unsigned long long)__float2ll_rn(c.z * kFxScale) * times);
        return;
    }
    atomicAdd(&acc[0], to_fixed(c.x) * times);
    atomicAdd(&acc[1], to_fixed(c.y) * times);
    atomicAdd(&acc[2], to_fixed(c.z) * times);
}
// LDS_GEOM: the scene's nodes and triangles are staged into LDS (lean traversal); otherwise they are read from global memory
// by the work-sharing traversals, which every lane of the wave enters together, a lane without a ray as a helper.
// TILED: the launch covers one rank's interleaved square tiles (ptamd_render_tiles) instead of a rectangle.
// BUDGET (LDS_GEOM builds, scenes whose last object is a mesh): a wave cuts the tail of that mesh's walk, and a lane whose
// walk is cut goes on with it in the wave's next search (WalkBudget in pt_device.h). Every extra search repeats the tests of
// the objects before the mesh, so the cut pays less the longer that list is (a sphere or a light in front of the mesh
// Cornell: +7..8 %; seven boxes and two spheres around it: -20 %): the host measures per scene whether to use this build.
// BUDGET without LDS_GEOM (scenes whose only mesh is their last object): the shared walk of that mesh is deferred. A search
// covers the other objects only; a lane whose ray then meets the mesh's bounds in front of its closest hit waits, and the wave
// walks the mesh for all waiting lanes at once, with every other lane as a helper, as soon as rp.walk_budget lanes wait or no
// lane has anything else to do. Between two walks the waiting lanes sit out the shading of the others, which go on to their
// next rays. A walk with few rays costs about as many wave trips as the tree is deep whatever their number (13 node trips
// for the 3.6 visits per ray of a 7680-triangle mesh among primitives), so fewer, fuller walks win more than the thinner
// shading loses: 2397 -> 3065 Msamples/s on that scene with 20 lanes. The closest hit of a ray does not depend on when its
// walk runs: the film is the same bit for bit.
template <bool LDS_GEOM, bool COUNT, bool EXT, bool TILED, bool BUDGET = false>
__global__ __launch_bounds__(kBlock, LDS_GEOM ? PT_WAVES_SM : PT_WAVES_COOP) void render_samples_kernel(DScene sc, DRenderParams rp)
{
    constexpr bool COOP = !LDS_GEOM;
    constexpr bool DEFER = COOP && BUDGET; // see below: the shared walk of the scene's one mesh waits for enough rays
    constexpr bool CUT = LDS_GEOM && BUDGET;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    StackEntry* stack = reinterpret_cast<StackEntry*>(lds_raw);
    const uint32_t stackBytes = (uint32_t)rp.stack_depth * kBlock * (uint32_t)sizeof(StackEntry);
    const int slot = threadIdx.x, lane = slot & 63;
    const int wave = __builtin_amdgcn_readfirstlane(slot >> 6); // wave-uniform by construction: tile addresses and origins stay in scalar registers
    SampleTile& tile = reinterpret_cast<SampleTile*>(lds_raw + stackBytes)[wave];
    float4* cache = reinterpret_cast<float4*>(lds_raw + stackBytes + kSamplesTileBytes) + wave * (16 * 5);
    uint32_t* lds_key_hi = reinterpret_cast<uint32_t*>(lds_raw + stackBytes + kSamplesTileBytes + kSamplesCacheBytes);
    // BUDGET: the parked walks, then the staged geometry (LDS_GEOM) or the work-sharing words (8 + 4 bytes per lane) in its place
    float4* lds_geom = reinterpret_cast<float4*>(lds_raw + stackBytes + kSamplesTileBytes + kSamplesCacheBytes + kSamplesKeyBytes + (CUT ? kSamplesParkBytes : 0u));
    WalkBudget wb;
    wb.park = reinterpret_cast<uint32_t*>(lds_raw + stackBytes + kSamplesTileBytes + kSamplesCacheBytes + kSamplesKeyBytes);
    wb.cut64 = rp.walk_budget; wb.resume = false; wb.unfinished = false;
    Coop coop;
    coop.res = reinterpret_cast<unsigned long long*>(lds_geom);
    coop.mail = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(lds_geom) + 8 * kBlock);
    const int p0 = lane & 15; // the pixel this lane traces the camera ray of, with three others
    const bool first = lane < 16;
    if (lane == 0) tile.next = 0u;
    if (first) { tile.acc[p0][0] = 0ull; tile.acc[p0][1] = 0ull; tile.acc[p0][2] = 0ull; tile.acc[p0][3] = 0ull; }
    const Geom geom = stage_geometry<LDS_GEOM>(sc, lds_geom); // with geometry to stage it ends in a workgroup barrier
    // the workgroup covers an 8x8-pixel block, its waves the four 4x4 quadrants
    // Workgroups go to the eight XCDs in turn, and each XCD has its own L2. Of every 64 consecutive 8x8-pixel blocks of the
    // film an XCD takes eight neighbours, so that most blocks which share film cache lines share an L2 (memory-side bytes
    // 130 -> 79 MB per 1080p launch, against 66 MB of film), while every XCD still works all over the film (an eighth of the
    // film per XCD balances worse: 520 against 506 ms).
    const int tileIndex = (int)((blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u));
    int bx0, by0; // the block's first pixel
    if (TILED) {
        // one rank's interleaved square tiles: blocks are numbered through the owned tiles one after the other
        if (tileIndex >= rp.tiles_owned * rp.tile_waves) return; // the launch is rounded up to a multiple of 64 blocks
        const int owned = tileIndex / rp.tile_waves, sub = tileIndex % rp.tile_waves;
        const int big = rp.tile_phase + owned * rp.tile_stride;
        bx0 = (big % rp.tile_cols) * rp.tile_size + (sub % rp.tile_side) * 8; by0 = (big / rp.tile_cols) * rp.tile_size + (sub / rp.tile_side) * 8;
    } else {
        if (tileIndex >= rp.tiles_x * rp.tiles_y) return;
        bx0 = rp.x0 + (tileIndex % rp.tiles_x) * 8; by0 = rp.y0 + (tileIndex / rp.tiles_x) * 8;
    }
    const int tx0 = bx0 + (wave & 1) * 4, ty0 = by0 + (wave >> 1) * 4;
    const int W = rp.cam.width;
    const uint32_t seed_hash = pcg_hash(rp.seed);
    const f3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);

    uint32_t nrays = 0, ray_trips = 0, shade_lanes = 0, shade_trips = 0;
    TraversalCounts tc_total; tc_total.nodes = 0; tc_total.tris = 0; tc_total.node_trips = 0; tc_total.tri_trips = 0; tc_total.steal_rounds = 0;
    uint32_t steal_rounds = 0;
    auto search = [&](const Ray& r, bool helper) { // helper (work-sharing builds): no ray of its own, takes over others' subtrees
        if (COUNT) {
            TraversalCounts tc;
            Best b = closest_hit<kBlock, true, COOP>(sc, geom, r, stack, coop, slot, &tc, helper);
            tc_total.nodes += tc.nodes; tc_total.tris += tc.tris; tc_total.node_trips += tc.node_trips; tc_total.tri_trips += tc.tri_trips;
            steal_rounds += tc.steal_rounds;
            if (wave_leader()) ++ray_trips;
            return b;
        }
        return closest_hit<kBlock, false, COOP>(sc, geom, r, stack, coop, slot, nullptr, helper);
    };
    // BUDGET: a search of the bounce loop that may be cut; `parked` lanes go on with their unfinished walk
    bool parked = false;
    auto search_budget = [&](const Ray& r) {
        wb.resume = parked;
        Best b;
        if (COUNT) {
            TraversalCounts tc;
            b = closest_hit<kBlock, true, false, true>(sc, geom, r, stack, coop, slot, &tc, false, &wb);
            tc_total.nodes += tc.nodes; tc_total.tris += tc.tris; tc_total.node_trips += tc.node_trips; tc_total.tri_trips += tc.tri_trips;
            if (wave_leader()) ++ray_trips;
        } else {
            b = closest_hit<kBlock, false, false, true>(sc, geom, r, stack, coop, slot, nullptr, false, &wb);
        }
        parked = wb.unfinished;
        return b;
    };

    // ---- the camera ray of every pixel of the tile, once (four lanes trace each: coherent, and once per launch)
    f3 camPos;
    uint32_t nvalid;
    {
        const int px = tx0 + (p0 & 3), py = ty0 + (p0 >> 2);
        const bool inside = (px < rp.x1) && (py < rp.y1);
        const Ray ray = camera_ray(rp.cam, px, py);
        camPos = ray.o;
        const TraversalCounts keep = tc_total;
        const Best best = search(ray, COOP && !first); // shared walks: sixteen rays, forty-eight helpers
        if (COUNT && !first && !COOP) tc_total = keep; // one search per pixel in the tallies (shared walks tally once anyway)
        const bool primaryHit = best.obj >= 0 && (first || !COOP);
        HitInfo h;
        h.p = mk3(0.f, 0.f, 0.f); h.n = mk3(0.f, 0.f, 1.f); h.frontFace = false; h.albedo = mk3(0.f, 0.f, 0.f); h.roughness = 1.f; h.metallic = 0.f;
        h.emission = mk3(0.f, 0.f, 0.f); h.ior = 1.f; h.transmission = 0.f;
        if (primaryHit) fill_hit(sc, ray, best, h);
        const bool valid = inside && primaryHit;
        const unsigned long long validMask = __ballot(valid) & 0xffffull;
        nvalid = (uint32_t)__popcll(validMask);
        if (first) {
            float4* rec = cache + p0 * 5;
            rec[0] = make_float4(h.p.x, h.p.y, h.p.z, h.n.x);
            rec[1] = make_float4(h.n.y, h.n.z, h.albedo.x, h.albedo.y);
            rec[2] = make_float4(h.albedo.z, h.roughness, h.metallic, h.emission.x);
            rec[3] = make_float4(h.emission.y, h.emission.z, h.ior, h.transmission);
            rec[4] = make_float4(ray.d.x, ray.d.y, ray.d.z, h.frontFace ? 1.0f : 0.0f);
            if (valid) tile.list[rank_in(validMask)] = (uint32_t)p0;
            if (inside && !primaryHit) {
                // the camera ray leaves the scene: every sample of the pixel is the miss radiance (PathTracer.cpp:59-60)
                add_sample_fx(tile.acc[p0], (EXT && sc.sky_size > 0) ? sky_lookup(sc, ray.d) : bg, (unsigned long long)rp.spp_count);
                if (COUNT) nrays += rp.spp_count;
            }
        }
    }
    __syncthreads();

    // tickets: t -> sample t / nvalid of the (t % nvalid)-th traced pixel; t < 16 * 2^20, so the multiply-high by
    // ceil(2^32 / nvalid) divides exactly. The counter is an LDS word of the tile (a returning LDS atomic). Measured on the
    // mesh Cornell at 1024 / 128 samples and the glass / mirror scene at 256 / 64, in Msamples/s: this form 4181 / 4117 /
    // 10709 / 9670; a wave-uniform counter with ballot + mbcnt ranking 4177 / 4112 / 9778 / 8811 (it costs the roulette
    // build 30 spilled registers); fixed shares per lane 4019 / 3840 / 9434 / 8016 (lanes wait for the longest share);
    // the pixel-major kernel 3900 / 3746 / 10304 / 9562.
    const uint32_t total = nvalid * rp.spp_count;
    const uint32_t magic = nvalid > 1u ? 0xFFFFFFFFu / nvalid + 1u : 0u;
    float ior = 1.0f;
    f3 T = mk3(1.f, 1.f, 1.f), L = mk3(0.f, 0.f, 0.f);
    int bounce = 0;
    uint32_t key = 0u;
    Ray ray; ray.o = camPos; ray.d = mk3(0.f, 0.f, 1.f);
    uint32_t pi = 0u;  // the pixel (0..15) of the lane's current sample
    bool fresh = true; // the lane's next step is the BSDF sample at a camera-ray hit from the tile's records: no search before it
    bool idle = false; // shared walks only: the tickets are gone, the lane helps until the wave's last path has ended
    bool pending = false;        // DEFER: the lane's ray has met the mesh's bounds and waits for the wave's next shared walk
    float heldT = 0.0f; int heldObj = -1; // ... with its closest hit among the other objects
    for (;;) {
        bool end = false;
        HitInfo h;
        Best best;
        bool arrived = !fresh; // the lane has a ray whose closest hit is known after this trip's search
        if (DEFER) {
            if (__all(idle)) break;
            if (!fresh && !pending) {
                best = closest_hit<kBlock, false, false, false, true>(sc, geom, ray, stack, coop, slot);
                if (COUNT && wave_leader()) ++ray_trips;
                pending = meets_tail_mesh(sc, ray, best.t);
                heldT = best.t; heldObj = best.obj;
            }
            // The walk runs once rp.walk_budget lanes wait for it, or when no lane of the wave has anything else to do
            const uint32_t waiting = (uint32_t)__popcll(__ballot(pending));
            const bool walk = waiting >= rp.walk_budget || (waiting != 0u && !__any(!idle && !pending));
            if (walk) {
                TraversalCounts tc; tc.nodes = 0; tc.tris = 0; tc.node_trips = 0; tc.tri_trips = 0; tc.steal_rounds = 0;
                best.t = heldT; best.obj = heldObj; best.tri = -1; best.u = 0.f; best.v = 0.f;
                shared_walk_hit<kBlock, COUNT>(sc, geom, sc.objects[sc.num_objects - 1], ray, pending, best, stack, coop, slot, tc);
                if (COUNT) {
                    tc_total.nodes += tc.nodes; tc_total.tris += tc.tris; tc_total.node_trips += tc.node_trips; tc_total.tri_trips += tc.tri_trips;
                    steal_rounds += tc.steal_rounds;
                }
                pending = false;
            }
            arrived = !fresh && !pending;
        } else if (COOP) {
            if (__all(idle)) break;
            best = search(ray, fresh); // every lane of the wave enters the shared walk; one without a ray helps
        }
        if (arrived) {
            if (CUT) {
                best = search_budget(ray);
                if (parked) continue; // the walk goes on in the wave's next search
            } else if (!COOP) {
                best = search(ray, false);
            }
            if (COUNT) ++nrays;
            if (best.obj < 0) {
                L = L + T * ((EXT && sc.sky_size > 0) ? sky_lookup(sc, ray.d) : bg); // PathTracer.cpp:59-60
                end = true;
            } else {
                fill_hit(sc, ray, best, h);
                L = L + T * h.emission;                    // PathTracer.cpp:76
                if (bounce + 1 >= rp.depth) end = true;    // the next TraceRay would return 0 (PathTracer.cpp:35-36)
            }
            if (end) add_sample_fx(tile.acc[pi], L);
        }
        if (COOP && idle) continue;
        if (DEFER && pending) continue;
        if (fresh || end) {
            const uint32_t t = atomicAdd(&tile.next, 1u);
            if (t >= total) {
                if (!COOP) break;
                idle = true; fresh = true;
                continue;
            }
            uint32_t q;
            if (nvalid == 16u) { q = t >> 4; pi = t & 15u; }
            else { q = nvalid > 1u ? __umulhi(t, magic) : t; pi = tile.list[t - q * nvalid]; }
            const uint32_t pixel = (uint32_t)((ty0 + (int)(pi >> 2)) * W + tx0 + (int)(pi & 3u));
            key = path_key(seed_hash, pixel, rp.spp_begin + q);
            lds_key_hi[slot] = path_key_hi(seed_hash, pixel, rp.spp_begin + q);
            // the camera ray's hit comes from the tile's record
            const float4* rec = cache + pi * 5u;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
            h.p = mk3(q0.x, q0.y, q0.z); h.n = mk3(q0.w, q1.x, q1.y); h.albedo = mk3(q1.z, q1.w, q2.x);
            h.roughness = q2.y; h.metallic = q2.z; h.emission = mk3(q2.w, q3.x, q3.y); h.ior = q3.z; h.transmission = q3.w;
            h.frontFace = q4.w != 0.0f;
            ray.o = camPos; ray.d = mk3(q4.x, q4.y, q4.z);
            T = mk3(1.f, 1.f, 1.f); L = h.emission; ior = 1.0f; bounce = 0;
            fresh = false;
            if (COUNT) ++nrays;
        }
        if (COUNT) { ++shade_lanes; if (wave_leader()) ++shade_trips; }
        f3 w;
        bool done = !sample_bsdf(h, ray, ior, w, key, lds_key_hi[slot], (uint32_t)bounce);
        T = T * w;
        if (EXT && (rp.flags & kFlagRoulette) && bounce + 1 >= 3 && !done) {
            float q = clampf(fmaxf(fmaxf(T.x, T.y), T.z), 0.05f, 1.0f);
            if (!(rand01(key, lds_key_hi[slot], (uint32_t)bounce, 7u) < q)) done = true;
            else T = T * __fdiv_rn(1.0f, q);
        }
        ++bounce;
        if (done) { // the path ends without a next ray: the lane draws its next ticket in its next trip
            add_sample_fx(tile.acc[pi], L);
            fresh = true;
        }
    }
    // The film: one wave updates the workgroup's whole 8x8-pixel block, rows of eight pixels (96 contiguous bytes), as the
    // pixel-major kernel does; four waves writing rows of four pixels each cost 1.6 times the memory-side bytes.
    __syncthreads();
    const int fx = bx0 + (lane & 7), fy = by0 + (lane >> 3);
    const bool writes = wave == 0 && fx < rp.x1 && fy < rp.y1;
    if (writes) {
        const SampleTile& src = reinterpret_cast<const SampleTile*>(lds_raw + stackBytes)[((lane >> 5) << 1) | ((lane >> 2) & 1)];
        const unsigned long long* a4 = src.acc[(((lane >> 3) & 3) << 2) | (lane & 3)];
        const uint32_t pixel = (uint32_t)(fy * W + fx);
        const unsigned long long bad = a4[3];
        float* a = rp.accum + 3 * (size_t)pixel;
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] = bad ? __builtin_nanf("") : a[k] + (float)((double)(long long)a4[k] * kFxInv);
        rp.samples[pixel] += rp.spp_count;
    }
    {
        const int px = fx, py = fy;
        const bool first = writes; // the lanes that stand for a pixel in the sample count
        if (COUNT && rp.counters) {
            unsigned long long c[kNumCounters] = { (first && px < rp.x1 && py < rp.y1) ? rp.spp_count : 0u, nrays, tc_total.nodes, tc_total.tris,
                                                   ray_trips, tc_total.node_trips, tc_total.tri_trips, shade_lanes, shade_trips, steal_rounds };
#pragma unroll
            for (int k = 0; k < kNumCounters; ++k) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) c[k] += __shfl_xor(c[k], off, 64);
                if (lane == 0) atomicAdd(&rp.counters[k], c[k]);
            }
        }
    }
}

// Camera::GetRay probe.
__global__ void camera_rays_kernel(DCamera cam, float* out)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cam.width || y >= cam.height) return;
    Ray r = camera_ray(cam, x, y);
    float* o = out + ((size_t)y * cam.width + x) * 6;
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

// Closest-hit probe: 20 floats per ray {hit, t, p, n, frontFace, material(11)}.
template <int STACK>
__global__ __launch_bounds__(kBlock) void closest_hits_kernel(DScene sc, const float* rays, uint32_t n, float* rec, uint32_t* counts)
{
    __shared__ StackEntry stack[STACK * kBlock];
    __shared__ unsigned long long coop_res[kBlock];
    __shared__ uint32_t coop_mail[kBlock];
    Coop coop; coop.res = coop_res; coop.mail = coop_mail;
    Geom geom; geom.nodes = sc.nodes; geom.tris = sc.tris;
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Ray ray;
    ray.o = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    ray.d = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    TraversalCounts tc;
    Best best = closest_hit<kBlock, true, true>(sc, geom, ray, stack, coop, (int)threadIdx.x, &tc);
    if (counts) { counts[2 * (size_t)i] = tc.nodes; counts[2 * (size_t)i + 1] = tc.tris; }
    float* o = rec + (size_t)i * 20;
    for (int k = 0; k < 20; ++k) o[k] = 0.f;
    if (best.obj < 0) return;
    HitInfo h;
    fill_hit(sc, ray, best, h);
    // raw material fields (emission is reported as colour/strength of the table entry when there is one)
    o[0] = 1.f; o[1] = best.t; o[2] = h.p.x; o[3] = h.p.y; o[4] = h.p.z; o[5] = h.n.x; o[6] = h.n.y; o[7] = h.n.z;
    o[8] = h.frontFace ? 1.f : 0.f;
    o[9] = h.albedo.x; o[10] = h.albedo.y; o[11] = h.albedo.z; o[12] = h.roughness; o[13] = h.metallic;
    o[14] = h.emission.x; o[15] = h.emission.y; o[16] = h.emission.z; // colour * strength
    o[17] = 1.f; o[18] = h.ior; o[19] = h.transmission;
}

// Film::AverageAt (Film.cpp:48-53) for every pixel.
__global__ void average_kernel(const float* accum, const uint32_t* samples, int n, float* avg)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t s = samples[p];
    float fs = (float)s;
    avg[3 * p] = s ? accum[3 * p] / fs : 0.f;
    avg[3 * p + 1] = s ? accum[3 * p + 1] / fs : 0.f;
    avg[3 * p + 2] = s ? accum[3 * p + 2] / fs : 0.f;
}

PT_DEV float srgb_oetf(float c)
{
    float u = fmaxf(0.0f, c);
    return (u <= 0.0031308f) ? (12.92f * u) : (1.055f * powf(u, 1.0f / 2.4f) - 0.055f);
}
// Film::ResolveToRGBA8 (Film.cpp:55-104): 16 B in, 4 B out per pixel.
__global__ void resolve_kernel(const float* accum, const uint32_t* samples, int n, int tonemap, int colourspace, uint32_t* rgba)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t s = samples[p];
    float fs = (float)s;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = s ? accum[3 * p + k] / fs : 0.f;
        if (tonemap == 1) v = v / (1.0f + v);
        if (colourspace == 1) v = srgb_oetf(v);
        c[k] = clampf(v, 0.0f, 1.0f);
    }
    uint32_t r = (uint32_t)lroundf(c[0] * 255.0f), g = (uint32_t)lroundf(c[1] * 255.0f), b = (uint32_t)lroundf(c[2] * 255.0f);
    rgba[p] = r | (g << 8) | (b << 16) | (255u << 24);
}

// ---------------------------------------------------------------- launchers (called from pt_host.cpp)
hipError_t launch_render(const DScene& sc, const DRenderParams& rp, hipStream_t stream)
{
    const bool tiled = rp.tile_stride > 0;
    int tiles = tiled ? rp.tiles_owned * rp.tile_waves : rp.tiles_x * rp.tiles_y; // 8x8 wave-tiles of the launch
    int blocks = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
    if (blocks <= 0) return hipSuccess;
    const bool fits = sc.lds_resident != 0; // the host built LDS-format nodes: the scene fits kLdsBlockBytes
    const size_t geomBytes = fits ? (size_t)(sc.lds_node_f4 + sc.num_tris * 3u) * 16 : 0;
    DRenderParams p = rp;
    p.stack_depth = sc.max_stack < 1 ? 1 : (sc.max_stack > kStackDepth ? kStackDepth : sc.max_stack);
    const size_t lds = ((size_t)p.stack_depth * sizeof(StackEntry) + 4 * kLaneFloats + (fits ? 0 : 32)) * kBlock + geomBytes;
    if (fits && lds > kLdsBlockBytes) return hipErrorInvalidValue; // the host's residency decision uses the same sum
    const bool count = rp.counters != nullptr;
    static const bool debug = getenv("PTAMD_DEBUG") != nullptr;
    const bool ext = (rp.flags & kFlagRoulette) != 0 || sc.sky_size > 0;
    // Many samples per pixel: the sample-major kernel (see render_samples_kernel). Fewer than kSampleMajorMinSpp leave its waves too
    // short-lived; albedo-only and depth < 2 renders end at the camera ray's hit anyway. PTAMD_NO_SAMPLE_MAJOR=1 (read per
    // launch) keeps every launch on the pixel-major kernel.
    const bool noSampleMajor = getenv("PTAMD_NO_SAMPLE_MAJOR") != nullptr;
    // rp.walk_budget (set by the host, 0 = none): the 64ths of a wave's walking lanes at which the LDS kernel cuts a walk and
    // parks what is left of it (see WalkBudget). Measured on the mesh Cornell at 512 samples, Msamples/s: never 4150; 20: 4492;
    // 22: 4506; 24: 4512; 26: 4519; 28: 4536; 30: 4531; 32: 4526; 34: 4519.
    if (p.walk_budget > 63u) p.walk_budget = 63u;
    const size_t ldsPlain = (size_t)p.stack_depth * sizeof(StackEntry) * kBlock + kSamplesTileBytes + kSamplesCacheBytes + kSamplesKeyBytes
                            + (fits ? geomBytes : (size_t)12 * kBlock);
    const bool budget = fits && sc.tail_mesh != 0u && p.walk_budget != 0u && p.stack_depth <= 31 // a parked stack height has five bits
                        && ldsPlain + kSamplesParkBytes <= kLdsBlockBytes;                       // the parked walks must not cost the scene this kernel
    // Global-memory scenes whose only mesh is their last object (sc.tail_mesh bit 1): the lanes whose rays meet its bounds wait
    // until rp.walk_budget of them (1..64) make a shared walk worth its trips (DEFER in render_samples_kernel).
    const bool defer = !fits && (sc.tail_mesh & 2u) != 0u && rp.walk_budget != 0u;
    if (defer) p.walk_budget = rp.walk_budget > 64u ? 64u : rp.walk_budget;
    const size_t ldsSamples = ldsPlain + (budget ? kSamplesParkBytes : 0u);
    if (!noSampleMajor && rp.spp_count >= kSampleMajorMinSpp && rp.spp_count <= kSampleMajorMaxSpp && rp.depth >= 2 && !rp.albedo_only
        && ldsSamples <= kLdsBlockBytes) {
        const int n8 = tiled ? rp.tiles_owned * rp.tile_waves : rp.tiles_x * rp.tiles_y; // 8x8-pixel blocks, one workgroup each
        const int nb = ((n8 + 63) / 64) * 64;                                            // dealt to the XCDs eight neighbours at a time
#define PT_LAUNCH_S(LDSG, COUNT, EXT, TILED, BUDGET)                                                                   \
        do {                                                                                                          \
            if (debug) fprintf(stderr, "ptamd: render_samples_kernel<%d,%d,%d,%d,%d> stack %d, lds %zu B/block, %d blocks, budget %u\n", (int)LDSG, (int)COUNT, (int)EXT, (int)TILED, (int)BUDGET, p.stack_depth, ldsSamples, nb, p.walk_budget); \
            hipLaunchKernelGGL((render_samples_kernel<LDSG, COUNT, EXT, TILED, BUDGET>), dim3(nb), dim3(kBlock), ldsSamples, stream, sc, p); \
        } while (0)
        switch ((budget || defer ? 16 : 0) | (tiled ? 8 : 0) | (fits ? 4 : 0) | (count ? 2 : 0) | (ext ? 1 : 0)) {
        case 0: PT_LAUNCH_S(false, false, false, false, false); break;
        case 1: PT_LAUNCH_S(false, false, true, false, false); break;
        case 2: PT_LAUNCH_S(false, true, false, false, false); break;
        case 3: PT_LAUNCH_S(false, true, true, false, false); break;
        case 4: PT_LAUNCH_S(true, false, false, false, false); break;
        case 5: PT_LAUNCH_S(true, false, true, false, false); break;
        case 6: PT_LAUNCH_S(true, true, false, false, false); break;
        case 7: PT_LAUNCH_S(true, true, true, false, false); break;
        case 8: PT_LAUNCH_S(false, false, false, true, false); break;
        case 9: PT_LAUNCH_S(false, false, true, true, false); break;
        case 10: PT_LAUNCH_S(false, true, false, true, false); break;
        case 11: PT_LAUNCH_S(false, true, true, true, false); break;
        case 12: PT_LAUNCH_S(true, false, false, true, false); break;
        case 13: PT_LAUNCH_S(true, false, true, true, false); break;
        case 14: PT_LAUNCH_S(true, true, false, true, false); break;
        case 15: PT_LAUNCH_S(true, true, true, true, false); break;
        case 16: PT_LAUNCH_S(false, false, false, false, true); break;
        case 17: PT_LAUNCH_S(false, false, true, false, true); break;
        case 18: PT_LAUNCH_S(false, true, false, false, true); break;
        case 19: PT_LAUNCH_S(false, true, true, false, true); break;
        case 20: PT_LAUNCH_S(true, false, false, false, true); break;
        case 21: PT_LAUNCH_S(true, false, true, false, true); break;
        case 22: PT_LAUNCH_S(true, true, false, false, true); break;
        case 23: PT_LAUNCH_S(true, true, true, false, true); break;
        case 24: PT_LAUNCH_S(false, false, false, true, true); break;
        case 25: PT_LAUNCH_S(false, false, true, true, true); break;
        case 26: PT_LAUNCH_S(false, true, false, true, true); break;
        case 27: PT_LAUNCH_S(false, true, true, true, true); break;
        case 28: PT_LAUNCH_S(true, false, false, true, true); break;
        case 29: PT_LAUNCH_S(true, false, true, true, true); break;
        case 30: PT_LAUNCH_S(true, true, false, true, true); break;
        default: PT_LAUNCH_S(true, true, true, true, true); break;
        }
#undef PT_LAUNCH_S
        return hipGetLastError();
    }
#define PT_LAUNCH(LDSG, COUNT, EXT, TILED)                                                                                   \
    do {                                                                                                              \
        if (debug) {                                                                                                  \
            int nb = 0;                                                                                               \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<LDSG, COUNT, EXT, TILED>, kBlock, lds);    \
            fprintf(stderr, "ptamd: render_kernel<%d,%d,%d,%d> stack %d, lds %zu B/block, %d blocks/CU resident\n", (int)LDSG, (int)COUNT, (int)EXT, (int)TILED, p.stack_depth, lds, nb); \
        }                                                                                                             \
        hipLaunchKernelGGL((render_kernel<LDSG, COUNT, EXT, TILED>), dim3(blocks), dim3(kBlock), lds, stream, sc, p);        \
    } while (0)
    const int variant = (tiled ? 8 : 0) | (fits ? 4 : 0) | (count ? 2 : 0) | (ext ? 1 : 0);
    switch (variant) {
    case 0: PT_LAUNCH(false, false, false, false); break;
    case 1: PT_LAUNCH(false, false, true, false); break;
    case 2: PT_LAUNCH(false, true, false, false); break;
    case 3: PT_LAUNCH(false, true, true, false); break;
    case 4: PT_LAUNCH(true, false, false, false); break;
    case 5: PT_LAUNCH(true, false, true, false); break;
    case 6: PT_LAUNCH(true, true, false, false); break;
    case 7: PT_LAUNCH(true, true, true, false); break;
    case 8: PT_LAUNCH(false, false, false, true); break;
    case 9: PT_LAUNCH(false, false, true, true); break;
    case 10: PT_LAUNCH(false, true, false, true); break;
    case 11: PT_LAUNCH(false, true, true, true); break;
    case 12: PT_LAUNCH(true, false, false, true); break;
    case 13: PT_LAUNCH(true, false, true, true); break;
    case 14: PT_LAUNCH(true, true, false, true); break;
    default: PT_LAUNCH(true, true, true, true); break;
    }
#undef PT_LAUNCH
    return hipGetLastError();
}
hipError_t launch_camera_rays(const DCamera& cam, float* d_out, hipStream_t stream)
{
    dim3 grid((cam.width + 63) / 64, cam.height);
    hipLaunchKernelGGL(camera_rays_kernel, grid, dim3(64), 0, stream, cam, d_out);
    return hipGetLastError();
}
hipError_t launch_closest_hits(const DScene& sc, const float* d_rays, uint32_t n, float* d_rec, uint32_t* d_counts, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(closest_hits_kernel<kStackDepth>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sc, d_rays, n, d_rec, d_counts);
    return hipGetLastError();
}
hipError_t launch_average(const float* accum, const uint32_t* samples, int n, float* avg, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(average_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, samples, n, avg);
    return hipGetLastError();
}
hipError_t launch_resolve(const float* accum, const uint32_t* samples, int n, int tonemap, int colourspace, uint32_t* rgba, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, samples, n, tonemap, colourspace, rgba);
    return hipGetLastError();
}
int max_stack_depth() { return kStackDepth; }

// Counters of ptamd_render_counted_noalloc: a counting launch tallies into this array instead of into memory that the
// library allocated, so that entry point calls no allocator at all. 80 bytes per GPU, held until the process ends.
// Every call on a GPU shares the array (calls that overlap on one GPU race) and clears it before its launch. In host
// code the array's name is only a handle for the symbol lookup below; kernels, memsets and copies take the address
// that the lookup returned.
static __device__ __align__(256) unsigned long long scratch[kNumCounters];
constexpr int kMaxDevices = 64;
static unsigned long long* g_scratch_address[kMaxDevices] = {};
// Address of `scratch` on `device`, which must be the current device (each GPU holds its own copy at its own address).
hipError_t scratch_counters(int device, unsigned long long** out)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_scratch_address[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_address[device] = static_cast<unsigned long long*>(p);
    }
    *out = g_scratch_address[device];
    return hipSuccess;
}

} // namespace ptamd
