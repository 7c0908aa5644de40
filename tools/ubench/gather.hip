// Gather micro-benchmark for gfx950: how a wave should read 64-byte records at random addresses.
//   own   : every lane reads the four 16-byte pieces of its own record (4 loads, 64 different cache lines each)
//   quad  : the four lanes of a quad read the records of the quad's lanes one after the other, each lane one piece
//           (4 loads, 16 different lines each; the pieces end up spread over the quad)
//   quadT : quad + the exchange through LDS that gives every lane its own record back
// Prints nanoseconds per wave-iteration per CU for a table that fits L2 and one that does not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 512;

__device__ inline uint32_t next(uint32_t x) { return x * 1664525u + 1013904223u; }

__global__ __launch_bounds__(256) void k_own(const float4* __restrict__ t, uint32_t mask, float* out)
{
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        r = next(r);
        const float4* n = t + (size_t)((r >> 8) & mask) * 4;
        float4 a = n[0], b = n[1], c = n[2], d = n[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_quad(const float4* __restrict__ t, uint32_t mask, float* out)
{
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63, piece = lane & 3;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        r = next(r);
        const uint32_t idx = (r >> 8) & mask;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t other = (uint32_t)__shfl((int)idx, (lane & ~3) | k, 64);
            v[k] = t[(size_t)other * 4 + piece];
        }
        acc += v[0].x + v[1].y + v[2].z + v[3].w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_quadT(const float4* __restrict__ t, uint32_t mask, float* out)
{
    __shared__ float4 ex[256 * 4];
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63, piece = lane & 3, wbase = threadIdx.x & ~63;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        r = next(r);
        const uint32_t idx = (r >> 8) & mask;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int owner = (lane & ~3) | k;
            const uint32_t other = (uint32_t)__shfl((int)idx, owner, 64);
            ex[(wbase + owner) * 4 + piece] = t[(size_t)other * 4 + piece];
        }
        __builtin_amdgcn_wave_barrier();
        const float4* mine = ex + threadIdx.x * 4;
        float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
        acc += a.x + b.y + c.z + d.w;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class K> static void run(const char* name, K kernel, const float4* t, uint32_t mask, float* out, int blocks)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, t, mask, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, t, mask, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double waveIters = (double)blocks * 4 * ITERS;
    printf("  %-6s %8.3f ms  %7.1f ns per wave-iteration per CU  %7.1f GB/s of records\n", name, ms, ms * 1e6 / (waveIters / 256.0), waveIters * 64 * 64 / (ms * 1e6));
}

int main()
{
    const int blocks = 256 * 8;
    float* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (uint32_t records : { 1u << 14, 1u << 21 }) {
        float4* t; CHECK(hipMalloc(&t, (size_t)records * 64));
        CHECK(hipMemset(t, 0, (size_t)records * 64));
        printf("%u records of 64 B (%.0f MB)\n", records, records * 64 / 1048576.0);
        run("own", k_own, t, records - 1, out, blocks);
        run("quad", k_quad, t, records - 1, out, blocks);
        run("quadT", k_quadT, t, records - 1, out, blocks);
        CHECK(hipFree(t));
    }
    return 0;
}
