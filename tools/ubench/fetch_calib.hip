// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the global-memory BVH walks: every lane of a
// wave reads a record at a random place of a table far larger than L2 (and, for the 1 GiB table, than the Infinity Cache).
//   rec64  : 64-byte records, 64-byte aligned, read as four 16-byte loads (a two-child node visit)
//   rec80  : 80-byte records, 16-byte aligned, read as five 16-byte loads (an eight-child node visit)
//   piece16: one 16-byte load per lane (a single gather)
//   stream : every lane reads consecutive 16-byte pieces (the wide coalesced pattern whose factor of 2 is known)
// Each kernel is launched ONCE per table; the program prints the bytes each launch requested. Run it under
//   rocprofv3 --pmc FETCH_SIZE -- tools/ubench/fetch_calib     and     rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- ...
// and divide: tools/fetch_calib_summary.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 256;
__device__ inline uint32_t next(uint32_t x) { return x * 1664525u + 1013904223u; }

__global__ __launch_bounds__(256) void calib_rec64(const float4* __restrict__ t, uint32_t mask, float* out)
{
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        r = next(r);
        const float4* n = t + (size_t)((r >> 6) & mask) * 4;
        float4 a = n[0], b = n[1], c = n[2], d = n[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_rec80(const float4* __restrict__ t, uint32_t mask, float* out)
{
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        r = next(r);
        const float4* n = t + (size_t)((r >> 6) & mask) * 5;
        float4 a = n[0], b = n[1], c = n[2], d = n[3], e = n[4];
        acc += a.x + b.y + c.z + d.w + e.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_piece16(const float4* __restrict__ t, uint32_t mask, float* out)
{
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        r = next(r);
        float4 a = t[(size_t)((r >> 6) & mask)];
        acc += a.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ t, uint32_t mask, float* out)
{
    float acc = 0.f;
    const size_t base = (size_t)blockIdx.x * 256 * ITERS;
    for (int i = 0; i < ITERS; ++i) acc += t[(base + (size_t)i * 256 + threadIdx.x) & mask].x;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 16;
    float* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (unsigned shift : { 27u, 30u }) { // 128 MiB, 1 GiB
        const size_t bytes = (size_t)1 << shift;
        float4* t; CHECK(hipMalloc(&t, bytes + 256));
        CHECK(hipMemset(t, 0, bytes + 256));
        CHECK(hipDeviceSynchronize());
        const double lanes = (double)blocks * 256 * ITERS;
        const uint32_t rec64 = (uint32_t)(bytes / 64) - 1, rec80 = (uint32_t)(bytes / 128) - 1 /* power of two below bytes / 80 */, p16 = (uint32_t)(bytes / 16) - 1;
        hipLaunchKernelGGL(calib_rec64, dim3(blocks), dim3(256), 0, 0, t, rec64, out); CHECK(hipDeviceSynchronize());
        printf("table %zu MiB rec64 requested_bytes %.0f\n", bytes >> 20, lanes * 64);
        hipLaunchKernelGGL(calib_rec80, dim3(blocks), dim3(256), 0, 0, t, rec80, out); CHECK(hipDeviceSynchronize());
        printf("table %zu MiB rec80 requested_bytes %.0f\n", bytes >> 20, lanes * 80);
        hipLaunchKernelGGL(calib_piece16, dim3(blocks), dim3(256), 0, 0, t, p16, out); CHECK(hipDeviceSynchronize());
        printf("table %zu MiB piece16 requested_bytes %.0f\n", bytes >> 20, lanes * 16);
        hipLaunchKernelGGL(calib_stream, dim3(blocks), dim3(256), 0, 0, t, p16, out); CHECK(hipDeviceSynchronize());
        printf("table %zu MiB stream requested_bytes %.0f\n", bytes >> 20, lanes * 16);
        CHECK(hipFree(t));
    }
    return 0;
}
