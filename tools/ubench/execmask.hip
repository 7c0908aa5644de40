// Does a vector instruction cost less when part of the wave is masked off? Eight independent v_fma_f32 chains per
// wave, 8 waves per SIMD, run under different EXEC masks (the branch around the loop sets the mask; inline assembly
// keeps the instruction stream identical). If a wave64 instruction is issued as passes over groups of lanes and a pass
// whose lanes are all off is skipped, masks that empty whole groups run faster than masks that spread the same
// number of live lanes over all groups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 8192;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b, unsigned long long mask)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if ((mask >> (threadIdx.x & 63)) & 1ull) {
        for (int i = 0; i < ITERS; ++i) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
            OP(x0); OP(x1); OP(x2); OP(x3); OP(x4); OP(x5); OP(x6); OP(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const struct { const char* name; unsigned long long mask; } cases[] = {
        { "all 64 lanes", ~0ull }, { "lanes 0-31", 0xffffffffull }, { "lanes 32-63", 0xffffffff00000000ull }, { "lanes 0-15", 0xffffull },
        { "lanes 0-15 + 32-47", 0x0000ffff0000ffffull }, { "every other lane", 0x5555555555555555ull }, { "every fourth lane", 0x1111111111111111ull },
        { "lane 0", 1ull }, { "lanes 16-31", 0xffff0000ull } };
    int clockKHz = 0; CHECK(hipDeviceGetAttribute(&clockKHz, hipDeviceAttributeClockRate, 0));
    for (const auto& c : cases) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_fma, dim3(256 * 8), dim3(256), 0, 0, out, 1.0001f, 0.5f, c.mask);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_fma, dim3(256 * 8), dim3(256), 0, 0, out, 1.0001f, 0.5f, c.mask);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("  %-20s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", c.name, ms, ms * 1e-3 * clockKHz * 1e3 / (8.0 * ITERS * 8));
    }
    return 0;
}
