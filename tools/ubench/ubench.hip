// Instruction-throughput micro-benchmarks for gfx950: plain vs packed fp32 multiply-add, min/max, and LDS 16-byte
// reads at full and partial lane occupancy. Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 16384;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b)
{
    v2f av = { a, a }, bv = { b, b };
    v2f x0 = { (float)threadIdx.x, 1.f }, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
        x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
        x4 = __builtin_elementwise_fma(x4, av, bv); x5 = __builtin_elementwise_fma(x5, av, bv);
        x6 = __builtin_elementwise_fma(x6, av, bv); x7 = __builtin_elementwise_fma(x7, av, bv);
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(256) void k_minmax(float* out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = fminf(x0, a) ; x1 = fmaxf(x1, b); x2 = fminf(x2, a); x3 = fmaxf(x3, b);
        x4 = fminf(x4, a) ; x5 = fmaxf(x5, b); x6 = fminf(x6, a); x7 = fmaxf(x7, b);
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// single dependent chains: latency per instruction
__global__ __launch_bounds__(256) void k_fma_dep(float* out, float a, float b)
{
    float x0 = threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
        x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0;
}
__global__ __launch_bounds__(256) void k_pkfma_dep(float* out, float a, float b)
{
    v2f av = { a, a }, bv = { b, b };
    v2f x0 = { (float)threadIdx.x, 1.f };
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_elementwise_fma(x0, av, bv); x0 = __builtin_elementwise_fma(x0, av, bv);
        x0 = __builtin_elementwise_fma(x0, av, bv); x0 = __builtin_elementwise_fma(x0, av, bv);
        x0 = __builtin_elementwise_fma(x0, av, bv); x0 = __builtin_elementwise_fma(x0, av, bv);
        x0 = __builtin_elementwise_fma(x0, av, bv); x0 = __builtin_elementwise_fma(x0, av, bv);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0.x + x0.y;
}
// partial exec mask: does the vector ALU skip quarter-waves with no active lane?
__global__ __launch_bounds__(256) void k_fma_mask(float* out, float a, float b, int active)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if ((int)(threadIdx.x & 63) < active) {
        for (int i = 0; i < ITERS; ++i) {
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// 16-byte LDS reads, every lane a different row (column-major layout as the render kernel uses), `active` lanes of 64
__global__ __launch_bounds__(256) void k_lds(float* out, int active, int stride)
{
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0.f;
    int lane = threadIdx.x & 63;
    if (lane < active) {
        uint32_t idx = (threadIdx.x * 7u) & 255u;
        for (int i = 0; i < ITERS; ++i) {
            float4 v0 = buf[idx], v1 = buf[idx + stride], v2 = buf[idx + 2 * stride], v3 = buf[idx + 3 * stride];
            acc += (v0.x + v0.y + v0.z + v0.w) + (v1.x + v1.y + v1.z + v1.w) + (v2.x + v2.y + v2.z + v2.w) + (v3.x + v3.y + v3.z + v3.w);
            idx = (idx + (uint32_t)__float_as_int(v0.y)) & 255u; // dependent address, like a tree walk (adds 1.0f bits & 255 = 0)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class F> double time_ms(F launch)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    for (int wavesPerSimd : { 1, 4, 5, 8 }) {
        int blocks = cus * wavesPerSimd; // 256 threads = 4 waves = one per SIMD
        auto report = [&](const char* name, double ms, double instrPerIter) {
            double cyc = ms * 1e-3 * ghz * 1e9; // cycles elapsed
            double perSimdInstr = (double)ITERS * instrPerIter * wavesPerSimd; // wave-instructions each SIMD issued
            printf("  %-28s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", name, wavesPerSimd, ms, cyc / perSimdInstr);
        };
        report("v_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), 8);
        report("v_pk_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), 8);
        report("v_fma_f32 dependent", time_ms([&] { hipLaunchKernelGGL(k_fma_dep, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), 8);
        report("v_pk_fma_f32 dependent", time_ms([&] { hipLaunchKernelGGL(k_pkfma_dep, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), 8);
        for (int act : { 48, 32, 16, 8 }) {
            char nm[64]; snprintf(nm, sizeof nm, "v_fma_f32 lanes 0..%d", act - 1);
            report(nm, time_ms([&] { hipLaunchKernelGGL(k_fma_mask, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, act); }), 8);
        }
        report("v_min/max_f32", time_ms([&] { hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, 0, out, 1e30f, -1e30f); }), 8);
        for (int active : { 64, 16 })
            for (int stride : { 256 }) {
                char nm[64]; snprintf(nm, sizeof nm, "ds_read_b128 x4 act %d str %d", active, stride);
                report(nm, time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, out, active, stride); }), 4);
            }
    }
    return 0;
}
