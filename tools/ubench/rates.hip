// Issue cost of single vector instructions on gfx950: cycles per wave-instruction per SIMD with 8 waves per SIMD,
// eight independent chains per wave (inline assembly, so the instruction measured is the one named).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 8192;

#define KERNEL(name, text)                                                                                        \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b)                                     \
    {                                                                                                             \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < ITERS; ++i) {                                                                         \
            asm volatile(text : "+v"(x0) : "v"(a), "v"(b)); asm volatile(text : "+v"(x1) : "v"(a), "v"(b));       \
            asm volatile(text : "+v"(x2) : "v"(a), "v"(b)); asm volatile(text : "+v"(x3) : "v"(a), "v"(b));       \
            asm volatile(text : "+v"(x4) : "v"(a), "v"(b)); asm volatile(text : "+v"(x5) : "v"(a), "v"(b));       \
            asm volatile(text : "+v"(x6) : "v"(a), "v"(b)); asm volatile(text : "+v"(x7) : "v"(a), "v"(b));       \
        }                                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                      \
    }
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_imax, "v_max_i32 %0, %0, %1")
KERNEL(k_imax3, "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_umin, "v_min_u32 %0, %0, %1")
KERNEL(k_iadd, "v_add_u32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp, "v_cmp_ge_f32 vcc, %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_mix_lo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")
KERNEL(k_mix_hi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
KERNEL(k_cvt_ub1, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_sin, "v_sin_f32 %0, %0")
KERNEL(k_cvt_u32, "v_cvt_f32_u32 %0, %0")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[36:37]")
KERNEL(k_cmp_cnd, "v_cmp_ge_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmp_e64_cnd, "v_cmp_ge_f32_e64 s[36:37], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[36:37]")
KERNEL(k_cmp_sgpr, "v_cmp_ge_f32_e64 s[36:37], %0, %1")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")

template <class K> static void run(const char* name, K kernel, float* out)
{
    const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    int clockKHz = 0; CHECK(hipDeviceGetAttribute(&clockKHz, hipDeviceAttributeClockRate, 0));
    const double instrPerSimd = 8.0 * ITERS * 8; // waves per SIMD * iterations * chains
    printf("  %-14s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms, ms * 1e-3 * clockKHz * 1e3 / instrPerSimd, clockKHz * 1e-6);
}
int main()
{
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
#define RUN(k) run(#k, k, out)
    RUN(k_fma); RUN(k_mul); RUN(k_add); RUN(k_max); RUN(k_min); RUN(k_max3); RUN(k_min3); RUN(k_med3); RUN(k_imax); RUN(k_imax3); RUN(k_umin);
    RUN(k_iadd); RUN(k_and); RUN(k_cndmask); RUN(k_cmp); RUN(k_rcp); RUN(k_mul24); RUN(k_mullo); RUN(k_lshladd); RUN(k_mov); RUN(k_mix_lo); RUN(k_mix_hi); RUN(k_cvt_ub1); RUN(k_cvt_f16); RUN(k_bfe); RUN(k_perm);
    RUN(k_cvt_ub0); RUN(k_and_or); RUN(k_lshl_or); RUN(k_bfi); RUN(k_sqrt); RUN(k_rsq); RUN(k_sin); RUN(k_cvt_u32); RUN(k_sub); RUN(k_cnd_e64); RUN(k_cmp_cnd); RUN(k_cmp_e64_cnd); RUN(k_cmp_sgpr); RUN(k_lshr); RUN(k_fmac);
    return 0;
}
