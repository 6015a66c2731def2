// Issue-rate microbenchmark for the gfx950 VALU instructions the step kernel is made of.
// Every kernel executes the same number of one instruction per wave (8 independent chains, 8 waves per SIMD),
// so time ratios are issue-cost ratios.   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIter = 2048, kChains = 8;

#define KERNEL32(name, ASM)                                                              \
__global__ void __launch_bounds__(256) name(float *out, float a, float b)                \
{                                                                                         \
    float x[kChains];                                                                     \
    for (int i = 0; i < kChains; ++i) x[i] = a + threadIdx.x * 1e-3f + i;                 \
    for (int it = 0; it < kIter; ++it) {                                                  \
        _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
    }                                                                                     \
    float s = 0; for (int i = 0; i < kChains; ++i) s += x[i];                             \
    if (s == 12345.678f) out[0] = s;                                                      \
}
#define KERNEL64(name, ASM)                                                              \
__global__ void __launch_bounds__(256) name(float *out, double a, double b)              \
{                                                                                         \
    double x[kChains];                                                                    \
    for (int i = 0; i < kChains; ++i) x[i] = a + threadIdx.x * 1e-3 + i;                  \
    for (int it = 0; it < kIter; ++it) {                                                  \
        _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
    }                                                                                     \
    double s = 0; for (int i = 0; i < kChains; ++i) s += x[i];                            \
    if (s == 12345.678) out[0] = (float)s;                                                \
}
// f32 -> f64 -> f32 round trip pair counts as two instructions
#define KERNELCVT(name)                                                                  \
__global__ void __launch_bounds__(256) name(float *out, float a, float b)                \
{                                                                                         \
    float x[kChains]; double d[kChains];                                                  \
    for (int i = 0; i < kChains; ++i) x[i] = a + threadIdx.x * 1e-3f + i;                 \
    for (int it = 0; it < kIter / 2; ++it) {                                              \
        _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(x[i])); \
        _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x[i]) : "v"(d[i])); \
    }                                                                                     \
    float s = 0; for (int i = 0; i < kChains; ++i) s += x[i];                             \
    if (s == 12345.678f) out[0] = s;                                                      \
}

KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL32(k_rsq_f32, "v_rsq_f32 %0, %0")
KERNEL32(k_log_f32, "v_log_f32 %0, %0")
KERNEL32(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL32(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
KERNEL32(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %2")
KERNEL32(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_f32, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL32(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cndmask_dep, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cmp_s, "v_cmp_lt_f32_e64 s[20:21], %0, %1")
KERNEL32(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL32(k_mul_abs, "v_mul_f32_e64 %0, |%0|, %1")
KERNEL32(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_cvt_i32, "v_cvt_f32_i32 %0, %0")
KERNEL32(k_mul_lit, "v_mul_f32 %0, 0x3f8ccccd, %0")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1")
KERNELCVT(k_cvt_pair)

template <class K, class A> float run(K k, A a, A b, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 * 8 / 4; // 8 waves per SIMD, 4 waves per block
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, a, b);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, a, b);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 4));
    const double n_inst = (double)kIter * kChains;          // per wave
    const float base = run(k_fma_f32, 1.0001f, 0.5f, out);
    printf("%-16s %8.3f ms  %.2f cycles/inst/wave at 2.4 GHz (8 waves per SIMD)  ratio 1.00\n", "v_fma_f32", base,
           base * 1e-3 * 2.4e9 / (n_inst * 8));
#define R32(k) { float t = run(k, 1.0001f, 0.5f, out); printf("%-16s %8.3f ms  ratio %.2f\n", #k, t, t / base); }
#define R64(k) { float t = run(k, 1.0001, 0.5, out); printf("%-16s %8.3f ms  ratio %.2f\n", #k, t, t / base); }
    R32(k_mul_f32) R32(k_add_f32) R32(k_rcp_f32) R32(k_sqrt_f32) R32(k_rsq_f32) R32(k_log_f32) R32(k_exp_f32)
    R32(k_div_scale) R32(k_div_fmas) R32(k_div_fixup) R32(k_cndmask) R32(k_cmp_f32) R32(k_cmp_class) R32(k_mov) R32(k_and)
    R32(k_lshl_add_u32) R32(k_cvt_pair) R32(k_cndmask_s) R32(k_cndmask_dep) R32(k_cmp_s) R32(k_max_f32) R32(k_mul_abs) R32(k_fmac_f32) R32(k_add_u32) R32(k_lshlrev) R32(k_cvt_i32) R32(k_mul_lit) R32(k_bcnt) R32(k_ldexp)
    R64(k_fma_f64) R64(k_mul_f64) R64(k_add_f64) R64(k_rcp_f64) R64(k_cmp_f64) R64(k_lshl_add_u64) R64(k_pk_fma) R64(k_pk_mul) R64(k_pk_add) R64(k_mov_b64)
    return 0;
}
