// Dependent-issue latency of gfx950 VALU instructions: ONE wavefront per SIMD (256 workgroups of 64 threads on 256 CUs...
// one block per CU, one wave), C independent chains of the same instruction; cycles per instruction = time * clock /
// (iterations * C).  With C = 1 it is the latency of a dependent instruction, with C large the issue cost.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIter = 4096;

#define KERNEL(name, T, ASM)                                                                 \
template <int C> __global__ void __launch_bounds__(64) name(float *out, T a, T b, unsigned long long *cyc) \
{                                                                                             \
    T x[C];                                                                                   \
    for (int i = 0; i < C; ++i) x[i] = a + (T)(threadIdx.x * 1e-3) + (T)i;                    \
    const unsigned long long c0 = clock64();                                                  \
    for (int it = 0; it < kIter; ++it) {                                                      \
        _Pragma("unroll") for (int i = 0; i < C; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
    }                                                                                         \
    const unsigned long long c1 = clock64();                                                  \
    T s = 0; for (int i = 0; i < C; ++i) s += x[i];                                           \
    if (s == (T)12345.678) out[0] = (float)s;                                                 \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;                                  \
}
KERNEL(k_fma32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul32, float, "v_mul_f32 %0, %0, %1")
KERNEL(k_rcp32, float, "v_rcp_f32 %0, %0")
KERNEL(k_log32, float, "v_log_f32 %0, %0")
KERNEL(k_fma64, double, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_rcp64, double, "v_rcp_f64 %0, %0")
KERNEL(k_cmpcnd, float, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cvt, float, "v_cvt_f64_f32 %0, %0" )

template <class K, class T> int run(const char *name, K k, T a, T b, int C, float *out, unsigned long long *cyc, int per)
{
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, a, b, cyc);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, a, b, cyc);
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-10s chains %d: %6.2f clock64 ticks per instruction\n", name, C, (double)h / ((double)kIter * C * per));
    return 0;
}
int main()
{
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 4)); CHECK(hipMalloc(&cyc, 8));
#define RUN(name, k, T, per) run(name, k<1>, (T)1.0001, (T)0.5, 1, out, cyc, per); run(name, k<2>, (T)1.0001, (T)0.5, 2, out, cyc, per); run(name, k<4>, (T)1.0001, (T)0.5, 4, out, cyc, per); run(name, k<8>, (T)1.0001, (T)0.5, 8, out, cyc, per);
    RUN("fma_f32", k_fma32, float, 1)
    RUN("mul_f32", k_mul32, float, 1)
    RUN("rcp_f32", k_rcp32, float, 1)
    RUN("log_f32", k_log32, float, 1)
    RUN("fma_f64", k_fma64, double, 1)
    RUN("rcp_f64", k_rcp64, double, 1)
    RUN("cmp+cnd", k_cmpcnd, float, 2)
    return 0;
}
