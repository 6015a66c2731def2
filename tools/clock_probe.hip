// What is a shader cycle worth on this box?  A kernel that does nothing but issue independent v_fma_f32 (8 chains per
// thread, 8 wavefronts per SIMD, every SIMD of the device) for ~40 ms: the VALU issues one wavefront instruction per 4
// cycles, so  effective clock = instructions per SIMD x 4 / elapsed.  Compare with `rocm-smi --showclocks` sampled
// meanwhile and with SQ_CYCLES / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_VALU of a counter pass over the same binary
// (the counters' cycles against the reported clock).
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe [fp64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kChains = 8;
__global__ void __launch_bounds__(256) k_fma32(float *out, float a, float b, int iters)
{
    float x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < kChains; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_fma64(float *out, double a, double b, int iters)
{
    double x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = a + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < kChains; ++i) s += x[i];
    if (s == 12345.678) out[0] = (float)s;
}
int main(int argc, char **argv)
{
    const bool f64 = argc > 1 && !std::strcmp(argv[1], "fp64");
    int ncu = 256;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    float *out = nullptr;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = ncu * 8; // 8 blocks of 4 wavefronts per compute unit = 8 wavefronts per SIMD
    const int iters = 400000;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        if (f64) hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        else hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1e-9f, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = 8.0 * (double)iters * kChains; // wavefront instructions per SIMD
        printf("%s rep %d: %.2f ms, %.0f instructions per SIMD -> effective clock %.3f GHz at 4 cycles per instruction\n",
               f64 ? "v_fma_f64" : "v_fma_f32", rep, ms, per_simd, per_simd * 4.0 / (ms * 1e6));
    }
    return 0;
}
