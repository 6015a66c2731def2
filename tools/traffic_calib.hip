// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the routing kernels (the MI355X
// guide calibrates FETCH_SIZE -- it reports half the bytes -- for 16-byte-per-lane streaming reads only and calls other widths
// and WRITE_SIZE uncalibrated).  Four kernels that move a KNOWN number of bytes, each far larger than the 256 MB Infinity Cache:
//   read4     4 bytes per lane, unit stride (how the step kernels read their columns and time rows)
//   read16    16 bytes per lane, unit stride (the guide's pattern, as a cross-check)
//   write4    4 bytes per lane, unit stride (the time-major rows)
//   write96   96-byte runs at a 3 456-byte stride, 16-byte pieces (a thread's 8 staged steps of out[row][step][q,v,d])
// Run under two counter-only passes (tools/traffic_calib.sh); the factors are known bytes / (counter x 1024).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) read4(const float *__restrict__ x, float *__restrict__ out, size_t n)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
    if (acc == 12345.678f) out[0] = acc; // (keeps the loads)
}
__global__ void __launch_bounds__(256) read16(const float4 *__restrict__ x, float *__restrict__ out, size_t n)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) write4(float *__restrict__ x, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = (float)i;
}
// rows of 864 floats (288 steps x 3); thread r writes the 24 floats of steps [8 k, 8 k + 8) of row r as six 16-byte pieces
__global__ void __launch_bounds__(256) write96(float *__restrict__ out, size_t nrows, int k)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    float4 *dst = reinterpret_cast<float4 *>(out + r * 864 + (size_t)k * 24);
    for (int j = 0; j < 6; ++j) dst[j] = make_float4((float)r, (float)j, 0.0f, 1.0f);
}

int main()
{
    const size_t n = (size_t)512 << 20; // 512 Mi floats = 2 GiB per pass
    float *x = nullptr, *o = nullptr;
    if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&o, 256) != hipSuccess) return 1;
    (void)hipMemset(x, 0, n * 4);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(read4, dim3(8192), dim3(256), 0, 0, x, o, n);
    hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const float4 *)x, o, n / 4);
    hipLaunchKernelGGL(write4, dim3(8192), dim3(256), 0, 0, x, n);
    const size_t nrows = n / 864;
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(write96, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, 0, x, nrows, k);
    (void)hipDeviceSynchronize();
    std::printf("bytes read4 %zu read16 %zu write4 %zu write96 %zu (per launch, 4 launches)\n", n * 4, n * 4, n * 4, nrows * 96);
    (void)hipFree(x);
    (void)hipFree(o);
    return 0;
}
