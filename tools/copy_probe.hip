// developer probe: what a plain copy / read / write kernel reaches on this part
// hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_probe tools/copy_probe.hip && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) b[k] = a[k];
}
__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ a, double *__restrict__ out, size_t n)
{
    double s = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) { double2 v = a[k]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(double2 *__restrict__ b, size_t n)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) b[k] = make_double2(1.0, 2.0);
}
int main()
{
    const size_t n = (size_t)8192 * 8192 / 2;      // double2 elements: 537 MB
    double2 *a, *b; double *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, 8);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        float ms[3];
        for (int w = 0; w < 3; w++) {
            for (int r = 0; r < 3; r++) {
                if (w == 0) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
                if (w == 1) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, o, n);
                if (w == 2) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n);
            }
            hipEventRecord(e0);
            for (int r = 0; r < 10; r++) {
                if (w == 0) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
                if (w == 1) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, o, n);
                if (w == 2) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[w], e0, e1); ms[w] /= 10;
        }
        printf("blocks %6d: copy %.1f us = %.2f TB/s (r+w)   read %.1f us = %.2f TB/s   write %.1f us = %.2f TB/s\n", blocks,
               ms[0] * 1e3, 2 * n * 16 / (ms[0] * 1e-3) / 1e12, ms[1] * 1e3, n * 16 / (ms[1] * 1e-3) / 1e12,
               ms[2] * 1e3, n * 16 / (ms[2] * 1e-3) / 1e12);
    }
    return 0;
}
