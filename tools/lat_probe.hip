// developer probe: dependent-issue latency of VALU operations on one wavefront
// (cycles per operation in a dependent chain, and with 2/4/8 independent chains)
// build: hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o tools/bin/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int OP>
__global__ void k_lat(double *out, long long *cyc, int n)
{
    double x[CH];
    float y[CH];
    for (int c = 0; c < CH; c++) { x[c] = out[c] + c; y[c] = (float)x[c]; }
    const double a = out[100] + 1.0000001, b = out[101] + 0.5;
    const float af = (float)a, bf = (float)b;
    long long c0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[c]) : "v"(af), "v"(bf));
                if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            }
    }
    long long c1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; c++) s += x[c] + y[c];
    out[threadIdx.x + 200] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
template <int CH, int OP> void run(const char *name, double *d, long long *c, int threads)
{
    long long h;
    const int n = 2000;
    hipLaunchKernelGGL((k_lat<CH, OP>), dim3(1), dim3(threads), 0, 0, d, c, n);
    hipLaunchKernelGGL((k_lat<CH, OP>), dim3(1), dim3(threads), 0, 0, d, c, n);
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-10s chains %d waves/SIMD %d: %.1f cycles per op per chain, %.2f cycles per op issued on the SIMD\n", name,
           CH, threads / 256 ? threads / 256 : 1, (double)h / (n * 16.0), (double)h / (n * 16.0 * CH * (threads >= 256 ? threads / 256 : 1)));
}
int main()
{
    double *d; long long *c;
    hipMalloc(&d, 4096 * 8); hipMalloc(&c, 64); hipMemset(d, 0, 4096 * 8);
    run<1, 0>("fma_f64", d, c, 64); run<2, 0>("fma_f64", d, c, 64); run<4, 0>("fma_f64", d, c, 64);
    run<8, 0>("fma_f64", d, c, 64); run<12, 0>("fma_f64", d, c, 64);
    run<1, 0>("fma_f64", d, c, 1024); run<4, 0>("fma_f64", d, c, 1024);
    run<1, 1>("add_f64", d, c, 64); run<4, 1>("add_f64", d, c, 64);
    run<1, 3>("mul_f64", d, c, 64);
    run<1, 2>("fma_f32", d, c, 64); run<4, 2>("fma_f32", d, c, 64);
    return 0;
}
