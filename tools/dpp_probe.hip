// developer probe: issue cost of the cross-lane moves the multigrid smoothers use
// build: hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o tools/bin/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int OP>
__global__ void k_dpp(int *out, long long *cyc, int n)
{
    int x[CH], y[CH];
    for (int c = 0; c < CH; c++) { x[c] = out[c] + c + threadIdx.x; y[c] = x[c] * 3; }
    long long c0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) {
                // independent destinations: y[c] <- op(x[c]); issue rate, not latency
                if (OP == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(y[c]) : "v"(x[c]));
                if (OP == 1) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y[c]) : "v"(x[c]));
                if (OP == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(y[c]) : "v"(x[c]));
                if (OP == 3) asm volatile("v_mov_b32_dpp %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "=v"(y[c]) : "v"(x[c]));
                if (OP == 4) asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(y[c]) : "v"(x[c]));
                if (OP == 5) asm volatile("v_add_f64 %0, %1, %1" : "=v"(*(double *)&y[c & ~1]) : "v"(*(double *)&x[c & ~1]));
            }
    }
    long long c1 = clock64();
    int s = 0;
    for (int c = 0; c < CH; c++) s += x[c] + y[c];
    out[threadIdx.x + 200] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
template <int CH, int OP> void run(const char *name, int *d, long long *c, int threads)
{
    long long h;
    const int n = 2000;
    hipLaunchKernelGGL((k_dpp<CH, OP>), dim3(1), dim3(threads), 0, 0, d, c, n);
    hipLaunchKernelGGL((k_dpp<CH, OP>), dim3(1), dim3(threads), 0, 0, d, c, n);
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const int wps = threads >= 256 ? threads / 256 : 1;
    printf("%-12s waves/SIMD %d: %.2f cycles per instruction issued on the SIMD\n", name, wps,
           (double)h / (n * 16.0 * CH * wps));
}
int main()
{
    int *d; long long *c;
    hipMalloc(&d, 4096 * 8); hipMalloc(&c, 64); hipMemset(d, 0, 4096 * 8);
    run<8, 2>("v_mov", d, c, 64); run<8, 2>("v_mov", d, c, 512);
    run<8, 1>("row_shr:1", d, c, 64); run<8, 1>("row_shr:1", d, c, 512);
    run<8, 0>("wave_ror:1", d, c, 64); run<8, 0>("wave_ror:1", d, c, 512);
    run<8, 3>("wave_rol:1", d, c, 64); run<8, 3>("wave_rol:1", d, c, 512);
    run<8, 4>("row_bcast:15", d, c, 64); run<8, 4>("row_bcast:15", d, c, 512);
    run<8, 5>("add_f64", d, c, 64); run<8, 5>("add_f64", d, c, 512);
    return 0;
}
