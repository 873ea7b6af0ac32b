// developer probe: what does a latency-bound single-workgroup kernel see on this GPU?
// shader clock while one CU is busy, dependent fp64 op latency, LDS round trip, barrier.
// build: hipcc --offload-arch=gfx950 -O3 tools/clk_probe.hip -o tools/bin/clk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(double *out, long long *cyc, int n)
{
    __shared__ double lds[2048];
    const int tid = threadIdx.x;
    lds[tid] = tid * 0.5; lds[tid + 1024] = 1.0;
    __syncthreads();
    long long w0 = wall_clock64(), c0 = clock64();
    double x = out[0];
    for (int i = 0; i < n; i++) x = fma(x, 1.0000001, 0.5);            // dependent fp64 chain
    long long c1 = clock64(), w1 = wall_clock64();
    double y = x;
    for (int i = 0; i < n; i++) {                                      // LDS write -> read chain, one wave
        lds[tid] = y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        y = lds[tid ^ 1] + 1.0;
    }
    long long c2 = clock64();
    double z = y;
    for (int i = 0; i < n; i++) {                                      // ... with a workgroup barrier
        lds[tid] = z;
        __syncthreads();
        z = lds[(tid + 64) & 1023] + 1.0;
        __syncthreads();
    }
    long long c3 = clock64(), w3 = wall_clock64();
    if (tid == 0) {
        cyc[0] = c1 - c0; cyc[1] = w1 - w0; cyc[2] = c2 - c1; cyc[3] = c3 - c2; cyc[4] = w3 - w0; cyc[5] = c3 - c0;
    }
    out[tid] = x + y + z;
}
int main()
{
    double *d; long long *c, h[6];
    hipMalloc(&d, 1024 * 8); hipMalloc(&c, 6 * 8); hipMemset(d, 0, 1024 * 8);
    const int n = 20000;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(1024), 0, 0, d, c, n);
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double mhz = (double)h[5] / (double)h[4] * 100.0;        // wall clock: 100 MHz
        printf("shader clock %.0f MHz | dependent fma %.1f cycles | LDS write->read (wave) %.1f cycles | "
               "LDS + 2 workgroup barriers (16 waves) %.1f cycles\n",
               mhz, (double)h[0] / n, (double)h[2] / n, (double)h[3] / n);
    }
    return 0;
}
