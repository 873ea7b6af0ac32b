// developer probe: what does a phase boundary cost when it is confined to ONE
// XCD -- 32 workgroups (one per CU of XCD 0) meeting at a counter that lives in that XCD's L2,
// nothing written back to memory -- against the chip-wide barrier of gridbar_probe.hip (12 us)
// and a dependent kernel boundary?  Three protocols, each checked word by word:
//   A  agent fences on both sides (release: buffer_wbl2 sc1 -- writes the L2's dirty lines back;
//      acquire: buffer_inv sc1), agent-scope counter                   [the placement-independent form]
//   B  no release fence: s_waitcnt vmcnt(0) (the stores are in the L2), workgroup-scope atomic
//      add (executed in the XCD's L2, line stays there), sc1 poll (bypasses the L1), agent
//      acquire (buffer_inv sc1: this CU's L1 only)
//   C  as B without the acquire: the consumer reads the neighbour's data with sc1 loads
// B and C are only valid while every participant sits on the same XCD (census by XCC_ID below).
// build: hipcc --offload-arch=gfx950 -O3 tools/xcdbar_probe.hip -o tools/bin/xcdbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__device__ __forceinline__ bool xbar(unsigned *bar, unsigned target)
{
    if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (MODE != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ok = __syncthreads_and(ok);
    return ok;
}

// participants: the workgroups on XCD `want`; rank among them by a census counter
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(double *data, unsigned *bar, int nphase, int per, int want,
                                               int npart, long long *cyc, unsigned *census, unsigned *errs)
{
    extern __shared__ double pad[];
    __shared__ int s_rank;
    const int t = threadIdx.x;
    const unsigned x = xcc_id();
    if (t == 0) {
        atomicAdd(&census[x], 1u);
        s_rank = (x == (unsigned)want) ? (int)atomicAdd(&census[8], 1u) : -1;
    }
    __syncthreads();
    const int b = s_rank;
    if (b < 0 || b >= npart) return;
    if (t == 0) pad[0] = 0;
    long long c0 = wall_clock64();
    unsigned bad = 0;
    for (int p = 0; p < nphase; p++) {
        for (int i = t; i < per; i += blockDim.x) data[(size_t)b * per + i] = p * 1000.0 + i + b * 0.001;
        if (!xbar<MODE>(bar, (unsigned)(2 * p + 1) * npart)) { if (t == 0) atomicAdd(&errs[1], 1u); return; }
        const int o = (b + 1) % npart;
        for (int i = t; i < per; i += blockDim.x) {
            double got;
            if (MODE == 2) {
                unsigned long long u = __hip_atomic_load((unsigned long long *)&data[(size_t)o * per + i],
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = __longlong_as_double((long long)u);
            } else got = data[(size_t)o * per + i];
            if (got != p * 1000.0 + i + o * 0.001) bad++;
        }
        if (!xbar<MODE>(bar, (unsigned)(2 * p + 2) * npart)) { if (t == 0) atomicAdd(&errs[1], 1u); return; }
    }
    long long c1 = wall_clock64();
    if (bad) atomicAdd(&errs[0], bad);
    if (t == 0) cyc[b] = c1 - c0;
}

template <int MODE> static void run(const char *name, int nwg, int npart, int per)
{
    double *d; unsigned *bar, *census, *errs; long long *cyc;
    hipMalloc(&d, (size_t)64 * per * 8); hipMalloc(&bar, 256); hipMalloc(&census, 64); hipMalloc(&errs, 8);
    hipMalloc(&cyc, 64 * 8);
    hipMemset(bar, 0, 256); hipMemset(census, 0, 64); hipMemset(errs, 0, 8); hipMemset(cyc, 0, 64 * 8);
    hipMemset(d, 0, (size_t)64 * per * 8);
    int nphase = 200, want = 0;
    // 100 KB of dynamic LDS: one workgroup per CU
    hipFuncSetAttribute((const void *)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nwg), dim3(256), 100 * 1024, 0, d, bar, nphase, per, want, npart, cyc,
                       census, errs);
    hipError_t e = hipDeviceSynchronize();
    long long h[64]; unsigned hc[9], he[2];
    hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, census, 36, hipMemcpyDeviceToHost);
    hipMemcpy(he, errs, 8, hipMemcpyDeviceToHost);
    printf("%-44s %s: %d WGs on XCD 0 (census %u %u %u %u %u %u %u %u), %6d doubles per WG and phase: %6.2f us per "
           "(write, barrier, read neighbour, barrier) = %5.2f us per barrier incl. its share of the copy; "
           "wrong words %u, timeouts %u\n", name, hipGetErrorString(e), npart, hc[0], hc[1], hc[2], hc[3], hc[4], hc[5],
           hc[6], hc[7], per, h[0] / 100.0 / nphase, h[0] / 100.0 / nphase / 2, he[0], he[1]);
    hipFree(d); hipFree(bar); hipFree(census); hipFree(errs); hipFree(cyc);
}

int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int nwg = pr.multiProcessorCount;      // one workgroup per CU: 32 land on every XCD
    for (int per : {256, 1024, 8192}) {
        run<0>("A agent release + acquire, agent counter", nwg, 32, per);
        run<1>("B L2-resident counter, acquire only", nwg, 32, per);
        run<2>("C L2-resident counter, sc1 data loads", nwg, 32, per);
    }
    // the same with 8 and 16 participants (smaller levels need fewer CUs)
    run<1>("B, 16 participants", nwg, 16, 1024);
    run<1>("B, 8 participants", nwg, 8, 1024);
    return 0;
}
