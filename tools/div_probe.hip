// Developer probe (GPU box): is the bit-faithful build's explicit division / square root
// (hydro.h: the Newton / Markstein core of the compiler's expansion without the exponent
// scaling and the special-case fix-up) bit-identical to `/` and sqrt()?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DPYRO_FAST=0 -Ipyro2_amd/csrc -o /tmp/div_probe tools/div_probe.hip
//   /tmp/div_probe            -> counts of differing results per operand class
#include "common.h"
#include "hydro.h"
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// random sign / mantissa, exponent uniform in [-erange, erange]
__device__ inline double rnd(uint64_t seed, int erange, bool positive)
{
    const uint64_t h = mix(seed);
    const uint64_t mant = h & 0xFFFFFFFFFFFFFull;
    const int e = (int)((h >> 52) % (uint64_t)(2 * erange + 1)) - erange;
    const uint64_t sign = positive ? 0 : (mix(h) & 1);
    const uint64_t bits = (sign << 63) | ((uint64_t)(1023 + e) << 52) | mant;
    return __longlong_as_double((long long)bits);
}

__global__ void k_probe(int erange, uint64_t seed0, unsigned long long *bad)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nd = 0, ns = 0, nr = 0;
    for (int it = 0; it < 64; it++) {
        const uint64_t s = seed0 + (t * 64 + it) * 3;
        const double a = rnd(s, erange, false), b = rnd(s + 1, erange, false), x = rnd(s + 2, erange, true);
        const double q1 = pyro::pdiv(a, b), q0 = a / b;
        const double s1 = pyro::psqrt(x), s0 = sqrt(x);
        const double r1 = pyro::prcp(b), r0 = 1.0 / b;
        nd += (__double_as_longlong(q1) != __double_as_longlong(q0));
        ns += (__double_as_longlong(s1) != __double_as_longlong(s0));
        nr += (__double_as_longlong(r1) != __double_as_longlong(r0));
    }
    if (nd) atomicAdd(&bad[0], (unsigned long long)nd);
    if (ns) atomicAdd(&bad[1], (unsigned long long)ns);
    if (nr) atomicAdd(&bad[2], (unsigned long long)nr);
}

int main()
{
    unsigned long long *d, h[3];
    hipMalloc((void **)&d, sizeof(h));
    const int ranges[] = {0, 8, 60, 300, 500};
    for (int erange : ranges) {
        hipMemset(d, 0, sizeof(h));
        const int blocks = 1 << 12, threads = 256;          // 2^26 triples per range
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), 0, 0, erange, 0x1234567ull * (erange + 1), d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("exponents in [-%d, %d]: %llu operand sets; differing from IEEE: a/b %llu, sqrt %llu, 1/b %llu\n",
               erange, erange, (unsigned long long)blocks * threads * 64, h[0], h[1], h[2]);
    }
    return 0;
}
