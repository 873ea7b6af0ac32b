// developer probe: issue cost (cycles per wave-instruction) of the VALU
// instructions the hydro kernels are made of, one wavefront on one SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
#define PROBE(NAME, ASM)                                                        \
    __global__ void NAME(long long *out, double *sink)                          \
    {                                                                           \
        double a = sink[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0;    \
        double e = a + 4.0, f = a + 5.0, g = a + 6.0, h = a + 7.0;              \
        int m = threadIdx.x, n = m + 1;                                         \
        long long t0 = __builtin_readcyclecounter();                            \
        for (int it = 0; it < 16; it++) { REP256(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(m), "+v"(n));) } \
        long long t1 = __builtin_readcyclecounter();                            \
        sink[threadIdx.x] = a + b + c + d + e + f + g + h + m + n;              \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                 \
    }
// independent streams: 4 different destinations round-robin is not expressible in one asm
// string repeated, so each string holds 4 instructions on different registers
PROBE(p_fma64, "v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3")
PROBE(p_fma64_dep, "v_fma_f64 %0, %0, %5, %4\n v_fma_f64 %0, %0, %5, %4\n v_fma_f64 %0, %0, %5, %4\n v_fma_f64 %0, %0, %5, %4")
PROBE(p_add64, "v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3")
PROBE(p_mul64, "v_mul_f64 %0, %4, %0\n v_mul_f64 %1, %4, %1\n v_mul_f64 %2, %4, %2\n v_mul_f64 %3, %4, %3")
PROBE(p_min64, "v_min_f64 %0, %4, %0\n v_min_f64 %1, %4, %1\n v_min_f64 %2, %4, %2\n v_min_f64 %3, %4, %3")
PROBE(p_mov32, "v_mov_b32 %8, %9\n v_mov_b32 %9, %8\n v_mov_b32 %8, %9\n v_mov_b32 %9, %8")
PROBE(p_mov64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7")
PROBE(p_cnd32, "v_cndmask_b32 %8, %8, %9, vcc\n v_cndmask_b32 %9, %9, %8, vcc\n v_cndmask_b32 %8, %8, %9, vcc\n v_cndmask_b32 %9, %9, %8, vcc")
PROBE(p_cnd_sgpr, "v_cndmask_b32_e64 %8, %8, %9, s[20:21]\n v_cndmask_b32_e64 %9, %9, %8, s[20:21]\n v_cndmask_b32_e64 %8, %8, %9, s[20:21]\n v_cndmask_b32_e64 %9, %9, %8, s[20:21]")
PROBE(p_cnd_indep, "v_cndmask_b32 %8, %9, %9, vcc\n v_cndmask_b32 %8, %9, %9, vcc\n v_cndmask_b32 %8, %9, %9, vcc\n v_cndmask_b32 %8, %9, %9, vcc")
PROBE(p_cmpcnd, "v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %8, %8, %9, vcc\n v_cndmask_b32 %9, %9, %8, vcc\n v_fma_f64 %2, %4, %5, %2")
PROBE(p_cmpcnd_s, "v_cmp_lt_f64 s[20:21], %0, %1\n v_cndmask_b32_e64 %8, %8, %9, s[20:21]\n v_cndmask_b32_e64 %9, %9, %8, s[20:21]\n v_fma_f64 %2, %4, %5, %2")
PROBE(p_cnd_fma, "v_cndmask_b32 %8, %8, %9, vcc\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2")
PROBE(p_dpp32, "v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %9 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %8 wave_shl:1 row_mask:0xf bank_mask:0xf")
PROBE(p_cmp64, "v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %4, %5\n v_cmp_lt_f64 vcc, %6, %7")
PROBE(p_rcp64, "v_rcp_f64 %0, %4\n v_rcp_f64 %1, %5\n v_rcp_f64 %2, %6\n v_rcp_f64 %3, %7")
PROBE(p_rsq64, "v_rsq_f64 %0, %4\n v_rsq_f64 %1, %5\n v_rsq_f64 %2, %6\n v_rsq_f64 %3, %7")
PROBE(p_add32, "v_add_u32 %8, %8, %9\n v_add_u32 %9, %9, %8\n v_add_u32 %8, %8, %9\n v_add_u32 %9, %9, %8")
PROBE(p_bfi32, "v_bfi_b32 %8, %8, %9, %9\n v_bfi_b32 %9, %9, %8, %8\n v_bfi_b32 %8, %8, %9, %9\n v_bfi_b32 %9, %9, %8, %8")
PROBE(p_pkmov, "v_pk_mov_b32 %0, %4, %5\n v_pk_mov_b32 %1, %5, %6\n v_pk_mov_b32 %2, %6, %7\n v_pk_mov_b32 %3, %7, %4")
PROBE(p_ldexp64, "v_ldexp_f64 %0, %4, -2\n v_ldexp_f64 %1, %5, -2\n v_ldexp_f64 %2, %6, -2\n v_ldexp_f64 %3, %7, -2")
PROBE(p_max64, "v_max_f64 %0, %4, %0\n v_max_f64 %1, %4, %1\n v_max_f64 %2, %4, %2\n v_max_f64 %3, %4, %3")
PROBE(p_fma64_8, "v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n v_fma_f64 %6, %4, %5, %6\n v_fma_f64 %7, %4, %5, %7\n v_mul_f64 %4, %4, %4\n v_mul_f64 %5, %5, %5")
PROBE(p_mix, "v_fma_f64 %0, %4, %5, %0\n v_mov_b32 %8, %9\n v_fma_f64 %1, %4, %5, %1\n v_mov_b32 %9, %8")
typedef void (*K)(long long *, double *);
int main()
{
    long long *o; double *s;
    (void)hipMalloc(&o, 8); (void)hipMalloc(&s, 1024 * 8); (void)hipMemset(s, 0, 1024 * 8);
    struct { const char *n; K k; } ks[] = {{"v_fma_f64 x4 indep", p_fma64}, {"v_fma_f64 dependent", p_fma64_dep},
        {"v_add_f64", p_add64}, {"v_mul_f64", p_mul64}, {"v_min_f64", p_min64}, {"v_mov_b32", p_mov32},
        {"v_mov_b64", p_mov64}, {"v_cndmask_b32", p_cnd32}, {"cndmask sgpr-pair mask", p_cnd_sgpr}, {"cndmask same-dst indep", p_cnd_indep}, {"cmp,cnd,cnd,fma (vcc)", p_cmpcnd}, {"cmp,cnd,cnd,fma (sgpr)", p_cmpcnd_s}, {"cnd,fma,fma,fma", p_cnd_fma}, {"v_mov_b32_dpp wave_sh", p_dpp32},
        {"v_cmp_lt_f64", p_cmp64}, {"v_rcp_f64", p_rcp64}, {"v_rsq_f64", p_rsq64}, {"v_add_u32", p_add32},
        {"v_bfi_b32", p_bfi32}, {"v_ldexp_f64", p_ldexp64}, {"v_max_f64", p_max64}, {"v_pk_mov_b32", p_pkmov}, {"fma64+mov32 alternating", p_mix}};
    for (auto &e : ks) {
        long long c[3];
        const int nt[3] = {64, 512, 1024};      // 1 wave; 2 waves per SIMD; 4 waves per SIMD
        for (int v = 0; v < 3; v++) {
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL(e.k, dim3(1), dim3(nt[v]), 0, 0, o, s);
            (void)hipMemcpy(&c[v], o, 8, hipMemcpyDeviceToHost);
        }
        printf("%-26s ticks/instr: 1 wave %.3f | 2 waves/SIMD %.3f | 4 waves/SIMD %.3f\n", e.n,
               (double)c[0] / (16.0 * 256 * 4), (double)c[1] / (16.0 * 256 * 4), (double)c[2] / (16.0 * 256 * 4));
    }
    // throughput with the whole chip full: grid 2048 blocks x 256 threads (4 waves/SIMD resident, several rounds)
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double *s2; (void)hipMalloc(&s2, 256 * 8); (void)hipMemset(s2, 0, 256 * 8);
    for (auto &e : ks) {
        hipLaunchKernelGGL(e.k, dim3(2048), dim3(256), 0, 0, o, s2);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL(e.k, dim3(2048), dim3(256), 0, 0, o, s2);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double winstr = 10.0 * 2048 * 4 * 16 * 256 * 4;        // wave-instructions
        printf("%-26s chip-full: %.3f ms, %.2f cycles per wave-instr per SIMD at 2.4 GHz (%.1f T lane-ops/s)\n", e.n, ms,
               ms * 1e-3 * 2.4e9 * 1024 / winstr, winstr * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
