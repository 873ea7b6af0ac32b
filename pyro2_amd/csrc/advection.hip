// Linear advection, 2nd-order unsplit CTU update: ONE launch per time step.
//
// Replaces (reference file:line)
//   pyro/advection/simulation.py:56-94        Simulation.evolve
//   pyro/advection/advective_fluxes.py:1-92   unsplit_fluxes
//   pyro/advection/interface.py:4-43          linear_interface
//   pyro/mesh/reconstruction.py:9-120         limit / limit2 / limit4
//   pyro/mesh/array_indexer.py:150-274        fill_ghost (when `fill` is set)
//
// Roofline: HBM bound, 16 B per cell update (read a, write a).
//
// Same design as the compressible row-marching kernel (comp_wave.hip): one
// wavefront = 64 columns (lane = column j, 512-B row loads), the inner 56 are
// updated; it walks down a strip of rows.  x direction (rows): a 5-row window of
// a, limit2_x and the x interface states in registers, each computed once; y
// direction: DPP lane rotation.  No LDS, no barrier.  Per cell: one limit2 and
// one limit4 per direction (the tile kernel of round 1 evaluated limit2 three
// times per face: ~165 VALU per cell, now ~75).
//
// Ghost cells.  With `fill` the boundary fill of the variable (outflow,
// reflect-even / -odd, periodic) is folded into the loads: a ghost cell's value
// is fetched from its interior source cell (index remap + sign), exactly the
// value fill_ghost would have stored.  The kernel also writes the ghost frame of
// the NEW buffer (the reference updates in place, so after a step the ghost
// cells hold the values the fill at the start of the step gave them), which
// removes the separate fill_x / fill_y / copy_frame launches: 4 launches -> 1.
#include "common.h"
#include "stencil.h"
#include <type_traits>

namespace pyro {

constexpr int AW_OUT = 56;        // columns a wavefront updates
// rows loaded ahead of their use.  Measured per 2048^2 step with the unrolled loop: 1 row
// 26.2 us (6 iterations unrolled), 3 rows 27.5 (24), 4 rows 27.2 (18), 7 rows 27.6 (12); 8192^2:
// no difference (four wavefronts per SIMD hide the load of the next row)
#ifndef PYRO_ADV_PF
#define PYRO_ADV_PF 1
#endif
constexpr int ADV_PF = PYRO_ADV_PF;

constexpr int adv_gcd(int a, int b) { return b == 0 ? a : adv_gcd(b, a % b); }
constexpr int adv_lcm(int a, int b) { return a / adv_gcd(a, b) * b; }

template <int N, int U = 0, class F> __device__ __forceinline__ void adv_static_for(F &&fn)
{
    if constexpr (U < N) {
        fn(std::integral_constant<int, U>{});
        adv_static_for<N, U + 1>(fn);
    }
}

struct AdvParams {
    double u, v, dt, dx, dy;
    int limiter;
    // uniform quotients, evaluated once on the host with the reference's
    // expressions (IEEE double on both sides: same bits); a division is ~14
    // VALU instructions per thread otherwise
    double cx, cy;          // u*dt/dx, v*dt/dy          interface.py:10-11
    double dtdx2, dtdy2;    // 0.5*dt/dx, 0.5*dt/dy      advective_fluxes.py:60-61
    double dtdx, dtdy;      // dt/dx, dt/dy              simulation.py:63-64
    int ncb, L;             // column strips, rows per strip
    int fill;               // fold the ghost fill into the loads
    int bxl, bxr, byl, byr; // boundary types of the variable (fill)
};

// lane l-1 / l+1 (rotation: the end lanes are apron, see comp_wave.hip)
#if !defined(PYRO_EMU)
template <int CTRL> __device__ __forceinline__ double adv_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double adv_m1(double v) { return adv_dpp<0x13C>(v); }   // wave_ror:1
__device__ __forceinline__ double adv_p1(double v) { return adv_dpp<0x134>(v); }   // wave_rol:1
#else
__device__ __forceinline__ double adv_m1(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double adv_p1(double v) { return __shfl_down(v, 1, 64); }
#endif

// limited slope from shared limit2 values (reconstruction.py:9-120)
template <int LIM>
__device__ __forceinline__ double adv_slope(double l2m, double l20, double l2p, double am1, double a0,
                                            double ap1)
{
    if (LIM == 0) return 0.5 * (ap1 - am1);
    if (LIM == 1) return l20;
    const double dc = (2. / 3.) * (ap1 - am1 - 0.25 * (l2p + l2m));
    const double dl = ap1 - a0;
    const double dr = a0 - am1;
    return mc_select_l4(dc, dl, dr);
}

// LIM: limiter (0 none, 1 MC2, 2 MC4); UNEG / VNEG: u < 0 / v < 0 (upwind side)
template <int LIM, bool UNEG, bool VNEG>
__global__ __launch_bounds__(64) void k_adv_step(const double *__restrict__ ain,
                                                 double *__restrict__ aout, Geom g, AdvParams P)
{
    const int l = threadIdx.x;
    // (the quotient is computed by vector instructions; without the hint the strip's row
    // range, the loop counter and every row offset derived from them stay in vector registers:
    // 12 quarter-rate v_mul_lo_u32 per row)
    const int cb = pyro_uniform(blockIdx.x % P.ncb), sb = pyro_uniform(blockIdx.x / P.ncb);
    const int i0 = g.ilo + sb * P.L;                       // strip rows [i0, i1)
    const int i1 = (i0 + P.L < g.ihi + 1) ? i0 + P.L : g.ihi + 1;
    const int j = g.jlo + cb * AW_OUT - 4 + l;             // this lane's column
    const bool jvalid = (j < g.qy);
    const bool jghost = jvalid && (j < g.jlo || j > g.jhi);
    const bool jout = (j >= g.jlo && j <= g.jhi && l >= 4 && l <= 59);
    const bool jown = jvalid && ((l >= 4 && l <= 59) || jghost);   // columns whose ghost cells we carry
    const int p = g.pitch;
    // row / column maps of the ghost fill
    const BcMap mr = bc_map(g.ilo, g.ihi, g.ng, P.bxl, P.bxr, P.fill != 0);
    const BcMap mc = bc_map(g.jlo, g.jhi, g.ng, P.byl, P.byr, P.fill != 0);
    const int jcl = jvalid ? j : g.qy - 1;
    const int js = bc_src(mc, jcl, g.jlo, g.jhi);
    const bool neg_c = (jcl < g.jlo && mc.odd_lo) || (jcl > g.jhi && mc.odd_hi);
    // the first / last strip also carries the ghost rows
    const int ka = pyro_uniform((i0 == g.ilo) ? 0 : i0 - 3);
    const int kb = pyro_uniform((i1 == g.ihi + 1) ? g.qx - 1 : i1 + 2);
    const double u = P.u, v = P.v;
    const double cx = P.cx, cy = P.cy;
    const int mx = (u <= 0) ? 0 : -1;   // advective_fluxes.py:71-79
    const int my = (v <= 0) ? 0 : -1;

    // source row of array row k under the ghost fill; its sign goes with the value
    auto row_src = [&](int k) { return bc_src(mr, k > kb ? kb : k, g.ilo, g.ihi); };
    const bool odd_lo = mr.odd_lo, odd_hi = mr.odd_hi;

    // The rows the march carries from one iteration to the next live in rings that are indexed
    // at compile time: the loop is unrolled over the least common period of the rings (18), so
    // a value stays in the register it was computed into until it is dead -- a fifth of the
    // loop's vector instructions were the moves of the sliding windows.
    //   rows   a of rows k-4 .. k (the stencil window) and k+1 .. k+ADV_PF (loads in flight)
    //   l2x    limit2_x of rows k-3, k-2, k-1
    //   X      x states of rows c-2, c-1, c  (c = k-2);  Y, Ax, Fx  a_y / a_x (as used: at column
    //          j-1 / j+my) and F_x of rows c-1, c
    constexpr int NR = 5 + ADV_PF, UNR = adv_lcm(NR, 6);
    static_assert(UNR % NR == 0 && UNR % 3 == 0 && UNR % 2 == 0 && UNR <= 36, "ring periods");
    double rows[NR], l2x[3] = {0, 0, 0}, Xr[3] = {0, 0, 0}, Yr[2] = {0, 0}, Axr[2] = {0, 0}, Fxr[2] = {0, 0};
#pragma unroll
    for (int n = 0; n < NR; n++) rows[n] = 0.0;
    // rows k .. k+ADV_PF-1 in flight: a wavefront consumes a row right after it arrives, so
    // the loads run ahead of the use
#pragma unroll
    for (int n = 0; n < ADV_PF; n++) rows[4 + n] = ain[(size_t)row_src(ka + n) * p + js];
    auto step = [&](auto uc, int k) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        // window row n (row k-4+n) and in-flight row n (row k+1+n)
#define ADV_W(n) rows[(U + (n)) % NR]
        {   // row k arrives (through the ghost fill's index map), row k+ADV_PF leaves
            const double raw = ADV_W(4);
            rows[(U + 4 + ADV_PF) % NR] = ain[(size_t)row_src(k + ADV_PF) * p + js];
            const bool neg = neg_c != ((k < g.ilo && odd_lo) || (k > g.ihi && odd_hi));
            ADV_W(4) = neg ? -raw : raw;
            // ghost frame of the new buffer
            const bool rghost = (k < g.ilo || k > g.ihi);
            if (jown && (rghost || (jghost && k >= i0 && k < i1))) aout[(size_t)k * p + j] = ADV_W(4);
        }
        const double l2b = l2x[U % 3], l2c = l2x[(U + 1) % 3];
        const double l2n = (LIM != 0) ? limit2(ADV_W(2), ADV_W(3), ADV_W(4)) : 0.0;   // limit2_x of row k-1
        l2x[(U + 2) % 3] = l2n;
        if (k < i0 + 1 || k > i1 + 2) return;
        const double Xm1 = Xr[(U + 1) % 3], Fxm1 = Fxr[U % 2];
        // ---- row c = k-2 (window index 2): limited slopes, interface states
        const double sx = adv_slope<LIM>(l2b, l2c, l2n, ADV_W(1), ADV_W(2), ADV_W(3));
        const double am = adv_m1(ADV_W(2)), ap = adv_p1(ADV_W(2));
        const double l2y = (LIM != 0) ? limit2(am, ADV_W(2), ap) : 0.0;
        const double l2ym = (LIM == 2) ? adv_m1(l2y) : 0.0, l2yp = (LIM == 2) ? adv_p1(l2y) : 0.0;
        const double sy = adv_slope<LIM>(l2ym, l2y, l2yp, am, ADV_W(2), ap);
        // upwind states of cell c (interface.py:25-41): its lower face if the
        // velocity is negative, its upper face otherwise
        const double X = UNEG ? ADV_W(2) - 0.5 * (1.0 + cx) * sx : ADV_W(2) + 0.5 * (1.0 - cx) * sx;
        const double Y = VNEG ? ADV_W(2) - 0.5 * (1.0 + cy) * sy : ADV_W(2) + 0.5 * (1.0 - cy) * sy;
        // a_x on the lower x face of row c; a_y on the lower y faces of rows c, c-1
        const double ax_c = UNEG ? X : Xm1;
        // (the lower row's values are the previous iteration's: kept as they were used there,
        // i.e. already shifted by a lane where the velocity is positive -- two DPP moves less each)
        const double ay_c = VNEG ? Y : adv_m1(Y), ay_m = Yr[U % 2];
        // F_x[c,j] = u*(a_x[c,j] - dtdy2*(F_yt[c+mx,j+1] - F_yt[c+mx,j]))
        const double ayt = (mx == 0) ? ay_c : ay_m;
        const double Fx = u * (ax_c - P.dtdy2 * (v * adv_p1(ayt) - v * ayt));
        // ---- row g = c-1: F_y and the conservative update
        const double axc_s = (my == 0) ? ax_c : adv_m1(ax_c);
        const double axm_s = Axr[U % 2];                  // a_x of row c-1 at column j + my
        if (k >= i0 + 3) {
            // F_y[g,j] = v*(a_y[g,j] - dtdx2*(F_xt[g+1,j+my] - F_xt[g,j+my]))
            const double Fy = v * (ay_m - P.dtdx2 * (u * axc_s - u * axm_s));
            const double Fyh = adv_p1(Fy);
            if (jout)
                aout[(size_t)(k - 3) * p + j] = ADV_W(1) + P.dtdx * (Fxm1 - Fx) + P.dtdy * (Fy - Fyh);
        }
        Xr[(U + 2) % 3] = X;
        Yr[(U + 1) % 2] = ay_c;
        Axr[(U + 1) % 2] = axc_s;
        Fxr[(U + 1) % 2] = Fx;
#undef ADV_W
    };
    for (int k0 = ka; k0 <= kb; k0 += UNR)
        adv_static_for<UNR>([&](auto uc) __attribute__((always_inline)) {
            const int k = k0 + decltype(uc)::value;
            if (k <= kb) step(uc, k);
        });
}

// rows per strip.  Measured (tools/adv_time.py): 16-20 rows at 2048^2 (enough
// wavefronts for 4 per SIMD matter more than the 6 apron rows a strip re-reads),
// 48-64 rows from 8192^2 on
static int adv_rows(int nx, int ncb, int cus)
{
    const long slots = 16L * cus;       // 4 wavefronts per SIMD
    int L = (int)(((long)nx * ncb + slots - 1) / slots);
    L = L < 16 ? 16 : (L > 64 ? 64 : L);
    return L < nx ? L : nx;
}

template <int LIM>
static void adv_launch(pyrohip_ctx *c, bool uneg, bool vneg, int nwg, const double *cur,
                       double *nxt, const Geom &g, const AdvParams &P)
{
    const dim3 grid(nwg), block(64);
    if (uneg && vneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, true, true>), grid, block, 0, cur, nxt, g, P);
    else if (uneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, true, false>), grid, block, 0, cur, nxt, g, P);
    else if (vneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, false, true>), grid, block, 0, cur, nxt, g, P);
    else
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, false, false>), grid, block, 0, cur, nxt, g, P);
}

}  // namespace pyro

using namespace pyro;

static bool simple_bc(int b)
{
    return b == PYROHIP_BC_OUTFLOW || b == PYROHIP_BC_REFLECT_EVEN || b == PYROHIP_BC_REFLECT_ODD ||
           b == PYROHIP_BC_PERIODIC;
}

extern "C" int pyrohip_adv_step_fill(pyrohip_state *s, int n, double dx, double dy, double u,
                                     double v, double dt, int limiter, int fill)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(s->g.ng >= 4, "advection needs ng >= 4 (advection/simulation.py:20)");
    PYRO_REQUIRE(limiter >= 0 && limiter <= 2, "limiter must be 0, 1 or 2");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    if (fill)
        for (int k = 0; k < 4; k++)
            PYRO_REQUIRE(simple_bc(s->bc[n * 4 + k]),
                         "fused ghost fill: outflow / reflect / periodic boundaries only");
    // scratch plane for the new time level
    if (s->work_planes < 1) {
        if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
        s->work = nullptr;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->work, (g.plane + 16) * sizeof(double)));
        s->work_planes = 1;
    }
    double *cur = s->d + (size_t)n * g.plane;
    double *nxt = s->work + geom_lead(g);
    AdvParams P;
    P.u = u; P.v = v; P.dt = dt; P.dx = dx; P.dy = dy; P.limiter = limiter;
    P.cx = u * dt / dx; P.cy = v * dt / dy;
    P.dtdx2 = 0.5 * dt / dx; P.dtdy2 = 0.5 * dt / dy;
    P.dtdx = dt / dx; P.dtdy = dt / dy;
    P.ncb = (g.ny + AW_OUT - 1) / AW_OUT;
    P.L = adv_rows(g.nx, P.ncb, c->num_cus > 0 ? c->num_cus : 256);
    if (const char *e = getenv("PYRO_ADV_ROWS")) {   // tuning / test knob
        const int r = atoi(e);
        if (r > 0) P.L = r < g.nx ? (r < 4 ? 4 : r) : g.nx;
    }
    P.fill = fill ? 1 : 0;
    P.bxl = s->bc[n * 4 + 0]; P.bxr = s->bc[n * 4 + 1];
    P.byl = s->bc[n * 4 + 2]; P.byr = s->bc[n * 4 + 3];
    const int nwg = P.ncb * ((g.nx + P.L - 1) / P.L);
    const bool uneg = (u < 0), vneg = (v < 0);   // interface.py:28,38
    if (limiter == 0) adv_launch<0>(c, uneg, vneg, nwg, cur, nxt, g, P);
    else if (limiter == 1) adv_launch<1>(c, uneg, vneg, nwg, cur, nxt, g, P);
    else adv_launch<2>(c, uneg, vneg, nwg, cur, nxt, g, P);
    PYRO_CHECK_HIP(hipGetLastError());
    if (s->nvar == 1) {
        // single-variable state: swap the two allocations
        double *old_base = s->base;
        s->base = s->work;
        s->work = old_base;
        s->d = s->base + geom_lead(g);
    } else {
        PYRO_CHECK_HIP(hipMemcpyAsync(cur, nxt, g.plane * sizeof(double),
                                      hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

extern "C" int pyrohip_adv_step(pyrohip_state *s, int n, double dx, double dy, double u, double v,
                                double dt, int limiter)
{
    return pyrohip_adv_step_fill(s, n, dx, dy, u, v, dt, limiter, 0);
}
