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
// Same design as the compressible row-marching kernel (comp_wave.hip): a wavefront
// walks down a strip of rows and keeps everything it needs of the rows above in
// registers (a 5-row window of a, limit2_x and the x interface states, each computed
// once); no LDS, no barrier.  Here every lane owns TWO adjacent columns (2 l, 2 l + 1 of
// the wavefront's 128, of which up to 123 are updated: adv_strip):
//   * the y neighbour of the left column's right side / the right column's left side is
//     in the lane's own registers, so a row needs 6-8 DPP double moves for two cells
//     where the one-column layout needed 10 for one;
//   * 5 apron columns in 128 (3 on the upwind side of v, 2 on the other) instead of 8 in 64;
//   * two independent cells per lane: instruction-level parallelism inside the wavefront
//     where the one-column kernel depended on four wavefronts per SIMD to hide latency
//     (VALU busy was 0.50, profiles/r02m_also_traffic.json).
// Per cell: one limit2 and one limit4 per direction.
//
// Ghost cells.  With `fill` the boundary fill of the variable (outflow,
// reflect-even / -odd, periodic) is folded into the loads: a ghost cell's value
// is fetched from its interior source cell (index remap + sign), exactly the
// value fill_ghost would have stored.  The kernel also writes the ghost frame of
// the NEW buffer (the reference updates in place, so after a step the ghost
// cells hold the values the fill at the start of the step gave them), which
// removes the separate fill_x / fill_y / copy_frame launches: 4 launches -> 1.
//
// Compiled twice (build.py): bit-faithful (-ffp-contract=off, the reference's operation
// order: results identical to NumPy) and contracted (-ffp-contract=fast: north_star's
// tolerance for advection is 1e-12, the multiply-add pairs of the slope / state / flux
// expressions fuse); pyrohip_adv_params.fast_math selects.
#include "common.h"
#include "stencil.h"
#include <type_traits>

#ifndef PYRO_FAST
#define PYRO_FAST 0
#endif
#if PYRO_FAST
#define PYRO_NS fastm
#else
#define PYRO_NS exact
#endif

namespace pyro {
namespace PYRO_NS {

// Column strips.  A wavefront's window is 128 columns; a cell's update reads three columns
// on the upwind side of v and two on the other (the y state of the upwind neighbour carries
// a limited slope: two more columns), so a strip inside the grid updates 128 - 5 = 123 of
// them.  The first strip's window starts at the ghost columns (it carries the ghost frame of
// the new buffer: 4 columns), the last one's must reach the last ghost column.  (Round 2/3
// until here: 4 + 120 + 4 whatever the sign -- 2048 columns were 17 x 120 + 8: eighteen
// strips, the last one for eight columns.)
constexpr int AW_WIN = 128, AW_GHOST = 4;
struct AdvStrip { int A, U0, U1; };           // window start, first / last column updated
__host__ __device__ inline AdvStrip adv_strip(int cb, int ncb, int jlo, int jhi, bool vneg)
{
    const int nl = vneg ? 2 : 3, nr = vneg ? 3 : 2;
    const int first = AW_WIN - AW_GHOST - nr, mid = AW_WIN - nl - nr;
    AdvStrip S;
    if (cb == 0) { S.A = jlo - AW_GHOST; S.U0 = jlo; S.U1 = jlo + first - 1; }
    else { S.U0 = jlo + first + (cb - 1) * mid; S.A = S.U0 - nl; S.U1 = S.U0 + mid - 1; }
    if (cb == ncb - 1) {                      // reach the ghost columns on the right
        const int amin = jhi + AW_GHOST - (AW_WIN - 1);
        if (S.A < amin) S.A = amin;
    }
    if (S.U1 > jhi) S.U1 = jhi;
    return S;
}
// strips of a grid of ny columns (ng = 4 ghost columns)
__host__ __device__ inline int adv_nstrips(int ny, bool vneg)
{
    const int nl = vneg ? 2 : 3, nr = vneg ? 3 : 2;
    const int first = AW_WIN - AW_GHOST - nr, mid = AW_WIN - nl - nr;
    int ncb = ny <= first ? 1 : 1 + (ny - first + mid - 1) / mid;
    // The last strip's window is moved right until it holds the ghost columns (adv_strip).
    // If that costs it its apron on the left -- or, for a single strip, the ghost columns on
    // the left -- one more strip carries the ghost columns alone (nothing to update).
    const AdvStrip S = adv_strip(ncb - 1, ncb, 0, ny - 1, vneg);
    if (ncb == 1 ? S.A != -AW_GHOST : S.U0 - S.A < nl) ncb++;
    return ncb;
}
// rows loaded ahead of their use.  PMC (profiles/r03_adv_pmc.json): with ONE row ahead a
// wavefront sat in s_waitcnt for 33 % (2048^2) / 53 % (8192^2) of its cycles -- an
// iteration is ~1800 cycles, a load under traffic takes longer.  The rows in flight have a
// small ring of their own (its length divides the period of the others, so the loop is still
// unrolled 6 times; a row costs one register move when it enters the stencil window).
// (2, 3 and 6 rows in flight measured at the end of round 3: 27.1 / 27.2 / 29.9 us per 2048^2
// launch under the event timers, 281 / 282 / 287 us at 8192^2 -- the strips do not wait for
// their rows; three wavefronts per SIMD take turns at ~160 instructions per row)
#ifndef PYRO_ADV_PF
#define PYRO_ADV_PF 3
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
    int ncb, L, nunits;     // column strips, rows per strip, strips in all
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

// the lane's two cells of a row: a at column j0 = 2 l (+ strip offset), b at j0 + 1
struct D2 { double a, b; };
// the same quantity one column to the left / right of each of the two cells: one DPP
// double move each, the other neighbour is the lane's own second cell
__device__ __forceinline__ D2 adv_left(const D2 &q) { return D2{adv_m1(q.b), q.a}; }
__device__ __forceinline__ D2 adv_right(const D2 &q) { return D2{q.b, adv_p1(q.a)}; }

// limited slope from shared limit2 values (reconstruction.py:9-120)
#if PYRO_FAST && defined(PYRO_ADV_HALFSLOPE)
// NOT the default (round 6, measured: tools/build_variant.sh halfslope "-DPYRO_ADV_HALFSLOPE" adv_fast):
// the form that pays in the compressible and shallow-water kernels keeps two more values alive per
// cell (a, b), and the three-steps-per-launch instance -- 248-256 registers already -- starts to
// spill (scratch 0 / 12 -> 20 / 68 B per lane): 8192^2 0.195 -> 0.209 ms per step, 2048^2 18.4 -> 18.0 us.
// Contracted build: HALF slopes in signed min / max form -- with lo = min(dl, dr),
// hi = max(dl, dr), a = max(lo, 0), b = min(hi, 0):  limit2 / 2 = max(min((ap - am) / 4, a), b),
// and the same with half the fourth-order centred slope for limit4 (where dl dr > 0 it shares
// their sign: stencil.h mc_select_l4); no sign copy, product, compare or select: 10 + 5 instead
// of 12 + 7 instructions per cell and direction.  The interface states take (1 -+ c) x half slope.
constexpr double ADV_HALF = 1.0;       // factor of the slope in the interface states
__device__ __forceinline__ double adv_limit2(double am, double a0, double ap)
{
    const double dl = ap - a0, dr = a0 - am;
    const double a = fmax(fmin(dl, dr), 0.0), b = fmin(fmax(dl, dr), 0.0);
    return fmax(fmin(0.25 * (ap - am), a), b);
}
template <int LIM>
__device__ __forceinline__ double adv_slope(double l2m, double l20, double l2p, double am1, double a0,
                                            double ap1)
{
    if (LIM == 0) return 0.25 * (ap1 - am1);
    if (LIM == 1) return l20;
    const double dl = ap1 - a0, dr = a0 - am1;
    const double a = fmax(fmin(dl, dr), 0.0), b = fmin(fmax(dl, dr), 0.0);
    return fmax(fmin((1. / 3.) * (ap1 - am1 - 0.5 * (l2p + l2m)), a), b);
}
#else
constexpr double ADV_HALF = 0.5;
__device__ __forceinline__ double adv_limit2(double am, double a0, double ap) { return limit2(am, a0, ap); }
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
#endif

// LIM: limiter (0 none, 1 MC2, 2 MC4); UNEG / VNEG: u < 0 / v < 0 (upwind side)
// wavefronts per workgroup (they do not cooperate: no LDS, no barrier).  Four per workgroup,
// so that the dispatcher hands out four strips at a time, was measured: 24.3 vs 23.2 us at
// 2048^2, 267 vs 261 us at 8192^2 -- one it is.
#ifndef PYRO_ADV_PRIO
#define PYRO_ADV_PRIO 1
#endif
#ifndef PYRO_ADV_WPB
#define PYRO_ADV_WPB 1
#endif
constexpr int ADV_WPB = PYRO_ADV_WPB;

template <int LIM, bool UNEG, bool VNEG>
__global__ __launch_bounds__(64 * ADV_WPB) void k_adv_step(const double *__restrict__ ain,
                                                           double *__restrict__ aout, Geom g, AdvParams P)
{
    const int l = threadIdx.x & 63;
    // (the quotient is computed by vector instructions; without the hint the strip's row
    // range, the loop counter and every row offset derived from them stay in vector registers)
    // workgroup -> strip: workgroups are dealt round-robin to the 8 XCDs (each with its own
    // L2); XCD x takes the strips [x per, (x + 1) per) in order, so the strips that share
    // apron rows and the cache lines at a column cut meet in ONE L2 at about the same time
    const int wg = (int)blockIdx.x * ADV_WPB + (int)(threadIdx.x >> 6);
    const int per = (P.nunits + 7) / 8;
    const int unit = pyro_uniform((wg % 8) * per + wg / 8);
    if (unit >= P.nunits) return;
    const int cb = pyro_uniform(unit % P.ncb), sb = pyro_uniform(unit / P.ncb);
    const int i0 = g.ilo + sb * P.L;                       // strip rows [i0, i1)
    const int i1 = (i0 + P.L < g.ihi + 1) ? i0 + P.L : g.ihi + 1;
    const AdvStrip strip = adv_strip(cb, P.ncb, g.jlo, g.jhi, VNEG);
    const int ja = strip.A + 2 * l;                        // this lane's columns ja, ja + 1
    const int p = g.pitch;
    // row / column maps of the ghost fill
    const BcMap mr = bc_map(g.ilo, g.ihi, g.ng, P.bxl, P.bxr, P.fill != 0);
    const BcMap mc = bc_map(g.jlo, g.jhi, g.ng, P.byl, P.byr, P.fill != 0);
    // per column: in the array, ghost, updated by this wavefront, carried (ghost frame)
    bool jout[2], jown[2], jghost[2], neg_c[2];
    int js[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int j = ja + q;
        const bool jvalid = (j < g.qy);
        jghost[q] = jvalid && (j < g.jlo || j > g.jhi);
        jout[q] = (j >= strip.U0 && j <= strip.U1);
        jown[q] = jvalid && (jout[q] || jghost[q]);        // columns whose ghost cells we carry
        const int jcl = jvalid ? j : g.qy - 1;
        js[q] = bc_src(mc, jcl, g.jlo, g.jhi);
        neg_c[q] = (jcl < g.jlo && mc.odd_lo) || (jcl > g.jhi && mc.odd_hi);
    }
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
    // (two 8-byte accesses per lane and row.  One 16-byte access where the two cells are
    // neighbours in memory -- whole cache lines per instruction -- was measured: the second
    // code path for remapped / ragged lanes costs more than it saves, 277-281 vs 262-269 us
    // at 8192^2)
    auto load_row = [&](int k) {
        const size_t r = (size_t)row_src(k) * p;
        return D2{ain[r + js[0]], ain[r + js[1]]};
    };
    auto store2 = [&](size_t k0, bool s0, bool s1, const D2 &val) {
        if (s0) aout[k0] = val.a;
        if (s1) aout[k0 + 1] = val.b;
    };

    // The rows the march carries from one iteration to the next live in rings that are indexed
    // at compile time: the loop is unrolled over the least common period of the rings, so
    // a value stays in the register it was computed into until it is dead.
    //   rows   a of rows k-4 .. k (the stencil window);  pre  rows k .. k+ADV_PF-1 (loads in flight)
    //   l2x    limit2_x of rows k-3, k-2, k-1
    //   X      x states of rows c-2, c-1, c  (c = k-2);  Y, Ax, Fx  a_y / a_x (as used: at column
    //          j-1 / j+my) and F_x of rows c-1, c
    // (where the loop closes the compiler's s_waitcnt bookkeeping falls back to draining
    // nearly all loads in flight -- vmcnt(2) instead of vmcnt(6) -- once per trip; unrolling
    // over two or three periods was measured and changes nothing, and so does issuing the
    // row loads by inline assembly with a hand-placed s_waitcnt vmcnt(2 ADV_PF): 271-274 vs
    // 262-269 us at 8192^2 -- the waits are not the compiler's)
#ifndef PYRO_ADV_UNR
#define PYRO_ADV_UNR 1
#endif
    constexpr int NR = 6, UNR = PYRO_ADV_UNR * adv_lcm(NR, 6);
    static_assert(UNR % NR == 0 && UNR % 3 == 0 && UNR % 2 == 0 && UNR % ADV_PF == 0 && UNR <= 36,
                  "ring periods");
    const D2 zero{0.0, 0.0};
    D2 rows[NR], l2x[3] = {zero, zero, zero}, Xr[3] = {zero, zero, zero}, Yr[2] = {zero, zero},
       Axr[2] = {zero, zero}, Fxr[2] = {zero, zero};
#pragma unroll
    for (int n = 0; n < NR; n++) rows[n] = zero;
    D2 pre[ADV_PF];
#pragma unroll
    for (int n = 0; n < ADV_PF; n++) pre[n] = load_row(ka + n);
    // per cell functions of the two-cell rows
    auto lim2 = [&](const D2 &m, const D2 &c, const D2 &q) {
        return (LIM != 0) ? D2{adv_limit2(m.a, c.a, q.a), adv_limit2(m.b, c.b, q.b)} : zero;
    };
    auto slope = [&](const D2 &lm, const D2 &l0, const D2 &lp, const D2 &m, const D2 &c, const D2 &q) {
        return D2{adv_slope<LIM>(lm.a, l0.a, lp.a, m.a, c.a, q.a),
                  adv_slope<LIM>(lm.b, l0.b, lp.b, m.b, c.b, q.b)};
    };
    auto step = [&](auto uc, int k) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        // window row n (row k-4+n) and in-flight row n (row k+1+n)
#define ADV_W(n) rows[(U + (n)) % NR]
        {   // row k arrives (through the ghost fill's index map), row k+ADV_PF leaves
            const D2 raw = pre[U % ADV_PF];
            pre[U % ADV_PF] = load_row(k + ADV_PF);
            const bool neg_r = (k < g.ilo && odd_lo) || (k > g.ihi && odd_hi);
            ADV_W(4) = D2{(neg_c[0] != neg_r) ? -raw.a : raw.a, (neg_c[1] != neg_r) ? -raw.b : raw.b};
            // ghost frame of the new buffer
            const bool rghost = (k < g.ilo || k > g.ihi), kin = (k >= i0 && k < i1);
            store2((size_t)k * p + ja, jown[0] && (rghost || (jghost[0] && kin)),
                   jown[1] && (rghost || (jghost[1] && kin)), ADV_W(4));
        }
        const D2 l2b = l2x[U % 3], l2c = l2x[(U + 1) % 3];
        const D2 l2n = lim2(ADV_W(2), ADV_W(3), ADV_W(4));                     // limit2_x of row k-1
        l2x[(U + 2) % 3] = l2n;
        if (k < i0 + 1 || k > i1 + 2) return;
        const D2 Xm1 = Xr[(U + 1) % 3], Fxm1 = Fxr[U % 2];
        // ---- row c = k-2 (window index 2): limited slopes, interface states
        const D2 ac = ADV_W(2);
        const D2 sx = slope(l2b, l2c, l2n, ADV_W(1), ac, ADV_W(3));
        const D2 am = adv_left(ac), ap = adv_right(ac);
        const D2 l2y = lim2(am, ac, ap);
        const D2 l2ym = (LIM == 2) ? adv_left(l2y) : zero, l2yp = (LIM == 2) ? adv_right(l2y) : zero;
        const D2 sy = slope(l2ym, l2y, l2yp, am, ac, ap);
        // upwind states of cell c (interface.py:25-41): its lower face if the
        // velocity is negative, its upper face otherwise
        const D2 X = UNEG ? D2{ac.a - ADV_HALF * (1.0 + cx) * sx.a, ac.b - ADV_HALF * (1.0 + cx) * sx.b}
                          : D2{ac.a + ADV_HALF * (1.0 - cx) * sx.a, ac.b + ADV_HALF * (1.0 - cx) * sx.b};
        const D2 Y = VNEG ? D2{ac.a - ADV_HALF * (1.0 + cy) * sy.a, ac.b - ADV_HALF * (1.0 + cy) * sy.b}
                          : D2{ac.a + ADV_HALF * (1.0 - cy) * sy.a, ac.b + ADV_HALF * (1.0 - cy) * sy.b};
        // a_x on the lower x face of row c; a_y on the lower y faces of rows c, c-1
        const D2 ax_c = UNEG ? X : Xm1;
        // (the lower row's values are the previous iteration's: kept as they were used there,
        // i.e. already shifted by a column where the velocity is positive)
        const D2 ay_c = VNEG ? Y : adv_left(Y), ay_m = Yr[U % 2];
        // F_x[c,j] = u*(a_x[c,j] - dtdy2*(F_yt[c+mx,j+1] - F_yt[c+mx,j]))
        const D2 ayt = (mx == 0) ? ay_c : ay_m;
        const D2 aytp = adv_right(ayt);
        const D2 Fx{u * (ax_c.a - P.dtdy2 * (v * aytp.a - v * ayt.a)),
                    u * (ax_c.b - P.dtdy2 * (v * aytp.b - v * ayt.b))};
        // ---- row g = c-1: F_y and the conservative update
        const D2 axc_s = (my == 0) ? ax_c : adv_left(ax_c);
        const D2 axm_s = Axr[U % 2];                  // a_x of row c-1 at column j + my
        if (k >= i0 + 3) {
            // F_y[g,j] = v*(a_y[g,j] - dtdx2*(F_xt[g+1,j+my] - F_xt[g,j+my]))
            const D2 Fy{v * (ay_m.a - P.dtdx2 * (u * axc_s.a - u * axm_s.a)),
                        v * (ay_m.b - P.dtdx2 * (u * axc_s.b - u * axm_s.b))};
            const D2 Fyh = adv_right(Fy);
            const D2 a1 = ADV_W(1);
            const size_t ko = (size_t)(k - 3) * p + ja;
            store2(ko, jout[0], jout[1],
                   D2{a1.a + P.dtdx * (Fxm1.a - Fx.a) + P.dtdy * (Fy.a - Fyh.a),
                      a1.b + P.dtdx * (Fxm1.b - Fx.b) + P.dtdy * (Fy.b - Fyh.b)});
        }
        Xr[(U + 2) % 3] = X;
        Yr[(U + 1) % 2] = ay_c;
        Axr[(U + 1) % 2] = axc_s;
        Fxr[(U + 1) % 2] = Fx;
#undef ADV_W
    };
#if !defined(PYRO_EMU) && PYRO_ADV_PRIO
    // the wavefronts of a SIMD are served oldest first (comp_wave.hip: the younger ones get
    // what is left and finish alone); they take turns at the priorities instead, by their
    // slot number on the SIMD, one block of the unrolled loop each
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    int turn = (int)(hw_id & 3u);
#endif
    for (int k0 = ka; k0 <= kb; k0 += UNR) {
#if !defined(PYRO_EMU) && PYRO_ADV_PRIO
        switch (turn & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
        turn++;
#endif
        adv_static_for<UNR>([&](auto uc) __attribute__((always_inline)) {
            const int k = k0 + decltype(uc)::value;
            if (k <= kb) step(uc, k);
        });
    }
}

// Rows per strip.  Three wavefronts per SIMD are resident (12 per CU: `slots`); a launch that
// fits them at once must fit them EXACTLY at once -- a handful of wavefronts more than the
// device holds run alone after all the others.  (Round 2's ceil(nx ncb / slots) rows did
// not count the strips it cut: 2048^2 -> 12 rows = 171 x 18 = 3078 wavefronts for 3072 slots;
// 13 rows, 2844 wavefronts: 26.1 -> 21.0 us per step; 3072^2: 47.1 -> 36.4 us, 4096^2:
// 78.4 -> 64.5 us; tools/adv_time.py, profiles/r03_adv_time.txt.)  So: the shortest strip
// of 12 .. 64 rows with which all wavefronts are resident at once (a strip costs L + 6
// iterations for L rows, four of them cheap); grids beyond that run several rounds, where
// 36 .. 52 rows are all within 1 % (8192^2: 265-271 us; 16-20 rows 288, 63-64 rows 280).
static int adv_rows(int nx, int ncb, int cus)
{
    const long slots = 12L * cus;
    if (nx <= 12) return nx;
    for (int L = 12; L <= 64 && L < nx; L++)
        if ((long)ncb * ((nx + L - 1) / L) <= slots) return L;
    return nx < 48 ? nx : 48;
}

template <int LIM>
static void adv_launch(pyrohip_ctx *c, bool uneg, bool vneg, int nwg, const double *cur,
                       double *nxt, const Geom &g, const AdvParams &P)
{
    const dim3 grid((8 * ((nwg + 7) / 8) + ADV_WPB - 1) / ADV_WPB), block(64 * ADV_WPB);
    if (uneg && vneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, true, true>), grid, block, 0, cur, nxt, g, P);
    else if (uneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, true, false>), grid, block, 0, cur, nxt, g, P);
    else if (vneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, false, true>), grid, block, 0, cur, nxt, g, P);
    else
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, false, false>), grid, block, 0, cur, nxt, g, P);
}

// one step of variable n from `cur` into `nxt` (both laid out like a plane of the state)
int adv_step_launch(pyrohip_state *s, int n, const pyrohip_adv_params *ap, double dt, const double *cur,
                    double *nxt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const double u = ap->u, v = ap->v, dx = ap->dx, dy = ap->dy;
    AdvParams P;
    P.u = u; P.v = v; P.dt = dt; P.dx = dx; P.dy = dy; P.limiter = ap->limiter;
    P.cx = u * dt / dx; P.cy = v * dt / dy;
    P.dtdx2 = 0.5 * dt / dx; P.dtdy2 = 0.5 * dt / dy;
    P.dtdx = dt / dx; P.dtdy = dt / dy;
    P.ncb = adv_nstrips(g.ny, v < 0);
    P.L = adv_rows(g.nx, P.ncb, c->num_cus > 0 ? c->num_cus : 256);
    if (ap->march_rows > 0) P.L = ap->march_rows < g.nx ? (ap->march_rows < 4 ? 4 : ap->march_rows) : g.nx;
    P.fill = ap->fill ? 1 : 0;
    P.bxl = s->bc[n * 4 + 0]; P.bxr = s->bc[n * 4 + 1];
    P.byl = s->bc[n * 4 + 2]; P.byr = s->bc[n * 4 + 3];
    const int nwg = P.ncb * ((g.nx + P.L - 1) / P.L);
    P.nunits = nwg;
    const bool uneg = (u < 0), vneg = (v < 0);   // interface.py:28,38
    if (ap->limiter == 0) adv_launch<0>(c, uneg, vneg, nwg, cur, nxt, g, P);
    else if (ap->limiter == 1) adv_launch<1>(c, uneg, vneg, nwg, cur, nxt, g, P);
    else adv_launch<2>(c, uneg, vneg, nwg, cur, nxt, g, P);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

// =========================================================================================
// Several time steps per pass over the grid (periodic grids): k_adv_multi<.., K>.
//
// One step per launch reads and writes the whole grid once per step: at 8192^2 the launch
// runs at the speed of a copy, at 2048^2 the launch's ramp and the write-back of its dirty
// lines at the kernel boundary are a third of the step (profiles/r03_copy_probe.txt,
// DESIGN 3.2).  Here the march is time-skewed like the multigrid marching smoother
// (mg_march.hip): stage s (level s-1 -> level s, s = 1 .. K) consumes the row stage s-1
// produced in the same iteration, t rows behind it (t = 2 for u > 0, 3 for u < 0: the rows a
// cell's update reaches downstream of the march, AdvReach) -- stage 1 takes row k from memory,
// stage s row k - t (s - 1), and the final level's row k - t K is stored.  Intermediate rows
// never leave the registers; memory sees one read and one write of the grid per K steps.
// Every stage is exactly the single step above (same expressions in the same order: the
// bit-faithful build is bit-identical to K launches of k_adv_step with the ghost fill
// folded in), with its own dt quotients (the driver's first steps grow, the last one lands
// on tmax: simulation_null.py:222-244).
//
// Ghost cells.  Between two steps the reference refills the ghost cells
// (pyro_sim.py:250-256).  On a periodic grid the filled array is the periodic image of the
// interior, so a strip simply works in unwrapped ("virtual") row / column numbers and wraps
// them where it loads level 0: a virtual ghost cell of an intermediate level is computed by
// the same operations on the same inputs as the interior cell it is the image of.  The
// aprons grow with K: (3 + 2) K rows per chunk, (3 + 2) K columns per strip.
// The ghost frame of the new buffer holds, as after K in-place steps of the reference, the
// fill of level K - 1: every strip stores the cells of level K - 1 it owns that lie within ng
// of a side to their ghost images as well (the input rows of the last stage).
// Other boundary types need the intermediate levels' ghost values from cells a strip does not
// compute at that moment (the mirror image is upstream of the march): pyrohip_adv_evolve
// takes single steps there.
constexpr int ADV_KMAX = 3;
struct AdvCoef { double cx, cy, dtdx2, dtdy2, dtdx, dtdy; };   // of one step (AdvParams)
struct AdvMultiParams {
    double u, v;
    AdvCoef st[ADV_KMAX];
    int ncb, L, nunits, W;    // column strips, rows per chunk, chunks in all, columns a strip updates
    int prio;                 // rotate wavefront priorities (k_adv_step)
};

// the rows one stage carries from one iteration to the next (k_adv_step's rings)
struct AdvRings { D2 rows[6], l2x[3], Xr[3], Yr[2], Axr[2], Fxr[2]; };

// Rows a cell's update reaches: three on the upwind side of u, two on the other (the x state
// of the upwind neighbour carries a fourth-order slope) -- so a stage needs ADV_LEAD rows of its
// input level above the first row it produces and ADV_TRAIL below the last, and produces row
// k - ADV_TRAIL when row k of its input arrives.
template <bool UNEG> struct AdvReach { static constexpr int lead = UNEG ? 2 : 3, trail = UNEG ? 3 : 2; };

// One iteration of one stage: row k of its input level arrives, row k - trail of its output
// level leaves (returns true and sets `out`) once that row is >= o0; the stage's output is
// needed on the rows [o0, o1).  U: position in the unrolled loop (ring indices).
template <int LIM, bool UNEG, bool VNEG, int U>
__device__ __forceinline__ bool adv_stage(AdvRings &R, const D2 &in, int k, int o0, int o1, double u,
                                          double v, const AdvCoef &C, D2 &out)
{
    constexpr int NR = 6;
    const D2 zero{0.0, 0.0};
#define ADV_W(n) R.rows[(U + (n)) % NR]
    ADV_W(4) = in;
    const D2 l2b = R.l2x[U % 3], l2c = R.l2x[(U + 1) % 3];
    const D2 l2n = (LIM != 0) ? D2{adv_limit2(ADV_W(2).a, ADV_W(3).a, ADV_W(4).a),
                                   adv_limit2(ADV_W(2).b, ADV_W(3).b, ADV_W(4).b)}
                              : zero;                                           // limit2_x of row k-1
    R.l2x[(U + 2) % 3] = l2n;
    // the states of row c = k-2 are needed for c in [o0 - 1, o1 - 1] (u > 0: the face below
    // row c takes the state of row c - 1) or [o0, o1] (u < 0)
    const int c = k - 2;
    if (UNEG ? (c < o0 || c > o1) : (c < o0 - 1 || c > o1 - 1)) return false;
    // ---- row c: limited slopes, interface states (interface.py:25-41)
    const D2 ac = ADV_W(2), a_up = ADV_W(1), a_dn = ADV_W(3);
    const D2 sx{adv_slope<LIM>(l2b.a, l2c.a, l2n.a, a_up.a, ac.a, a_dn.a),
                adv_slope<LIM>(l2b.b, l2c.b, l2n.b, a_up.b, ac.b, a_dn.b)};
    const D2 am = adv_left(ac), ap = adv_right(ac);
    const D2 l2y = (LIM != 0) ? D2{adv_limit2(am.a, ac.a, ap.a), adv_limit2(am.b, ac.b, ap.b)} : zero;
    const D2 l2ym = (LIM == 2) ? adv_left(l2y) : zero, l2yp = (LIM == 2) ? adv_right(l2y) : zero;
    const D2 sy{adv_slope<LIM>(l2ym.a, l2y.a, l2yp.a, am.a, ac.a, ap.a),
                adv_slope<LIM>(l2ym.b, l2y.b, l2yp.b, am.b, ac.b, ap.b)};
    const double cx = C.cx, cy = C.cy;
    // the x state row c gives to a face: its lower face (u < 0) or its upper one, which is the
    // lower face of row c + 1 (u > 0); y likewise
    const D2 X = UNEG ? D2{ac.a - ADV_HALF * (1.0 + cx) * sx.a, ac.b - ADV_HALF * (1.0 + cx) * sx.b}
                      : D2{ac.a + ADV_HALF * (1.0 - cx) * sx.a, ac.b + ADV_HALF * (1.0 - cx) * sx.b};
    const D2 Y = VNEG ? D2{ac.a - ADV_HALF * (1.0 + cy) * sy.a, ac.b - ADV_HALF * (1.0 + cy) * sy.b}
                      : D2{ac.a + ADV_HALF * (1.0 - cy) * sy.a, ac.b + ADV_HALF * (1.0 - cy) * sy.b};
    // a_y on the lower y faces of row c (u, v != 0 here: the upwind offsets of
    // advective_fluxes.py:71-79 are the signs)
    const D2 ay_c = VNEG ? Y : adv_left(Y);
    // F_x on the x face X belongs to -- face c (u < 0) or c + 1 (u > 0); its transverse term
    // takes a_y of the row upwind of the face, which is row c either way:
    // F_x[i,j] = u*(a_x[i,j] - dtdy2*(F_yt[i+mx,j+1] - F_yt[i+mx,j]))
    const D2 aytp = adv_right(ay_c);
    const D2 Fx{u * (X.a - C.dtdy2 * (v * aytp.a - v * ay_c.a)),
                u * (X.b - C.dtdy2 * (v * aytp.b - v * ay_c.b))};
    const D2 Xs = VNEG ? X : adv_left(X);           // a_x of that face at column j + my
    const D2 Xs_m = R.Axr[U % 2], Fx_m = R.Fxr[U % 2];   // the same of the face above it
    bool made = false;
    if (UNEG ? (c >= o0 + 1) : (c >= o0)) {
        // ---- row g = c (u > 0: its faces are those of X's of rows c - 1 and c) or c - 1 (u < 0):
        // F_y[g,j] = v*(a_y[g,j] - dtdx2*(F_xt[g+1,j+my] - F_xt[g,j+my])) and the update
        const D2 ay_g = UNEG ? R.Yr[U % 2] : ay_c;
        const D2 a_g = UNEG ? a_up : ac;
        const D2 Fy{v * (ay_g.a - C.dtdx2 * (u * Xs.a - u * Xs_m.a)),
                    v * (ay_g.b - C.dtdx2 * (u * Xs.b - u * Xs_m.b))};
        const D2 Fyh = adv_right(Fy);
        out = D2{a_g.a + C.dtdx * (Fx_m.a - Fx.a) + C.dtdy * (Fy.a - Fyh.a),
                 a_g.b + C.dtdx * (Fx_m.b - Fx.b) + C.dtdy * (Fy.b - Fyh.b)};
        made = true;
    }
    if (UNEG) R.Yr[(U + 1) % 2] = ay_c;
    R.Axr[(U + 1) % 2] = Xs;
    R.Fxr[(U + 1) % 2] = Fx;
#undef ADV_W
    return made;
}

// wavefronts per SIMD the instances are built for (registers: 84 carried per stage + the
// rows in flight + ~50 temporaries)
#ifndef PYRO_ADVM_NT
#define PYRO_ADVM_NT 0      // non-temporal stores of the final level (developer A/B)
#endif
#ifndef PYRO_ADVM_WPE2
#define PYRO_ADVM_WPE2 2
#endif
#ifndef PYRO_ADVM_WPE3
#define PYRO_ADVM_WPE3 2
#endif
constexpr int advm_wpe(int K) { return K <= 1 ? 3 : (K == 2 ? PYRO_ADVM_WPE2 : PYRO_ADVM_WPE3); }

template <int LIM, bool UNEG, bool VNEG, int K>
__global__ __launch_bounds__(64, advm_wpe(K)) void k_adv_multi(const double *__restrict__ ain,
                                                               double *__restrict__ aout, Geom g,
                                                               AdvMultiParams P)
{
    const int l = threadIdx.x & 63;
    const int wg = (int)blockIdx.x;
    const int per = (P.nunits + 7) / 8;                    // XCD x: the units [x per, (x + 1) per)
    const int unit = pyro_uniform((wg % 8) * per + wg / 8);
    if (unit >= P.nunits) return;
    const int cb = pyro_uniform(unit % P.ncb), sb = pyro_uniform(unit / P.ncb);
    const int i0 = g.ilo + sb * P.L;                       // final rows [i0, i1)
    const int i1 = (i0 + P.L < g.ihi + 1) ? i0 + P.L : g.ihi + 1;
    constexpr int nl = VNEG ? 2 : 3, nr = VNEG ? 3 : 2;
    const int U0 = g.jlo + cb * P.W;                       // final columns [U0, U1]
    const int U1 = (U0 + P.W - 1 < g.jhi) ? U0 + P.W - 1 : g.jhi;
    const int ja = U0 - nl * K + 2 * l;                    // this lane's (virtual) columns ja, ja + 1
    const int p = g.pitch, nx = g.nx, ny = g.ny, ng = g.ng;
    bool jout[2], jgl[2], jgh[2];
    int js[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int j = ja + q;
        int m = (j - g.jlo) % ny;
        if (m < 0) m += ny;
        js[q] = g.jlo + m;                                 // the interior column j is the image of
        jout[q] = (j >= U0 && j <= U1);
        jgl[q] = jout[q] && (j - g.jlo < ng);              // ... with a ghost image at j + ny
        jgh[q] = jout[q] && (g.jhi - j < ng);              // ... at j - ny
    }
    const bool colghost = pyro_uniform((U0 - g.jlo < ng || g.jhi - U1 < ng) ? 1 : 0) != 0;
    constexpr int LEAD = AdvReach<UNEG>::lead, TRAIL = AdvReach<UNEG>::trail;
    const int ka = pyro_uniform(i0 - LEAD * K), kb = pyro_uniform(i1 + TRAIL * K - 1);
    const double u = P.u, v = P.v;
    auto load_row = [&](int k) {
        int ks = k > kb ? kb : k;
        ks = ks < g.ilo ? ks + nx : (ks > g.ihi ? ks - nx : ks);
        const size_t r = (size_t)ks * p;
        return D2{ain[r + js[0]], ain[r + js[1]]};
    };
    // level K - 1 on row r (owned: i0 <= r < i1) to its images in the ghost frame
    auto frame_row = [&](int t, bool own, const D2 &val) {
        const size_t o = (size_t)t * p + ja;
        if (own) {
            if (jout[0]) aout[o] = val.a;
            if (jout[1]) aout[o + 1] = val.b;
        }
        if (colghost) {
            if (jgl[0]) aout[o + ny] = val.a;
            if (jgl[1]) aout[o + 1 + ny] = val.b;
            if (jgh[0]) aout[o - ny] = val.a;
            if (jgh[1]) aout[o + 1 - ny] = val.b;
        }
    };
    constexpr int UNR = 6;
    const D2 zero{0.0, 0.0};
    AdvRings R[K];
#pragma unroll
    for (int s = 0; s < K; s++) {
#pragma unroll
        for (int n = 0; n < 6; n++) R[s].rows[n] = zero;
#pragma unroll
        for (int n = 0; n < 3; n++) { R[s].l2x[n] = zero; R[s].Xr[n] = zero; }
#pragma unroll
        for (int n = 0; n < 2; n++) { R[s].Yr[n] = zero; R[s].Axr[n] = zero; R[s].Fxr[n] = zero; }
    }
    D2 pre[ADV_PF];
#pragma unroll
    for (int n = 0; n < ADV_PF; n++) pre[n] = load_row(ka + n);
    auto step = [&](auto uc, int k) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        D2 cur = pre[U % ADV_PF];
        pre[U % ADV_PF] = load_row(k + ADV_PF);
        adv_static_for<K>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int ks = k - TRAIL * s;
            if constexpr (s == K - 1) {
                if (ks >= i0 && ks < i1) {                 // an owned row of level K - 1
                    const bool rlo = (ks - g.ilo < ng), rhi = (g.ihi - ks < ng);
                    if (colghost) frame_row(ks, false, cur);
                    if (rlo) frame_row(ks + nx, true, cur);
                    if (rhi) frame_row(ks - nx, true, cur);
                }
            }
            D2 nxt = cur;
            const bool made = adv_stage<LIM, UNEG, VNEG, U>(R[s], cur, ks, i0 - LEAD * (K - 1 - s),
                                                            i1 + TRAIL * (K - 1 - s), u, v, P.st[s], nxt);
            if constexpr (s == K - 1) {
                if (made) {
                    const size_t ko = (size_t)(ks - TRAIL) * p + ja;
#if PYRO_ADVM_NT == 1
                    if (jout[0]) __builtin_nontemporal_store(nxt.a, &aout[ko]);
                    if (jout[1]) __builtin_nontemporal_store(nxt.b, &aout[ko + 1]);
#elif PYRO_ADVM_NT == 2     // write-through (sc1): no dirty lines left for the kernel boundary
                    if (jout[0]) __hip_atomic_store(&aout[ko], nxt.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (jout[1]) __hip_atomic_store(&aout[ko + 1], nxt.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                    if (jout[0]) aout[ko] = nxt.a;
                    if (jout[1]) aout[ko + 1] = nxt.b;
#endif
                }
            }
            cur = nxt;
        });
    };
#if !defined(PYRO_EMU) && PYRO_ADV_PRIO
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    int turn = (int)(hw_id & 3u);
#endif
    for (int k0 = ka; k0 <= kb; k0 += UNR) {
#if !defined(PYRO_EMU) && PYRO_ADV_PRIO
        if (P.prio) {
            switch (turn & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
            turn++;
        }
#endif
        adv_static_for<UNR>([&](auto uc) __attribute__((always_inline)) {
            const int k = k0 + decltype(uc)::value;
            if (k <= kb) step(uc, k);
        });
    }
}

// chunk length of a K-step launch: the shortest chunk with which every wavefront is resident
// at once (adv_rows), for the wavefronts per SIMD the instance is built for; beyond that
// several rounds of chunks long enough to make the 6 K apron rows cheap
static int advm_rows(int nx, int ncb, int cus, int K)
{
    const long slots = 4L * advm_wpe(K) * cus;
    const int lmin = 6 * K;
    if (nx <= lmin) return nx;
    for (int L = lmin; L <= 40 * K && L < nx; L++)
        if ((long)ncb * ((nx + L - 1) / L) <= slots) return L;
    // whole rounds: the fewest rounds whose chunks stay below ~48 K rows
    for (int rounds = 1; rounds < 64; rounds++) {
        const long chunks = slots * rounds / ncb;
        if (chunks < 1) continue;
        const int L = (int)((nx + chunks - 1) / chunks);
        if (L <= 48 * K) return L < lmin ? lmin : L;
    }
    return 48 * K;
}

template <int LIM, int K>
static void advm_launch(pyrohip_ctx *c, bool uneg, bool vneg, int nwg, const double *cur, double *nxt,
                        const Geom &g, const AdvMultiParams &P)
{
    const dim3 grid(8 * ((nwg + 7) / 8)), block(64);
    if (uneg && vneg)
        PYRO_LAUNCH(c, "k_adv_multi", (k_adv_multi<LIM, true, true, K>), grid, block, 0, cur, nxt, g, P);
    else if (uneg)
        PYRO_LAUNCH(c, "k_adv_multi", (k_adv_multi<LIM, true, false, K>), grid, block, 0, cur, nxt, g, P);
    else if (vneg)
        PYRO_LAUNCH(c, "k_adv_multi", (k_adv_multi<LIM, false, true, K>), grid, block, 0, cur, nxt, g, P);
    else
        PYRO_LAUNCH(c, "k_adv_multi", (k_adv_multi<LIM, false, false, K>), grid, block, 0, cur, nxt, g, P);
}

// K steps (dts[0 .. K)) of variable n from `cur` into `nxt`, periodic sides, u, v != 0
int adv_multi_launch(pyrohip_state *s, const pyrohip_adv_params *ap, const double *dts, int K,
                     const double *cur, double *nxt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const double u = ap->u, v = ap->v, dx = ap->dx, dy = ap->dy;
    AdvMultiParams P;
    memset(&P, 0, sizeof(P));
    P.u = u; P.v = v;
    for (int k = 0; k < K; k++) {
        const double dt = dts[k];
        AdvCoef &C = P.st[k];
        C.cx = u * dt / dx; C.cy = v * dt / dy;
        C.dtdx2 = 0.5 * dt / dx; C.dtdy2 = 0.5 * dt / dy;
        C.dtdx = dt / dx; C.dtdy = dt / dy;
    }
    P.W = AW_WIN - 5 * K;
    P.ncb = (g.ny + P.W - 1) / P.W;
    P.L = advm_rows(g.nx, P.ncb, c->num_cus > 0 ? c->num_cus : 256, K);
    if (ap->march_rows > 0) P.L = ap->march_rows < g.nx ? (ap->march_rows < 4 ? 4 : ap->march_rows) : g.nx;
    P.prio = (ap->multi_prio >= 0);   // measured: 2048^2 19.0 -> 18.1 us per step, 8192^2 212.5 -> 209.7
    const int nwg = P.ncb * ((g.nx + P.L - 1) / P.L);
    P.nunits = nwg;
    const bool uneg = (u < 0), vneg = (v < 0);
#define ADVM_K(KK)                                                                         \
    do {                                                                                   \
        if (ap->limiter == 0) advm_launch<0, KK>(c, uneg, vneg, nwg, cur, nxt, g, P);       \
        else if (ap->limiter == 1) advm_launch<1, KK>(c, uneg, vneg, nwg, cur, nxt, g, P);  \
        else advm_launch<2, KK>(c, uneg, vneg, nwg, cur, nxt, g, P);                        \
    } while (0)
    if (K == 1) ADVM_K(1);
    else if (K == 2) ADVM_K(2);
    else ADVM_K(3);
#undef ADVM_K
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace PYRO_NS
}  // namespace pyro

#if !PYRO_FAST
// ---- extern "C" entry points (in the bit-faithful unit; the contracted unit only adds
// its kernel instances) ------------------------------------------------------------------
namespace pyro {
namespace fastm {
int adv_step_launch(pyrohip_state *, int, const pyrohip_adv_params *, double, const double *, double *);
int adv_multi_launch(pyrohip_state *, const pyrohip_adv_params *, const double *, int, const double *,
                     double *);
}
}  // namespace pyro

using namespace pyro;

constexpr int AW_GHOST_COLS = 4;   // = pyro::exact::AW_GHOST

static bool simple_bc(int b)
{
    return b == PYROHIP_BC_OUTFLOW || b == PYROHIP_BC_REFLECT_EVEN || b == PYROHIP_BC_REFLECT_ODD ||
           b == PYROHIP_BC_PERIODIC;
}

// after a step of variable n from s->d into s->work: the new level becomes the state's
static int adv_commit(pyrohip_state *s, int n)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    if (s->nvar == 1) {
        // single-variable state: swap the two allocations
        double *old_base = s->base;
        s->base = s->work;
        s->work = old_base;
        s->d = s->base + geom_lead(g);
    } else {
        PYRO_CHECK_HIP(hipMemcpyAsync(s->d + (size_t)n * g.plane, s->work + geom_lead(g),
                                      g.plane * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    s->next_cfl_min = -1.0;
    return 0;
}

static int adv_ensure_work(pyrohip_state *s)
{
    if (s->work_planes < 1) {
        if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
        s->work = nullptr;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->work, (s->g.plane + 16) * sizeof(double)));
        s->work_planes = 1;
    }
    return 0;
}

extern "C" int pyrohip_adv_step_p(pyrohip_state *s, int n, const pyrohip_adv_params *ap, double dt)
{
    PYRO_REQUIRE(s && ap, "NULL argument");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(s->g.ng >= 4, "advection needs ng >= 4 (advection/simulation.py:20)");
    // the kernel carries the ghost frame of the new buffer in windows cut for four ghost
    // columns (adv_strip): with a wider frame the outermost columns would stay unwritten
    PYRO_REQUIRE(!ap->fill || s->g.ng == AW_GHOST_COLS,
                 "the step with the ghost fill folded in is built for ng = 4 (advection/simulation.py:20)");
    PYRO_REQUIRE(ap->limiter >= 0 && ap->limiter <= 2, "limiter must be 0, 1 or 2");
    PYRO_REQUIRE(ap->dx > 0.0 && ap->dy > 0.0, "bad dx / dy");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    if (ap->fill)
        for (int k = 0; k < 4; k++)
            PYRO_REQUIRE(simple_bc(s->bc[n * 4 + k]),
                         "fused ghost fill: outflow / reflect / periodic boundaries only");
    PYRO_TRY(comm_wait_halo(s));
    PYRO_TRY(adv_ensure_work(s));
    double *cur = s->d + (size_t)n * g.plane;
    double *nxt = s->work + geom_lead(g);
    PYRO_TRY(ap->fast_math ? fastm::adv_step_launch(s, n, ap, dt, cur, nxt)
                           : exact::adv_step_launch(s, n, ap, dt, cur, nxt));
    return adv_commit(s, n);
}

// nsteps x (ghost fill of variable n + step) with the time steps dts[0 .. nsteps): what the
// driver's loop (pyro_sim.py:241-281) does to the data when nothing reads it in between.  On
// periodic grids up to p->multi_k (0: the library's choice, at most 3) steps go into one launch
// of k_adv_multi; everything else takes the single-step kernel with the fill folded in.
extern "C" int pyrohip_adv_evolve(pyrohip_state *s, int n, const pyrohip_adv_params *ap,
                                  const double *dts, int nsteps)
{
    PYRO_REQUIRE(s && ap && (dts || nsteps == 0), "NULL argument");
    PYRO_REQUIRE(nsteps >= 0, "negative step count");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(s->g.ng == AW_GHOST_COLS, "pyrohip_adv_evolve is built for ng = 4 (advection/simulation.py:20)");
    PYRO_REQUIRE(ap->limiter >= 0 && ap->limiter <= 2, "limiter must be 0, 1 or 2");
    PYRO_REQUIRE(ap->dx > 0.0 && ap->dy > 0.0, "bad dx / dy");
    const Geom &g = s->g;
    for (int k = 0; k < 4; k++)
        PYRO_REQUIRE(simple_bc(s->bc[n * 4 + k]),
                     "pyrohip_adv_evolve: outflow / reflect / periodic boundaries only");
    bool periodic = true;
    for (int k = 0; k < 4; k++) periodic = periodic && s->bc[n * 4 + k] == PYROHIP_BC_PERIODIC;
    // (measured, profiles/r04_adv_multi_sweep.txt: for u > 0 -- a stage reaches two rows
    // downstream, 248 registers -- three steps per launch 190.5 / 47 / 16.2 us per step at 8192^2 /
    // 4096^2 / 2048^2 against 204.5 / 51 / 16.7 with two; for u < 0 -- three rows, 256 registers and
    // 12 B of scratch -- 206.7 / 17.0 against 202.3 / 16.9: two)
    int kmax = ap->multi_k > 0 ? ap->multi_k : (ap->u > 0.0 ? 3 : 2);
    if (kmax > 3) kmax = 3;
    // the unwrapped indices of a K-step launch wrap once: 3 K rows / the strip's apron columns
    // must fit the grid; u = 0 or v = 0 keep the single step (its upwind offsets are not the signs)
    if (!periodic || ap->u == 0.0 || ap->v == 0.0 || s->nb_set || g.nx < 16 || g.ny < 16) kmax = 1;
    PYRO_TRY(comm_wait_halo(s));
    PYRO_TRY(adv_ensure_work(s));
    pyrohip_adv_params one = *ap;
    one.fill = 1;
    int done = 0;
    while (done < nsteps) {
        const int K = (nsteps - done < kmax) ? nsteps - done : kmax;
        double *cur = s->d + (size_t)n * g.plane;
        double *nxt = s->work + geom_lead(g);
        if (K == 1 && !(ap->multi_k == 1 && periodic && ap->u != 0.0 && ap->v != 0.0 && !s->nb_set)) {
            PYRO_TRY(ap->fast_math ? fastm::adv_step_launch(s, n, &one, dts[done], cur, nxt)
                                   : exact::adv_step_launch(s, n, &one, dts[done], cur, nxt));
        } else {
            PYRO_TRY(ap->fast_math ? fastm::adv_multi_launch(s, &one, dts + done, K, cur, nxt)
                                   : exact::adv_multi_launch(s, &one, dts + done, K, cur, nxt));
        }
        PYRO_TRY(adv_commit(s, n));
        done += K;
    }
    return 0;
}

extern "C" int pyrohip_adv_step_fill(pyrohip_state *s, int n, double dx, double dy, double u,
                                     double v, double dt, int limiter, int fill)
{
    pyrohip_adv_params ap;
    memset(&ap, 0, sizeof(ap));
    ap.dx = dx; ap.dy = dy; ap.u = u; ap.v = v; ap.limiter = limiter; ap.fill = fill;
    return pyrohip_adv_step_p(s, n, &ap, dt);
}

extern "C" int pyrohip_adv_step(pyrohip_state *s, int n, double dx, double dy, double u, double v,
                                double dt, int limiter)
{
    return pyrohip_adv_step_fill(s, n, dx, dy, u, v, dt, limiter, 0);
}
#endif
