// Shallow-water solver: the unsplit CTU scheme of
// pyro/swe on the device, staged through global work planes like kernel_set 0
// of the compressible solver.
//
//   pyro/swe/simulation.py:48-80, 143-193    cons/prim, CFL, evolve
//   pyro/swe/unsplit_fluxes.py:132-380       orchestration, transverse terms
//   pyro/swe/interface.py:5-578              states, riemann_roe, riemann_hllc,
//                                            consFlux
//
// State: 4 planes height, x-momentum, y-momentum, fuel (= h X); primitive
// h, u, v, X.  Compiled with -ffp-contract=off and the reference's operation
// order (4-term in-order dot products included): bit-identical to the oracle.
//
// Compiled twice (build.py): PYRO_FAST=0 -- the above, with every entry point -- and PYRO_FAST=1
// (-ffp-contract=fast, unit swe_fast): the one-launch kernel again with reciprocal / rsq based
// quotients and roots and the characteristic sums written out without their structural zeros
// (sw_trace, sw_roe: the reference's dense 4 x 4 loops multiply by 0 and 1 a hundred times per
// cell), held to 1e-10 element-wise against the bit-faithful build (gpu.fast_math = 1).
#ifndef PYRO_FAST
#define PYRO_FAST 0
#endif
#if PYRO_FAST
#define PYRO_SWNS swf
#else
#define PYRO_SWNS swx
#endif
#include "common.h"
#include "hydro.h"     // pdiv / psqrt: the IEEE quotient / root without the expansion's scaling (bit-identical)
#include "reduce.h"
#include "stencil.h"

namespace pyro {
namespace PYRO_SWNS {

struct SW {   // kernel parameters
    double dx, dy, dt, g;
    int limiter, riemann;   // riemann: 0 Roe, 1 HLLC
};

enum {   // work planes
    SW_Q = 0,                                // 4: h u v X
    SW_XM = 4, SW_XP = 8, SW_YM = 12, SW_YP = 16,   // face states of the cell (conserved)
    SW_FXT = 20, SW_FYT = 24, SW_FX = 28, SW_FY = 32,
    SW_NPL = 36
};

struct V4 { double a[4]; };   // conserved: h, mx, my, hX

__device__ __forceinline__ V4 ld4(const double *__restrict__ p, size_t pl, size_t k)
{
    return V4{{p[k], p[pl + k], p[2 * pl + k], p[3 * pl + k]}};
}
__device__ __forceinline__ void st4(double *__restrict__ p, size_t pl, size_t k, const V4 &v)
{
    p[k] = v.a[0]; p[pl + k] = v.a[1]; p[2 * pl + k] = v.a[2]; p[3 * pl + k] = v.a[3];
}

// u = m / h of the primitive variables (simulation.py:48-63).  The bit-faithful build's quotient
// (hydro.h: reciprocal + Markstein correction) is the IEEE quotient except for the SIGN of a zero:
// -0 / h comes out as +0 -- and the tracing takes copysign(1, u) (interface.py:191-193), so a cell
// whose momentum is exactly -0.0 took the other branch than the reference (found on the GPU with
// the contracted build, whose a * (1 / h) keeps the sign: 2.5e-6 in the cells around one such cell)
__device__ __forceinline__ double sw_vel(double m, double h)
{
    const double q = pdiv(m, h);
    return (!PYRO_FAST && m == 0.0) ? m : q;
}

// simulation.py:65-80
__device__ __forceinline__ V4 sw_prim_to_cons(const double q[4])
{
    V4 U;
    U.a[0] = q[0];
    U.a[1] = q[1] * U.a[0];
    U.a[2] = q[2] * U.a[0];
    U.a[3] = q[3] * q[0];
    return U;
}

// interface.py:557-578; x: idir == 1
__device__ __forceinline__ V4 sw_cons_flux(const V4 &U, double g, bool x)
{
    const double u = pdiv(U.a[1], U.a[0]), v = pdiv(U.a[2], U.a[0]);
    const double w = x ? u : v;
    V4 F;
    F.a[0] = U.a[0] * w;
    F.a[1] = U.a[1] * w;
    F.a[2] = U.a[2] * w;
    const double pr = 0.5 * g * (U.a[0] * U.a[0]);
    if (x) F.a[1] = F.a[1] + pr; else F.a[2] = F.a[2] + pr;
    F.a[3] = U.a[3] * w;
    return F;
}

// characteristic tracing of one cell in one direction, interface.py:5-213:
// primitive states on the cell's lower face (q_r[face]) and upper face
// (q_l[face+1])
#if PYRO_FAST && !defined(SWE_DENSE_TRACE)
// the same sums without the terms that are structurally zero: l_0 . dq = (dq_h / h - dq_n / c) / 2,
// l_2 . dq = -(dq_h / h + dq_n / c) / 2 (sic: the reference's sign), l_1 . dq = dq_t, l_3 . dq = dq_X;
// beta_l of wave 2 and beta_r of wave 0 vanish (e_2 - e_2, e_0 - e_0); r_0 = (h, -c), r_2 = (h, c)
__device__ __forceinline__ void sw_trace(const double q[4], const double dq[4], double g,
                                         double dtdx, bool x, double lo[4], double hi[4])
{
    const int in = x ? 1 : 2, it = x ? 2 : 1;
    double rcs;
    const double cs = psqrt_r(g * q[0], rcs);
    const double dtdx3 = 0.33333 * dtdx;   // sic, interface.py:100
    const double un = q[in];
    const double e0 = un - cs, e2 = un + cs;
    // (1 / h = g / (g h) from the reciprocal root: no second seed)
    const double a = 0.5 * dq[0] * (g * (rcs * rcs)), b = 0.5 * dq[in] * rcs;
    const double as0 = a - b, as2 = -(a + b);
    const double fhi = 0.5 * (1.0 - dtdx * fmax(e2, 0.0)), flo = 0.5 * (1.0 + dtdx * fmin(e0, 0.0));
#pragma unroll
    for (int m = 0; m < 4; m++) { hi[m] = q[m] + fhi * dq[m]; lo[m] = q[m] - flo * dq[m]; }
    // (sign(e) + 1) is 2 for e >= +0, 0 below; (1 - sign(e)) the other way round
    const double pl0 = (e0 >= 0.0 && !(e0 == 0.0 && __builtin_signbit(e0))) ? 2.0 : 0.0;
    const double pl1 = (un >= 0.0 && !(un == 0.0 && __builtin_signbit(un))) ? 2.0 : 0.0;
    const double pl2 = (e2 >= 0.0 && !(e2 == 0.0 && __builtin_signbit(e2))) ? 2.0 : 0.0;
    const double bl0 = dtdx3 * (e2 - e0) * pl0 * as0;            // wave 0 seen from the upper face
    const double bl1 = dtdx3 * cs * pl1;                        // waves 1, 3: (e_2 - u) = c, times dq_t / dq_X
    const double br2 = dtdx3 * (e0 - e2) * (2.0 - pl2) * as2;    // wave 2 seen from the lower face
    const double br1 = -dtdx3 * cs * (2.0 - pl1);
    hi[0] += bl0 * q[0]; hi[in] -= bl0 * cs; hi[it] += bl1 * dq[it]; hi[3] += bl1 * dq[3];
    lo[0] += br2 * q[0]; lo[in] += br2 * cs; lo[it] += br1 * dq[it]; lo[3] += br1 * dq[3];
}
// the same from HALF slopes hq = dq / 2 (round 6, the one-launch kernel: the limited slopes in the
// signed min / max form of half_limit2 below): the factors 1/2 of the projections and of the
// reference states disappear into hq, the 2 of (sign + 1) into the wave factors
__device__ __forceinline__ void sw_trace_h(const double q[4], const double hq[4], double g,
                                           double dtdx, bool x, double lo[4], double hi[4])
{
    const int in = x ? 1 : 2, it = x ? 2 : 1;
    double rcs;
    const double cs = psqrt_r(g * q[0], rcs);
    const double dtdx3 = 0.33333 * dtdx;   // sic, interface.py:100
    const double un = q[in];
    const double e0 = un - cs, e2 = un + cs;
    const double a = hq[0] * (g * (rcs * rcs)), b = hq[in] * rcs;
    const double as0 = a - b, as2 = -(a + b);
    const double fhi = 1.0 - dtdx * fmax(e2, 0.0), flo = 1.0 + dtdx * fmin(e0, 0.0);
#pragma unroll
    for (int m = 0; m < 4; m++) { hi[m] = q[m] + fhi * hq[m]; lo[m] = q[m] - flo * hq[m]; }
    const double pl0 = (e0 >= 0.0 && !(e0 == 0.0 && __builtin_signbit(e0))) ? 2.0 : 0.0;
    const double pl1 = (un >= 0.0 && !(un == 0.0 && __builtin_signbit(un))) ? 4.0 : 0.0;
    const double pl2 = (e2 >= 0.0 && !(e2 == 0.0 && __builtin_signbit(e2))) ? 2.0 : 0.0;
    const double bl0 = dtdx3 * (e2 - e0) * pl0 * as0;
    const double bl1 = dtdx3 * cs * pl1;
    const double br2 = dtdx3 * (e0 - e2) * (2.0 - pl2) * as2;
    const double br1 = -dtdx3 * cs * (4.0 - pl1);
    hi[0] += bl0 * q[0]; hi[in] -= bl0 * cs; hi[it] += bl1 * hq[it]; hi[3] += bl1 * hq[3];
    lo[0] += br2 * q[0]; lo[in] += br2 * cs; lo[it] += br1 * hq[it]; lo[3] += br1 * hq[3];
}
// MC-limited slopes as half slopes (the compressible kernel's: fused_common.h half_limit2 /
// half_slope_shared -- signed min / max, no sign copy, product, compare or select)
__device__ __forceinline__ void sw_half_clip(double dl, double dr, double &a, double &b)
{
    a = fmax(fmin(dl, dr), 0.0);
    b = fmin(fmax(dl, dr), 0.0);
}
__device__ __forceinline__ double sw_half_limit2(double am, double a0, double ap)
{
    double a, b;
    sw_half_clip(ap - a0, a0 - am, a, b);
    return fmax(fmin(0.25 * (ap - am), a), b);
}
__device__ __forceinline__ double sw_half_limit4_from(double h2m, double h2p, double am1, double a0, double ap1)
{
    double a, b;
    sw_half_clip(ap1 - a0, a0 - am1, a, b);
    return fmax(fmin((1. / 3.) * (ap1 - am1 - 0.5 * (h2p + h2m)), a), b);
}
#else
__device__ __forceinline__ void sw_trace(const double q[4], const double dq[4], double g,
                                         double dtdx, bool x, double lo[4], double hi[4])
{
    const int in = x ? 1 : 2, it = x ? 2 : 1;
    const double cs = psqrt(g * q[0]);
    const double dtdx3 = 0.33333 * dtdx;   // sic, interface.py:100
    double lvec[4][4] = {}, rvec[4][4] = {}, e_val[4], betal[4], betar[4];
    e_val[0] = q[in] - cs; e_val[1] = q[in]; e_val[2] = q[in] + cs; e_val[3] = q[in];
    lvec[0][0] = cs;   lvec[0][in] = -q[0];
    lvec[1][it] = 1.0;
    lvec[2][0] = cs;   lvec[2][in] = q[0];
    rvec[0][0] = q[0]; rvec[0][in] = -cs;
    rvec[1][it] = 1.0;
    rvec[2][0] = q[0]; rvec[2][in] = cs;
    lvec[3][3] = 1.0; rvec[3][3] = 1.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lvec[0][k] = pdiv(lvec[0][k] * 0.50, cs * q[0]);
        lvec[2][k] = pdiv(-lvec[2][k] * 0.50, cs * q[0]);
    }
    double factor = 0.5 * (1.0 - dtdx * fmax(e_val[2], 0.0));
#pragma unroll
    for (int m = 0; m < 4; m++) hi[m] = q[m] + factor * dq[m];
    factor = 0.5 * (1.0 + dtdx * fmin(e_val[0], 0.0));
#pragma unroll
    for (int m = 0; m < 4; m++) lo[m] = q[m] - factor * dq[m];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        double asum = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) asum += lvec[m][k] * dq[k];
        betal[m] = dtdx3 * (e_val[2] - e_val[m]) * (copysign(1.0, e_val[m]) + 1.0) * asum;
        betar[m] = dtdx3 * (e_val[0] - e_val[m]) * (1.0 - copysign(1.0, e_val[m])) * asum;
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        double sum_l = 0.0, sum_r = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sum_l += betal[k] * rvec[k][m];
            sum_r += betar[k] * rvec[k][m];
        }
        hi[m] = hi[m] + sum_l;
        lo[m] = lo[m] + sum_r;
    }
}
#endif

// interface.py:216-385
#if PYRO_FAST && !defined(SWE_DENSE_ROE)
// The contracted build's Roe solver: the reference's formulas with every quotient and root that
// shares an operand taken ONCE (round 6; the straightforward form below has ~25 reciprocal /
// root evaluations per face -- Ul_n / sqrt(h_l), Ur_n / sqrt(h_r), ... / (sqrt(h_l) + sqrt(h_r))
// for every component, U_n / h twice over in delta and in the two consFlux calls, three roots of
// g h -- each a seed + Newton steps: a third of the kernel's instructions, profiles/
// r06_isa_hist_sw_wave.txt).  Here: 1 / sqrt(h_l), 1 / sqrt(h_r) (rsq: the roots and, squared,
// 1 / h come with them), 1 / (sqrt(h_l) + sqrt(h_r)), and sqrt + reciprocal of the Roe sound speed
// -- four seeds per face; c = sqrt(g) sqrt(h); c* = |h*-root| (sqrt(g (1 / g) x^2)); the
// transcritical fix's quotients only where a wave speed is near zero.  Algebraically the
// reference's expressions; held to 1e-10 element-wise against the bit-faithful build
// (tests/test_swe.py, tests/test_fastseed_emu.py, tests/test_fullsize_legs.py).
__device__ __forceinline__ V4 sw_roe(const V4 &Ul, const V4 &Ur, double g, bool x, double sg = -1.0)
{   // sg: sqrt(g) if the caller has it (the marching kernel takes the root once per launch)
    const double smallc = 1.e-10, tol = 0.1e-1;
    const int im = x ? 1 : 2, it = x ? 2 : 1;
    const double h_l = Ul.a[0], h_r = Ur.a[0];
    double isl, isr;                                   // 1 / sqrt(h)
    const double sl = psqrt_r(h_l, isl), sr = psqrt_r(h_r, isr);
    const double r_l = isl * isl, r_r = isr * isr;     // 1 / h
    if (sg < 0.0) sg = psqrt(g);
    const double un_l = Ul.a[im] * r_l, un_r = Ur.a[im] * r_r;
    const double c_l = fmax(smallc, sg * sl), c_r = fmax(smallc, sg * sr);
    const double rs = prcp(sl + sr);
    const double h_roe = sl * sr;                      // sqrt(h_l h_r)
    const double un_roe = (Ul.a[im] * isl + Ur.a[im] * isr) * rs;
    const double ut_roe = (Ul.a[it] * isl + Ur.a[it] * isr) * rs;
    const double d0 = h_r - h_l;
    const double dn = un_r - un_l, dt_ = Ur.a[it] * r_r - Ul.a[it] * r_l, dX = Ur.a[3] * r_r - Ul.a[3] * r_l;
    double rc_roe;
    const double c_roe = psqrt_r(0.5 * (c_l * c_l + c_r * c_r), rc_roe);
    double lam0 = un_roe - c_roe, lam2 = un_roe + c_roe;
    const double hc = h_roe * rc_roe * dn;
    // 0.5 (F(Ul) + F(Ur)) (consFlux, interface.py:557-578)
    V4 F;
    F.a[0] = 0.5 * (h_l * un_l + h_r * un_r);
    F.a[im] = 0.5 * ((Ul.a[im] * un_l + 0.5 * g * (h_l * h_l)) + (Ur.a[im] * un_r + 0.5 * g * (h_r * h_r)));
    F.a[it] = 0.5 * (Ul.a[it] * un_l + Ur.a[it] * un_r);
    F.a[3] = 0.5 * (Ul.a[3] * un_l + Ur.a[3] * un_r);
    if (fabs(lam0) < tol || fabs(lam2) < tol) {        // Harten-Hyman fix of a transcritical wave
        const double hs = 0.5 * (c_l + c_r) + 0.25 * (un_l - un_r);
        const double u_star = 0.5 * (un_l + un_r) + c_l - c_r;
        const double c_star = psqrt(g * (pdiv(1.0, g) * (hs * hs)));
        if (fabs(lam0) < tol)
            lam0 = pdiv(lam0 * (u_star - c_star - lam0), u_star - c_star - (un_l - c_l));
        if (fabs(lam2) < tol)
            lam2 = pdiv(lam2 * (u_star + c_star - lam2), u_star + c_star - (un_r + c_r));
    }
    // K_0 = (1, u - c, u_t), K_1 = e_t, K_2 = (1, u + c, u_t), K_3 = e_X: the non-zero terms only
    // w_m = 0.5 alpha_m |lambda_m| with alpha_0,2 = 0.5 (dh -+ h c^-1 du_n), alpha_1 = h du_t, alpha_3 = h dX
    const double hh = 0.5 * h_roe * fabs(un_roe);
    const double w0 = (0.25 * fabs(lam0)) * (d0 - hc), w1 = hh * dt_;
    const double w2 = (0.25 * fabs(lam2)) * (d0 + hc), w3 = hh * dX;
    F.a[0] -= w0 + w2;
    F.a[im] -= w0 * (un_roe - c_roe) + w2 * (un_roe + c_roe);
    F.a[it] -= (w0 + w2) * ut_roe + w1;
    F.a[3] -= w3;
    return F;
}
#else
__device__ __forceinline__ V4 sw_roe(const V4 &Ul, const V4 &Ur, double g, bool x)
{
    const double smallc = 1.e-10, tol = 0.1e-1;
    const int im = x ? 1 : 2, it = x ? 2 : 1;
    const double h_l = Ul.a[0], un_l = pdiv(Ul.a[im], h_l);
    const double h_r = Ur.a[0], un_r = pdiv(Ur.a[im], h_r);
    const double c_l = fmax(smallc, psqrt(g * h_l)), c_r = fmax(smallc, psqrt(g * h_r));
    double U_roe[4], delta[4], lambda[4], alpha[4], K[4][4] = {};
#pragma unroll
    for (int n = 0; n < 4; n++) {
        U_roe[n] = pdiv(pdiv(Ul.a[n], psqrt(h_l)) + pdiv(Ur.a[n], psqrt(h_r)), psqrt(h_l) + psqrt(h_r));
        delta[n] = pdiv(Ur.a[n], h_r) - pdiv(Ul.a[n], h_l);
    }
    U_roe[0] = psqrt(h_l * h_r);
    const double c_roe = psqrt(0.5 * (c_l * c_l + c_r * c_r));
    delta[0] = h_r - h_l;
    const double un_roe = U_roe[im];
    lambda[0] = un_roe - c_roe; lambda[1] = un_roe; lambda[2] = un_roe + c_roe; lambda[3] = un_roe;
    alpha[0] = 0.5 * (delta[0] - pdiv(U_roe[0], c_roe) * delta[im]);
    alpha[1] = U_roe[0] * delta[it];
    alpha[2] = 0.5 * (delta[0] + pdiv(U_roe[0], c_roe) * delta[im]);
    alpha[3] = U_roe[0] * delta[3];
    K[0][0] = 1.0; K[0][im] = un_roe - c_roe; K[0][it] = U_roe[it];
    K[1][it] = 1.0;
    K[2][0] = 1.0; K[2][im] = un_roe + c_roe; K[2][it] = U_roe[it];
    K[3][3] = 1.0;
    const V4 Fl = sw_cons_flux(Ul, g, x), Fr = sw_cons_flux(Ur, g, x);
    V4 F;
#pragma unroll
    for (int n = 0; n < 4; n++) F.a[n] = 0.5 * (Fl.a[n] + Fr.a[n]);
    const double hs = 0.5 * (c_l + c_r) + 0.25 * (un_l - un_r);
    const double h_star = pdiv(1.0, g) * (hs * hs);
    const double u_star = 0.5 * (un_l + un_r) + c_l - c_r;
    const double c_star = psqrt(g * h_star);
    if (fabs(lambda[0]) < tol)
        lambda[0] = pdiv(lambda[0] * (u_star - c_star - lambda[0]), u_star - c_star - (un_l - c_l));
    if (fabs(lambda[2]) < tol)
        lambda[2] = pdiv(lambda[2] * (u_star + c_star - lambda[2]), u_star + c_star - (un_r + c_r));
#if PYRO_FAST && !defined(SWE_DENSE_ROE)
    {   // K_0 = (1, u - c, u_t), K_1 = e_t, K_2 = (1, u + c, u_t), K_3 = e_X: the non-zero terms only
        const double w0 = 0.5 * alpha[0] * fabs(lambda[0]), w1 = 0.5 * alpha[1] * fabs(lambda[1]);
        const double w2 = 0.5 * alpha[2] * fabs(lambda[2]), w3 = 0.5 * alpha[3] * fabs(lambda[3]);
        F.a[0] -= w0 + w2;
        F.a[im] -= w0 * (un_roe - c_roe) + w2 * (un_roe + c_roe);
        F.a[it] -= (w0 + w2) * U_roe[it] + w1;
        F.a[3] -= w3;
    }
#else
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int m = 0; m < 4; m++) F.a[n] -= 0.5 * alpha[m] * fabs(lambda[m]) * K[m][n];
#endif
    return F;
}
#endif

// interface.py:388-554
__device__ __forceinline__ V4 sw_hllc(const V4 &Ul, const V4 &Ur, double g, bool x)
{
    const double smallc = 1.e-10;
    const int im = x ? 1 : 2, it = x ? 2 : 1;
    const double h_l = Ul.a[0], un_l = pdiv(Ul.a[im], h_l), ut_l = pdiv(Ul.a[it], h_l);
    const double h_r = Ur.a[0], un_r = pdiv(Ur.a[im], h_r), ut_r = pdiv(Ur.a[it], h_r);
    const double c_l = fmax(smallc, psqrt(g * h_l)), c_r = fmax(smallc, psqrt(g * h_r));
    const double h_avg = 0.5 * (h_l + h_r), c_avg = 0.5 * (c_l + c_r);
    const double hstar = h_avg - pdiv(0.25 * (un_r - un_l) * h_avg, c_avg);
    const double S_l = (hstar <= h_l) ? un_l - c_l
                                      : un_l - pdiv(c_l * psqrt(0.5 * (hstar + h_l) * hstar), h_l);
    const double S_r = (hstar <= h_r) ? un_r + c_r
                                      : un_r + pdiv(c_r * psqrt(0.5 * (hstar + h_r) * hstar), h_r);
    const double S_c = pdiv(S_l * h_r * (un_r - S_r) - S_r * h_l * (un_l - S_l),
                            h_r * (un_r - S_r) - h_l * (un_l - S_l));
    V4 Us, F;
    if (S_r <= 0.0) return sw_cons_flux(Ur, g, x);
    if (S_c <= 0.0 && 0.0 < S_r) {
        const double fac = pdiv(h_r * (S_r - un_r), S_r - S_c);
        Us.a[0] = fac; Us.a[im] = fac * S_c; Us.a[it] = fac * ut_r;
        Us.a[3] = pdiv(fac * Ur.a[3], h_r);
        F = sw_cons_flux(Ur, g, x);
#pragma unroll
        for (int n = 0; n < 4; n++) F.a[n] = F.a[n] + S_r * (Us.a[n] - Ur.a[n]);
        return F;
    }
    if (S_l < 0.0 && 0.0 < S_c) {
        const double fac = pdiv(h_l * (S_l - un_l), S_l - S_c);
        Us.a[0] = fac; Us.a[im] = fac * S_c; Us.a[it] = fac * ut_l;
        Us.a[3] = pdiv(fac * Ul.a[3], h_l);
        F = sw_cons_flux(Ul, g, x);
#pragma unroll
        for (int n = 0; n < 4; n++) F.a[n] = F.a[n] + S_l * (Us.a[n] - Ul.a[n]);
        return F;
    }
    return sw_cons_flux(Ul, g, x);
}

__device__ __forceinline__ V4 sw_riemann(const V4 &Ul, const V4 &Ur, const SW &P, bool x)
{
    return P.riemann == 1 ? sw_hllc(Ul, Ur, P.g, x) : sw_roe(Ul, Ur, P.g, x);
}

#if !PYRO_FAST     // the staged kernels (every stage dumpable) exist in the bit-faithful unit only
// ---- stage 0: primitives over the whole array (simulation.py:48-63) ------
__global__ __launch_bounds__(256) void k_sw_prim(const double *__restrict__ U,
                                                 double *__restrict__ W, Geom g)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= g.qy) return;
    const size_t k = (size_t)i * g.pitch + j, pl = g.plane;
    const V4 Uc = ld4(U, pl, k);
    double *Q = W + (size_t)SW_Q * pl;
    Q[k] = Uc.a[0];
    Q[pl + k] = sw_vel(Uc.a[1], Uc.a[0]);
    Q[2 * pl + k] = sw_vel(Uc.a[2], Uc.a[0]);
    Q[3 * pl + k] = sw_vel(Uc.a[3], Uc.a[0]);
}

// ---- stage 1: limited slopes + tracing for the cells of R(1) --------------
__global__ __launch_bounds__(256) void k_sw_states(double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo - 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + blockIdx.y;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    const double *Q = W + (size_t)SW_Q * pl;
    double q[4], dqx[4], dqy[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *a = Q + (size_t)n * pl;
        q[n] = a[k];
        // xi = 1.0 (unsplit_fluxes.py:175-177; no flattening for swe)
        dqx[n] = 1.0 * limited_slope(a[k - 2 * p], a[k - p], a[k], a[k + p], a[k + 2 * p],
                                     P.limiter);
        dqy[n] = 1.0 * limited_slope(a[k - 2], a[k - 1], a[k], a[k + 1], a[k + 2], P.limiter);
    }
    double lo[4], hi[4];
    sw_trace(q, dqx, P.g, P.dt / P.dx, true, lo, hi);
    st4(W + (size_t)SW_XM * pl, pl, k, sw_prim_to_cons(lo));
    st4(W + (size_t)SW_XP * pl, pl, k, sw_prim_to_cons(hi));
    sw_trace(q, dqy, P.g, P.dt / P.dy, false, lo, hi);
    st4(W + (size_t)SW_YM * pl, pl, k, sw_prim_to_cons(lo));
    st4(W + (size_t)SW_YP * pl, pl, k, sw_prim_to_cons(hi));
}

// ---- stage 2: transverse Riemann problems on the lower faces of R(1) ------
__global__ __launch_bounds__(256) void k_sw_riemann_t(double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo - 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + blockIdx.y;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    if (i >= g.ilo)
        st4(W + (size_t)SW_FXT * pl, pl, k,
            sw_riemann(ld4(W + (size_t)SW_XP * pl, pl, k - p), ld4(W + (size_t)SW_XM * pl, pl, k),
                       P, true));
    if (j >= g.jlo)
        st4(W + (size_t)SW_FYT * pl, pl, k,
            sw_riemann(ld4(W + (size_t)SW_YP * pl, pl, k - 1), ld4(W + (size_t)SW_YM * pl, pl, k),
                       P, false));
}

#endif   // !PYRO_FAST
__device__ __forceinline__ V4 sw_corrected(const V4 &U, const V4 &Fhi, const V4 &Flo, double c)
{
    // U += -0.5*dtdy*(F_hi - F_lo), unsplit_fluxes.py:336-352 (c = 0.5*dt/d)
    V4 r;
#pragma unroll
    for (int n = 0; n < 4; n++) r.a[n] = U.a[n] + (-c * (Fhi.a[n] - Flo.a[n]));
    return r;
}
#if !PYRO_FAST

// ---- stage 3: transverse correction + final Riemann problems --------------
// thread (i,j) in [ilo, ihi+1] x [jlo, jhi+1]
__global__ __launch_bounds__(256) void k_sw_final(double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    const double *FXT = W + (size_t)SW_FXT * pl, *FYT = W + (size_t)SW_FYT * pl;
    const double hdtdx = 0.5 * (P.dt / P.dx), hdtdy = 0.5 * (P.dt / P.dy);
    if (j <= g.jhi) {
        const V4 Uxl = sw_corrected(ld4(W + (size_t)SW_XP * pl, pl, k - p), ld4(FYT, pl, k - p + 1),
                                    ld4(FYT, pl, k - p), hdtdy);
        const V4 Uxr = sw_corrected(ld4(W + (size_t)SW_XM * pl, pl, k), ld4(FYT, pl, k + 1),
                                    ld4(FYT, pl, k), hdtdy);
        st4(W + (size_t)SW_FX * pl, pl, k, sw_riemann(Uxl, Uxr, P, true));
    }
    if (i <= g.ihi) {
        const V4 Uyl = sw_corrected(ld4(W + (size_t)SW_YP * pl, pl, k - 1), ld4(FXT, pl, k + p - 1),
                                    ld4(FXT, pl, k - 1), hdtdx);
        const V4 Uyr = sw_corrected(ld4(W + (size_t)SW_YM * pl, pl, k), ld4(FXT, pl, k + p),
                                    ld4(FXT, pl, k), hdtdx);
        st4(W + (size_t)SW_FY * pl, pl, k, sw_riemann(Uyl, Uyr, P, false));
    }
}

// ---- stage 4: conservative update (simulation.py:172-181) -----------------
__global__ __launch_bounds__(256) void k_sw_update(double *__restrict__ U,
                                                   const double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    const double dtdx = P.dt / P.dx, dtdy = P.dt / P.dy;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *fx = W + (size_t)(SW_FX + n) * pl, *fy = W + (size_t)(SW_FY + n) * pl;
        U[(size_t)n * pl + k] += dtdx * (fx[k] - fx[k + p]) + dtdy * (fy[k] - fy[k + 1]);
    }
}

// simulation.py:143-153: min over the whole array of dx/(|u|+c), dy/(|v|+c)
__global__ __launch_bounds__(256) void k_sw_cfl(const double *__restrict__ U, Geom g, double grav,
                                                double dx, double dy, double *__restrict__ partial)
{
    double m = INFINITY;
    for (int i = blockIdx.y; i < g.qx; i += gridDim.y)
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < g.qy; j += gridDim.x * blockDim.x) {
            const size_t k = (size_t)i * g.pitch + j;
            const double h = U[k], u = pdiv(U[g.plane + k], h), v = pdiv(U[2 * g.plane + k], h);
            const double cs = psqrt(grav * h);
            m = fmin(m, fmin(pdiv(dx, fabs(u) + cs), pdiv(dy, fabs(v) + cs)));
        }
    m = block_reduce_min(m);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
}

#endif   // !PYRO_FAST

// ---- the whole step in ONE launch: row-marching wavefronts (like comp_wave.hip) ------------
// One wavefront = 64 columns (lane = column, 58 updated: a cell's update reaches three columns
// either way), marching down a chunk of rows.  Row k arrives (load, primitives); the rows'
// primitive window (k-4 .. k), the conserved rows still to be updated, and what row c - 1 hands
// to row c = k - 2 (its x / y face states, F_xT, F_yT, F_x) live in registers; the y direction
// comes from the neighbouring lanes by whole-wave DPP rotations.  Per iteration: slopes, tracing
// and the two transverse Riemann problems of row c, the final x flux on its lower face, the
// final y flux and the conservative update of row c - 1 -- the same device functions, on the
// same operands, in the same order as the staged kernels above (which stay: every stage
// dumpable for the parity tests): bit-identical.  No LDS, no barrier; HBM sees the state once
// in (x 64/58 and the chunks' apron rows) and once out instead of the staged set's ~100
// doubles per cell.  The new time level goes to the state's second buffer (neighbouring
// chunks still read the old one); the ghost frame is carried over.
constexpr int SWW_OUT = 58, SWW_REACH = 3;
// stage boundary: the scheduler may not move instructions across it (left alone it
// interleaves the four Riemann problems of an iteration for ILP and spills: comp_wave.hip)
#if defined(PYRO_EMU)
#define SWW_FENCE() do {} while (0)
#else
#define SWW_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
struct SWW {
    double dx, dy, dt, g;
    int limiter;
    int ncb, L, nunits;
    int prio_duty;      // eighths of the time the second wavefront of a SIMD holds the priority (0: age decides)
    int nsb, n_extra;   // row strips; column strips [0, n_extra) are cut into nsb + 1 (common.h: wave_fill_extra)
    int *prio_board;    // rows-left board of the SIMD pairs (comp_wave.hip; nullptr: turns by prio_duty) and the launch's tag
    int prio_tag;
};

#if !defined(PYRO_EMU)
template <int CTRL> __device__ __forceinline__ double sww_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sww_m1(double v) { return sww_dpp<0x13C>(v); }   // wave_ror:1
__device__ __forceinline__ double sww_p1(double v) { return sww_dpp<0x134>(v); }   // wave_rol:1
#else
__device__ __forceinline__ double sww_m1(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double sww_p1(double v) { return __shfl_down(v, 1, 64); }
#endif
__device__ __forceinline__ V4 sww_m1(const V4 &v) { return V4{{sww_m1(v.a[0]), sww_m1(v.a[1]), sww_m1(v.a[2]), sww_m1(v.a[3])}}; }
__device__ __forceinline__ V4 sww_p1(const V4 &v) { return V4{{sww_p1(v.a[0]), sww_p1(v.a[1]), sww_p1(v.a[2]), sww_p1(v.a[3])}}; }

template <int RS>
__device__ __forceinline__ V4 sww_riemann(const V4 &Ul, const V4 &Ur, double g, bool x, double sg)
{
#if PYRO_FAST && !defined(SWE_DENSE_ROE)
    return RS == 1 ? sw_hllc(Ul, Ur, g, x) : sw_roe(Ul, Ur, g, x, sg);
#else
    (void)sg;
    return RS == 1 ? sw_hllc(Ul, Ur, g, x) : sw_roe(Ul, Ur, g, x);
#endif
}

// S (device-side stepping, pyrohip_swe_evolve): this step's dt from the step scalars the policy
// kernel left; partial: the wavefront's minimum of dx / (|u| + c), dy / (|v| + c) over the cells it
// updated (swe/simulation.py:143-153 on the new state: the next step's CFL minimum without a
// pass of its own -- 98 us of a 1.2 ms step at 4096^2)
// L4: swe.limiter = 2 (the default).  limit4 needs the limit2 slopes of the two neighbouring cells:
// every cell's centred limit2 is computed once -- along x carried in the row window, along y
// fetched from the neighbouring lanes -- instead of twice per direction (round 6: 8 of the 16
// limit2 evaluations per cell and row, and the two-cells-away DPP moves; bit-identical).
#if PYRO_FAST
#define SW_LIMIT2 sw_half_limit2
#define SW_LIMIT4_FROM sw_half_limit4_from
#else
#define SW_LIMIT2 limit2
#define SW_LIMIT4_FROM limit4_from
#endif
template <int RS, bool L4>   // swe.riemann: 0 Roe, 1 HLLC
__global__ __launch_bounds__(64, 2) void k_sw_wave(const double *__restrict__ Uin,
                                                   double *__restrict__ Uout, Geom g, SWW P,
                                                   const StepScalars *__restrict__ S,
                                                   double *__restrict__ partial)
{
    const int l = threadIdx.x & 63;
    const int per = (P.nunits + 7) / 8;                     // XCD x: the units [x per, (x + 1) per)
    const int unit = pyro_uniform(((int)blockIdx.x % 8) * per + (int)blockIdx.x / 8);
    if (unit >= P.nunits) return;
    if (S) {
        if (!S->active) {      // past tmax: nothing happens
            if (l == 0 && partial) partial[unit] = INFINITY;
            return;
        }
        P.dt = S->dt;
    }
    double amax = 0.0, bmax = 0.0;      // running maxima of |u| + c, |v| + c over the cells updated
    const int nreg = P.ncb * P.nsb;      // (units behind: the extra strip of the column strips [0, n_extra))
    const int cb = pyro_uniform(unit < nreg ? unit % P.ncb : unit - nreg);
    const int sb = pyro_uniform(unit < nreg ? unit / P.ncb : P.nsb);
    int i0 = g.ilo + sb * P.L;                              // rows [i0, i1)
    int i1 = (i0 + P.L < g.ihi + 1) ? i0 + P.L : g.ihi + 1;
    if (cb < P.n_extra) {      // nsb + 1 strips of equal length (to a row)
        i0 = g.ilo + (int)((long)sb * g.nx / (P.nsb + 1));
        i1 = g.ilo + (int)((long)(sb + 1) * g.nx / (P.nsb + 1));
    }
    const int j = g.jlo - SWW_REACH + cb * SWW_OUT + l;
    const int jc = j < 0 ? 0 : (j < g.qy ? j : g.qy - 1);
    const bool jout = l >= SWW_REACH && l < SWW_REACH + SWW_OUT && j >= g.jlo && j <= g.jhi;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const double dtdx = P.dt / P.dx, dtdy = P.dt / P.dy;                  // k_sw_update
    const double hdtdx = 0.5 * (P.dt / P.dx), hdtdy = 0.5 * (P.dt / P.dy);   // k_sw_final
    const double sg = (PYRO_FAST && RS == 0) ? psqrt(P.g) : 0.0;             // (contracted Roe solver)
    // the rows of a strip as [scalar base of the strip's first row, per plane] + [32-bit byte offset]
    // (saddr form of the loads / stores: comp_wave.hip)
    const int rbase = (i0 - 6 > 0) ? i0 - 6 : 0;
    const char *const sbase_in = (const char *)(Uin + (size_t)rbase * p);
    char *const sbase_out = (char *)(Uout + (size_t)rbase * p);
    const unsigned pitch8 = (unsigned)p * 8u, lane8 = (unsigned)jc * 8u;
    const size_t plb = pl * sizeof(double);
    auto loadU = [&](int row) {
        row = row < 0 ? 0 : (row > g.qx - 1 ? g.qx - 1 : row);
#if defined(PYRO_EMU)
        return ld4(Uin, pl, (size_t)row * p + jc);
#else
        const unsigned off = (unsigned)(row - rbase) * pitch8 + lane8;
        return V4{{*(const double *)(sbase_in + off), *(const double *)(sbase_in + plb + off),
                   *(const double *)(sbase_in + 2 * plb + off), *(const double *)(sbase_in + 3 * plb + off)}};
#endif
    };
    const V4 zero{{1.0, 0.0, 0.0, 0.0}};      // (h = 1: the warm-up rows divide by it)
    // rows k-4 .. k of the primitives in registers.  (The conserved row k-3, which is updated
    // when row k arrives, is read a second time one iteration ahead -- an L1 / L2 hit.)  What
    // row c-1 = k-3 left for row c -- its upper x state, its y states, F_xT, F_yT and F_x on
    // its lower faces: six 4-vectors -- sits in a per-lane LDS stash (own-lane slots, no
    // barrier: the LDS operations of a wavefront complete in order) and is read where it is
    // used: with them in registers the Roe instance spilled 164 B per lane.
    __shared__ double stash[6 * 4][64];
    enum { S_XP = 0, S_YP = 4, S_YM = 8, S_FXT = 12, S_FYT = 16, S_FX = 20 };
    auto put = [&](int slot, const V4 &v) {
#pragma unroll
        for (int n = 0; n < 4; n++) stash[slot + n][l] = v.a[n];
    };
    auto get = [&](int slot) { return V4{{stash[slot][l], stash[slot + 1][l], stash[slot + 2][l], stash[slot + 3][l]}}; };
    put(S_XP, zero); put(S_YP, zero); put(S_YM, zero); put(S_FXT, zero); put(S_FYT, zero); put(S_FX, zero);
    double q[5][4];
#pragma unroll
    for (int r = 0; r < 5; r++) { q[r][0] = 1.0; q[r][1] = q[r][2] = q[r][3] = 0.0; }
    // SWW_DELAY (contracted build on the GPU, round 6; comp_wave.hip has the measurement): the stores of a
    // row's update are issued at the top of the NEXT iteration, behind the consumption of the row that
    // arrived and in front of the next request -- loads and stores share vmcnt and complete in order, the
    // wait for the prefetched row included the stores issued just before it.  The old state the update
    // starts from is still read a second time (rebuilt from the primitives every step a conserved state
    // drifts: comp_wave.hip), requested behind the delayed stores at the top of the iteration that consumes it.
#if PYRO_FAST && !defined(PYRO_EMU) && !defined(PYRO_SWW_NO_DELAY)
    constexpr bool SWW_DELAY = true;
#else
    constexpr bool SWW_DELAY = false;
#endif
    V4 Upre = loadU(i0 - 3), Urep = SWW_DELAY ? zero : loadU(i0 - 6);
    V4 Upend = zero;
    auto store_row = [&](const V4 &V, int row) {
#if defined(PYRO_EMU)
        const size_t ko = (size_t)row * p + j;
#pragma unroll
        for (int n = 0; n < 4; n++) Uout[(size_t)n * pl + ko] = V.a[n];
#else
        const unsigned offo = (unsigned)(row - rbase) * pitch8 + (unsigned)j * 8u;
#pragma unroll
        for (int n = 0; n < 4; n++) *(double *)(sbase_out + (size_t)n * plb + offo) = V.a[n];
#endif
    };
    double l2a[4] = {0.0, 0.0, 0.0, 0.0}, l2b[4] = {0.0, 0.0, 0.0, 0.0};   // L4: limit2 along x centred on rows k-3, k-2
#if !defined(PYRO_EMU)
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int wslot = (int)(hw_id & 1u);
    // (comp_wave.hip: the two wavefronts of a SIMD tell each other their rows left, the one behind takes the priority)
    const bool prio_fb = P.prio_board != nullptr;
    int *prio_mine = nullptr, *prio_other = nullptr;
    int prio_seen = 0;
    if (prio_fb) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        int *const pair = P.prio_board + 2 * (int)(((xcc & 15u) << 12) | ((hw_id >> 4) & 0xfffu));
        prio_mine = pair + wslot;
        prio_other = pair + (wslot ^ 1);
    }
    auto prio_publish = [&](int k) {      // (behind the request of the next row)
        if (!prio_fb) return;
        __hip_atomic_store(prio_mine, (P.prio_tag << 16) | (i1 + 2 - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prio_seen = __hip_atomic_load(prio_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
#else
    auto prio_publish = [](int) {};
#endif
    for (int k = i0 - 3; k <= i1 + 2; k++) {
#if !defined(PYRO_EMU)
        if (prio_fb) {
            const int seen = __builtin_amdgcn_readfirstlane(prio_seen);
            const int left = i1 + 2 - k;
            const int other_left = ((seen >> 16) == P.prio_tag) ? (seen & 0xffff) : left;
            if (left > other_left) __builtin_amdgcn_s_setprio(1);
            else if (left < other_left) __builtin_amdgcn_s_setprio(0);
        } else
        if (P.prio_duty > 0) {     // the two wavefronts of a SIMD take turns at the priority (comp_wave.hip)
            const int phase = ((k - i0) >> 1) & 7;
            if (wslot ? (phase < P.prio_duty) : (phase >= P.prio_duty)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        // ---- row k arrives: primitives (k_sw_prim)
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int n = 0; n < 4; n++) q[r][n] = q[r + 1][n];
        }
        const V4 Uk = Upre;                  // row k
        V4 Uold = Urep;                      // row k-3: requested an iteration ago (or, SWW_DELAY, below)
        if (!SWW_DELAY) {
        Upre = loadU(k + 1);
        Urep = loadU(k - 2);
        prio_publish(k);
        }
        q[4][0] = Uk.a[0];
        q[4][1] = sw_vel(Uk.a[1], Uk.a[0]);
        q[4][2] = sw_vel(Uk.a[2], Uk.a[0]);
        q[4][3] = sw_vel(Uk.a[3], Uk.a[0]);
        if (SWW_DELAY) {
#if !defined(PYRO_EMU)
            asm volatile("" : "+v"(q[4][0]), "+v"(q[4][1]), "+v"(q[4][2]), "+v"(q[4][3]));   // (the arrived row is consumed HERE)
#endif
            SWW_FENCE();
            if (k - 1 - 2 >= i0 + 1 && jout) store_row(Upend, k - 4);      // the update iteration k-1 made (row c-1 = k-4)
            Uold = loadU(k - 3);                 // (in front of the next row's request: its wait leaves that one out)
            Upre = loadU(k + 1);
            prio_publish(k);
            SWW_FENCE();
        }
        const int c = k - 2;                 // window row 2
        double l2n[4], l2m_[4];              // L4: limit2 along x centred on row k-1 = c+1, on row c-1
        if (L4) {
#pragma unroll
            for (int n = 0; n < 4; n++) {
                l2n[n] = SW_LIMIT2(q[2][n], q[3][n], q[4][n]);
                l2m_[n] = l2a[n]; l2a[n] = l2b[n]; l2b[n] = l2n[n];
            }
        }
        if (c < i0 - 1) continue;
        // ---- row c: limited slopes, tracing (k_sw_states)
        double qc[4], dqx[4], dqy[4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            qc[n] = q[2][n];
            const double m1 = sww_m1(qc[n]), p1 = sww_p1(qc[n]);
            if (L4) {
                // (contracted build: half slopes, traced by sw_trace_h)
                dqx[n] = 1.0 * SW_LIMIT4_FROM(l2m_[n], l2n[n], q[1][n], q[2][n], q[3][n]);
                const double l2y = SW_LIMIT2(m1, qc[n], p1);
                dqy[n] = 1.0 * SW_LIMIT4_FROM(sww_m1(l2y), sww_p1(l2y), m1, qc[n], p1);
            } else {
                dqx[n] = 1.0 * limited_slope(q[0][n], q[1][n], q[2][n], q[3][n], q[4][n], P.limiter);
                dqy[n] = 1.0 * limited_slope(sww_m1(m1), m1, qc[n], p1, sww_p1(p1), P.limiter);
            }
        }
        double lo[4], hi[4];
#if PYRO_FAST
        if (L4) sw_trace_h(qc, dqx, P.g, P.dt / P.dx, true, lo, hi); else
#endif
        sw_trace(qc, dqx, P.g, P.dt / P.dx, true, lo, hi);
        const V4 XM = sw_prim_to_cons(lo), XP = sw_prim_to_cons(hi);
#if PYRO_FAST
        if (L4) sw_trace_h(qc, dqy, P.g, P.dt / P.dy, false, lo, hi); else
#endif
        sw_trace(qc, dqy, P.g, P.dt / P.dy, false, lo, hi);
        const V4 YM = sw_prim_to_cons(lo), YP = sw_prim_to_cons(hi);
        SWW_FENCE();
        // ---- transverse Riemann problems on the lower faces of row c (k_sw_riemann_t)
        const V4 XPm = get(S_XP);
        const V4 FXT = sww_riemann<RS>(XPm, XM, P.g, true, sg);
        SWW_FENCE();
        const V4 FYT = sww_riemann<RS>(sww_m1(YP), YM, P.g, false, sg);
        const V4 FYT_p = sww_p1(FYT);
        SWW_FENCE();
        if (c >= i0) {
            // ---- final x flux on the lower face of row c (k_sw_final)
            const V4 FYTm = get(S_FYT);
            const V4 Uxl = sw_corrected(XPm, sww_p1(FYTm), FYTm, hdtdy);
            const V4 Uxr = sw_corrected(XM, FYT_p, FYT, hdtdy);
            const V4 Fx = sww_riemann<RS>(Uxl, Uxr, P.g, true, sg);
            SWW_FENCE();
            if (c >= i0 + 1) {
                // ---- row r = c-1: final y flux, conservative update (k_sw_final, k_sw_update)
                const V4 FXTm = get(S_FXT);
                const V4 Uyl = sww_m1(sw_corrected(get(S_YP), FXT, FXTm, hdtdx));
                const V4 Uyr = sw_corrected(get(S_YM), FXT, FXTm, hdtdx);
                const V4 Fy = sww_riemann<RS>(Uyl, Uyr, P.g, false, sg);
                const V4 Fy_p = sww_p1(Fy);
                const V4 Fxm = get(S_FX);
                if (jout) {
                    V4 Un;
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        Un.a[n] = Uold.a[n] + (dtdx * (Fxm.a[n] - Fx.a[n]) + dtdy * (Fy.a[n] - Fy_p.a[n]));
                    if (SWW_DELAY) Upend = Un;
                    else store_row(Un, c - 1);
                    if (partial) {     // k_sw_cfl's quantities: one division per direction at the end
                        const double u = pdiv(Un.a[1], Un.a[0]), v = pdiv(Un.a[2], Un.a[0]);
                        const double cs = psqrt(P.g * Un.a[0]);
                        amax = fmax(amax, fabs(u) + cs);
                        bmax = fmax(bmax, fabs(v) + cs);
                    }
                }
            }
            put(S_FX, Fx);
        }
        put(S_XP, XP); put(S_YP, YP); put(S_YM, YM); put(S_FXT, FXT); put(S_FYT, FYT);
    }
    if (SWW_DELAY && i1 >= i0 + 1 && jout) store_row(Upend, i1 - 1);      // the update the last iteration made
    if (partial) {
        // min over cells of dx / a = dx / max a (the correctly rounded quotient is monotone)
        const double m = fmin(amax > 0.0 ? pdiv(P.dx, amax) : INFINITY, bmax > 0.0 ? pdiv(P.dy, bmax) : INFINITY);
        const double wm = wave_reduce_min(m);
        if (l == 0) partial[unit] = wm;
    }
}

// rows per chunk of the fused step (a chunk costs L + 6 iterations): whole rounds of resident
// wavefronts (two per SIMD) + one chunk time for the stragglers of the last round; ONE round when
// everything is resident at once -- the rule of the compressible kernel (comp_wave.hip: wave_rows).
// (Round 5 left the one-round case out: without turns at the priority a single round ends with
// every SIMD's younger wavefront alone -- 4096^2: one round of 147 rows 1.26 ms, 1.5 rounds of 96
// rows 1.21.  Round 6, with the turns (sww_prio_duty): 96 rows 0.614 ms, 147 rows 0.620 / 0.581 /
// 0.573 ms with the second wavefront holding the priority 0 / 5 / 6 eighths of the time.)
static int sww_rows(int nx, int ncb, int cus)
{
    const long slots = 8L * cus;
    if (nx <= 16) return nx;
    long best_cost = -1;
    int best = 16;
    for (int L = 16; L <= 160 && L <= nx; L++) {
        const long waves = (long)ncb * ((nx + L - 1) / L);
        const long rounds = (waves + slots - 1) / slots;
        const long cost = (waves <= slots ? 1 : rounds + 1) * (L + 6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = L; }
    }
    return best;
}

#if !defined(PYRO_SWW_NO_EXTRA)
static int sww_fill_extra(int ncb, int nsb, int nx, int slots) { return wave_fill_extra(ncb, nsb, nx, slots); }
#else
static int sww_fill_extra(int, int, int, int) { return 0; }
#endif

// ghost frame of the four planes from one buffer to the other (1-d grid: 2 ng blocks of
// columns for the ghost rows, then row blocks for the ghost columns)
__global__ __launch_bounds__(256) void k_sw_copy_frame(const double *__restrict__ src,
                                                        double *__restrict__ dst, Geom g)
{
    const int ng = g.ng, ncolb = (g.qy + 255) / 256, nrowb = 2 * ng * ncolb;
    const int b = blockIdx.x;
    int i, j;
    if (b < nrowb) {
        const int gr = b / ncolb;                       // ghost row index 0 .. 2 ng - 1
        i = gr < ng ? gr : g.ihi + 1 + (gr - ng);
        j = (b % ncolb) * 256 + (int)threadIdx.x;
        if (j >= g.qy) return;
    } else {
        const int rpb = 256 / (2 * ng);                 // interior rows per block
        const int t = (int)threadIdx.x;
        if (t >= rpb * 2 * ng) return;
        i = g.ilo + (b - nrowb) * rpb + t / (2 * ng);
        if (i > g.ihi) return;
        const int gc = t % (2 * ng);
        j = gc < ng ? gc : g.jhi + 1 + (gc - ng);
    }
    const size_t k = (size_t)i * g.pitch + j;
#pragma unroll
    for (int n = 0; n < 4; n++) dst[(size_t)n * g.plane + k] = src[(size_t)n * g.plane + k];
}

// one step by the one-launch kernel: new time level into the second buffer, ghost frame carried
// over (unless the caller has filled both frames: frame_done), buffers swapped.  S / part: see
// k_sw_wave; *nparts = wavefronts of the launch
int swe_step_wave(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann,
                  double dt, const StepScalars *S, double *part, int *nparts, bool frame_done)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    if (!s->alt_base) {
        const size_t n = (size_t)s->nvar * g.plane + 16;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->alt_base, n * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(s->alt_base, 0, n * sizeof(double), c->stream));
    }
    SWW P{dx, dy, dt, grav, limiter, 0, 0, 0, 0, 0, 0, nullptr, 0};
    const int slots = 8 * (c->num_cus > 0 ? c->num_cus : 256);
    P.ncb = (g.ny + SWW_OUT - 1) / SWW_OUT;
    P.L = sww_rows(g.nx, P.ncb, c->num_cus > 0 ? c->num_cus : 256);
    P.nsb = (g.nx + P.L - 1) / P.L;
    P.n_extra = sww_fill_extra(P.ncb, P.nsb, g.nx, slots);
    P.nunits = P.ncb * P.nsb + P.n_extra;
    if (nparts) *nparts = P.nunits;
    {
        // one round of resident wavefronts: the pair of a SIMD ends together -- the one with more rows left
        // takes the priority (GPU; comp_wave.hip), turns by phase otherwise
        P.prio_duty = (P.nunits <= slots) ? 6 : 0;
#if !defined(PYRO_EMU) && !defined(PYRO_SWW_NO_FEEDBACK)
        if (P.prio_duty > 0) PYRO_TRY(prio_board_acquire(c, &P.prio_board, &P.prio_tag));
#endif
    }
    double *Uout = s->alt_base + geom_lead(g);
    const dim3 grid(8 * ((P.nunits + 7) / 8)), block(64);
    using KernelT = void (*)(const double *, double *, Geom, SWW, const StepScalars *, double *);
    static const KernelT kernels[2][2] = {{k_sw_wave<0, false>, k_sw_wave<0, true>},
                                          {k_sw_wave<1, false>, k_sw_wave<1, true>}};
    PYRO_LAUNCH(c, "k_sw_wave", kernels[riemann == 1 ? 1 : 0][limiter == 2 ? 1 : 0], grid, block, 0,
                (const double *)s->d, Uout, g, P, S, part);
    if (!frame_done) {
        // the ghost frame is carried over (the reference updates the interior in place)
        const int fb = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + 256 / (2 * g.ng) - 1) / (256 / (2 * g.ng));
        hipLaunchKernelGGL(k_sw_copy_frame, dim3(fb), dim3(256), 0, c->stream, (const double *)s->d, Uout, g);
    }
    PYRO_CHECK_HIP(hipGetLastError());
    double *old_base = s->base;       // the buffers change places
    s->base = s->alt_base;
    s->alt_base = old_base;
    s->d = s->base + geom_lead(g);
    s->next_cfl_min = -1.0;
    s->ghost_by_rules = false;
    s->stages_valid = false;       // the one-launch kernel keeps no stage planes
    return 0;
}

#if !PYRO_FAST
// wavefronts of a launch of the one-launch step on this grid (= CFL partials it leaves): the
// caller of a device-side run sizes the reduction buffer with it BEFORE the first launch
int swe_wave_units(const Geom &g, int cus)
{
    const int ncb = (g.ny + SWW_OUT - 1) / SWW_OUT;
    const int L = sww_rows(g.nx, ncb, cus > 0 ? cus : 256);
    const int nsb = (g.nx + L - 1) / L;
    return ncb * nsb + sww_fill_extra(ncb, nsb, g.nx, 8 * (cus > 0 ? cus : 256));
}
#endif

}  // namespace PYRO_SWNS
#if !PYRO_FAST
namespace swf {    // the contracted unit (swe_fast)
int swe_step_wave(pyrohip_state *, double, double, double, int, int, double, const StepScalars *, double *,
                  int *, bool);
}
// (comp_api.hip: the small launches of a device-side run)
int launch_fill_frame2(pyrohip_state *s, bool *done);
int launch_dt_policy(pyrohip_ctx *c, StepScalars *S, const double *cflmin, const int *flag, double *dts,
                     int slot, int final_call, const double *part, int nparts, double *minout);
int launch_fill_frame2_policy(pyrohip_state *s, StepScalars *S, const double *cflmin, const int *flag, double *dts,
                              int slot, const double *part, int nparts, double *minout, bool *merged);
int restore_frame_after_inactive(pyrohip_state *s, int steps, int max_steps, bool halo_ok, bool sph_ok);
#endif
}  // namespace pyro

#if !PYRO_FAST
using namespace pyro;
using namespace pyro::swx;

static int sw_work(pyrohip_state *s)
{
    if (s->work_planes >= (size_t)SW_NPL) return 0;
    if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
    s->work = nullptr; s->work_planes = 0;
    const size_t n = s->g.plane * SW_NPL + 16;
    PYRO_CHECK_HIP(hipMalloc((void **)&s->work, n * sizeof(double)));
    PYRO_CHECK_HIP(hipMemsetAsync(s->work, 0, n * sizeof(double), s->ctx->stream));
    s->work_planes = SW_NPL;
    return 0;
}

static int sw_check(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(s->nvar == 4, "swe state must have 4 variables (height, x-momentum, y-momentum, fuel)");
    PYRO_REQUIRE(s->g.ng >= 4, "swe needs ng >= 4 (swe/simulation.py:98)");
    PYRO_REQUIRE(dx > 0 && dy > 0 && grav > 0, "bad dx / dy / grav");
    PYRO_REQUIRE(limiter >= 0 && limiter <= 2, "limiter must be 0, 1 or 2");
    PYRO_REQUIRE(riemann == 0 || riemann == 1, "riemann must be 0 (Roe) or 1 (HLLC)");
    return 0;
}

// the CFL minimum of the state left in device memory (-> *dmin)
constexpr int kSwCflBlocks = 8 * 128;
// doubles of reduction scratch a device-side run needs: the step kernel's partials (one per
// wavefront, `front` of them) in front, the partials + stages of k_sw_cfl behind them
static size_t sw_reduce_doubles(size_t front) { return front + 2 * kSwCflBlocks + kMinStageBlocks + 2; }

static int sw_cfl_min_device(pyrohip_state *s, double dx, double dy, double grav, const double **dmin,
                             size_t front = 65536)
{
    pyrohip_ctx *c = s->ctx;
    const dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    // (behind the partials of the step kernel, which use the front of the same buffer)
    PYRO_TRY(c->reduce.ensure(sw_reduce_doubles(front) * sizeof(double)));
    double *part = (double *)c->reduce.p + front;
    hipLaunchKernelGGL(k_sw_cfl, grid, block, 0, c->stream, (const double *)s->d, s->g, grav, dx, dy,
                       part);
    *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int pyrohip_swe_dt(pyrohip_state *s, double dx, double dy, double grav, double cfl, double *dt_out)
{
    PYRO_TRY(sw_check(s, dx, dy, grav, 0, 0));
    PYRO_REQUIRE(dt_out, "dt_out is NULL");
    pyrohip_ctx *c = s->ctx;
    const double *dmin;
    PYRO_TRY(sw_cfl_min_device(s, dx, dy, grav, &dmin));
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *dt_out = cfl * ((double *)c->reduce_host)[0];
    return 0;
}

// kernel_set: 0 the staged kernels (every stage dumpable: pyrohip_swe_stage_dump), 1 the whole
// step in one launch (k_sw_wave), -1 the library's choice (1).  fast_math (one-launch kernel
// only): 1 the contracted build (1e-10 element-wise of the bit-faithful one), 0 bit-faithful
int pyrohip_swe_step_ex(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann,
                        double dt, int kernel_set, int fast_math)
{
    PYRO_TRY(sw_check(s, dx, dy, grav, limiter, riemann));
    PYRO_REQUIRE(dt > 0.0, "dt must be positive");
    PYRO_REQUIRE(kernel_set >= -1 && kernel_set <= 1, "kernel_set must be -1, 0 or 1");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    if (kernel_set != 0)
        return fast_math ? swf::swe_step_wave(s, dx, dy, grav, limiter, riemann, dt, nullptr, nullptr, nullptr, false)
                         : swx::swe_step_wave(s, dx, dy, grav, limiter, riemann, dt, nullptr, nullptr, nullptr, false);
    PYRO_TRY(sw_work(s));
    const SW P{dx, dy, dt, grav, limiter, riemann};
    double *W = s->work + geom_lead(g);
    const dim3 block(256);
    const dim3 gridA((g.qy + 255) / 256, g.qx), gridR1((g.ny + 2 + 255) / 256, g.nx + 2),
        gridF((g.ny + 1 + 255) / 256, g.nx + 1), gridI((g.ny + 255) / 256, g.nx);
    PYRO_LAUNCH(c, "k_sw_prim", k_sw_prim, gridA, block, 0, (const double *)s->d, W, g);
    PYRO_LAUNCH(c, "k_sw_states", k_sw_states, gridR1, block, 0, W, g, P);
    PYRO_LAUNCH(c, "k_sw_riemann_t", k_sw_riemann_t, gridR1, block, 0, W, g, P);
    PYRO_LAUNCH(c, "k_sw_final", k_sw_final, gridF, block, 0, W, g, P);
    PYRO_LAUNCH(c, "k_sw_update", k_sw_update, gridI, block, 0, s->d, (const double *)W, g, P);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;
    s->ghost_by_rules = false;
    s->stages_valid = true;
    return 0;
}

int pyrohip_swe_step_ks(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann,
                        double dt, int kernel_set)
{
    return pyrohip_swe_step_ex(s, dx, dy, grav, limiter, riemann, dt, kernel_set, 0);
}

int pyrohip_swe_step(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann,
                     double dt)
{
    return pyrohip_swe_step_ex(s, dx, dy, grav, limiter, riemann, dt, -1, 0);
}

// Up to max_steps iterations of the swe driver loop (pyro_sim.py:241-281 with swe/simulation.py:
// 143-193: ghost fill, CFL time step, evolve) without a host round trip per step -- as
// pyrohip_comp_evolve: the ghost fill (both buffers' frames in one launch where the boundaries
// are outflow / reflect / periodic), the driver's dt policy in a kernel on the CFL minimum the
// previous step's wavefronts left, the one-launch step kernel.  One synchronisation at the end.
int pyrohip_swe_evolve(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann,
                       int fast_math, double cfl, pyrohip_dt_policy *pol, int max_steps,
                       int *steps_done, double *dts_out)
{
    PYRO_TRY(sw_check(s, dx, dy, grav, limiter, riemann));
    PYRO_REQUIRE(pol && steps_done && max_steps >= 1, "NULL argument / max_steps must be positive");
    PYRO_REQUIRE(!s->nb_set && !s->user_bc && !s->ramp_bc && !s->sph,
                 "device-side stepping: swe runs on a single Cartesian domain with the standard boundary types");
    // from the second step on the CFL minimum is the step kernel's (interior of the new state):
    // equal to the reference's whole-array minimum only where every ghost cell is an image of an
    // interior cell (a constant-value side is not: swe/simulation.py:127-141 after fill_BC_all)
    for (int k = 0; k < 4 * s->nvar; k++) {
        const int b = s->bc[k];
        PYRO_REQUIRE(b == PYROHIP_BC_OUTFLOW || b == PYROHIP_BC_REFLECT_EVEN || b == PYROHIP_BC_REFLECT_ODD ||
                         b == PYROHIP_BC_PERIODIC,
                     "device-side stepping: outflow / reflect / periodic boundaries only");
    }
    pyrohip_ctx *c = s->ctx;
    if (!s->d_scal) PYRO_CHECK_HIP(hipMalloc((void **)&s->d_scal, sizeof(StepScalars)));
    if (!s->d_flag) {
        PYRO_CHECK_HIP(hipMalloc((void **)&s->d_flag, sizeof(int)));
    }
    if (s->dts_cap < max_steps + 1) {
        if (s->d_dts) PYRO_CHECK_HIP(hipFree(s->d_dts));
        s->d_dts = nullptr;
        // (not less than 1024: a run that asks for more steps call by call must not free + allocate every time)
        const int cap = max_steps + 1 > 1024 ? max_steps + 1 : 1024;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->d_dts, (size_t)cap * sizeof(double)));
        s->dts_cap = cap;
    }
    if (!s->alt_base) {     // (k_fill_frame2 writes the second buffer's frame)
        const size_t n = (size_t)s->nvar * s->g.plane + 16;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->alt_base, n * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(s->alt_base, 0, n * sizeof(double), c->stream));
    }
    StepScalars H;
    memset(&H, 0, sizeof(H));
    H.t = pol->t; H.dt_old = pol->dt_old; H.n = pol->n;
    H.tmax = pol->tmax; H.f0 = pol->init_tstep_factor; H.mx = pol->max_dt_change;
    H.fix_dt = pol->fix_dt; H.cfl = cfl; H.dx = dx; H.dy = dy;
    // (the CFL minimum the previous call's last step left, where nothing touched the state since:
    // pyrohip_comp_evolve)
    const bool min_cached = cfl_min_cached(s, 2, grav, dx, dy);
    H.min0 = min_cached ? s->next_cfl_min : 0.0;
    PYRO_CHECK_HIP(hipMemcpyAsync(s->d_scal, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));      // H is on this stack frame
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    // one partial per wavefront of the step kernel (46 978 at 16384^2, 128 820 at 32768^2 on 256
    // CUs), sized BEFORE the first launch: growing the buffer later would move what dmin points at
    const size_t nunits = (size_t)swx::swe_wave_units(s->g, c->num_cus);
    PYRO_TRY(c->reduce.ensure(sw_reduce_doubles(nunits) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    const double *dmin = nullptr;
    const double *pend = nullptr;
    int npend = 0, rc = 0;
    for (int m = 0; m < max_steps && rc == 0; m++) {
        bool frame_done = false, merged = false;
        if (m > 0) {       // ghost frames of both buffers + the dt policy in one launch (comp_api.hip)
            rc = launch_fill_frame2_policy(s, s->d_scal, dmin, s->d_flag, s->d_dts, m, pend, npend,
                                           const_cast<double *>(dmin), &merged);
            if (rc) break;
            frame_done = merged;
        }
        if (!merged) {
        rc = launch_fill_frame2(s, &frame_done);          // pyro_sim.py:250: fill_BC_all
        if (rc) break;
        if (m == 0) {      // the CFL minimum of the state as handed over (whole array, filled)
            if (min_cached) dmin = &s->d_scal->min0;
            else rc = sw_cfl_min_device(s, dx, dy, grav, &dmin, nunits);
            if (rc) break;
        }
        // (later steps: the policy kernel takes the minimum of the step kernel's partials itself)
        rc = launch_dt_policy(c, s->d_scal, dmin, s->d_flag, s->d_dts, m, 0, pend, npend,
                              const_cast<double *>(dmin));
        if (rc) break;
        }
        int np = 0;
        rc = fast_math ? swf::swe_step_wave(s, dx, dy, grav, limiter, riemann, 0.0, s->d_scal, part, &np, frame_done)
                       : swx::swe_step_wave(s, dx, dy, grav, limiter, riemann, 0.0, s->d_scal, part, &np, frame_done);
        if (rc == 0 && (size_t)np > nunits) {
            set_error("pyrohip_swe_evolve: the step kernel left more CFL partials than were sized for");
            rc = PYROHIP_ERR_ARG;
        }
        pend = part; npend = np;
    }
    PYRO_TRY(rc);
    PYRO_TRY(launch_dt_policy(c, s->d_scal, dmin, s->d_flag, s->d_dts, max_steps, 1, pend, npend,
                              const_cast<double *>(dmin)));
    char *hb = (char *)c->reduce_host;                       // 256 pinned bytes
    PYRO_CHECK_HIP(hipMemcpyAsync(hb, s->d_scal, sizeof(StepScalars), hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(hb + sizeof(StepScalars) + 8, dmin, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    if (dts_out)
        PYRO_CHECK_HIP(hipMemcpyAsync(dts_out, s->d_dts, (size_t)max_steps * sizeof(double),
                                      hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    memcpy(&H, hb, sizeof(H));
    const double lastmin = *(double *)(hb + sizeof(StepScalars) + 8);
    // max_steps swaps were made; the last state that advanced sits H.steps swaps from the start
    if ((max_steps - H.steps) % 2) {
        double *old_base = s->base;
        s->base = s->alt_base;
        s->alt_base = old_base;
        s->d = s->base + geom_lead(s->g);
    }
    PYRO_TRY(restore_frame_after_inactive(s, H.steps, max_steps, false, false));
    // the minimum of the last launch belongs to the state only if that launch advanced it
    s->next_cfl_min = (H.steps == max_steps && !H.dead) ? lastmin : -1.0;
    s->cfl_kind = 2;
    s->cfl_par[0] = grav; s->cfl_par[1] = dx; s->cfl_par[2] = dy;
    s->ghost_by_rules = false;
    pol->t = H.t; pol->dt_old = H.dt_old; pol->n = H.n;
    *steps_done = H.steps;
    return 0;
}

// stage: 0 Uxl0 1 Uxr0 2 Uyl0 3 Uyr0 (face states before the transverse
// terms, reference face indexing), 4 FxT 5 FyT 6 Fx 7 Fy -> host (qx, qy, 4)
int pyrohip_swe_stage_dump(pyrohip_state *s, int stage, double *out)
{
    PYRO_REQUIRE(s && out, "NULL argument");
    PYRO_REQUIRE(stage >= 0 && stage < 8, "stage out of range");
    PYRO_REQUIRE(s->work_planes >= (size_t)SW_NPL && s->stages_valid,
                 "no staged swe step has been run on this state (the last step was the one-launch "
                 "kernel, which keeps no stage planes: pyrohip_swe_step_ks with kernel_set 0)");
    static const int first[8] = {SW_XP, SW_XM, SW_YP, SW_YM, SW_FXT, SW_FYT, SW_FX, SW_FY};
    const Geom &g = s->g;
    pyrohip_ctx *c = s->ctx;
    std::vector<double> tmp((size_t)g.qx * g.qy);
    // XP / YP are stored at the cell whose upper face they sit on: shift by one
    // cell so that out[i, j] is the reference's U_xl[i, j] / U_yl[i, j]
    const int si = (stage == 0) ? 1 : 0, sj = (stage == 2) ? 1 : 0;
    for (int n = 0; n < 4; n++) {
        PYRO_CHECK_HIP(hipMemcpy2DAsync(tmp.data(), g.qy * sizeof(double),
                                        s->work + geom_lead(g) + (size_t)(first[stage] + n) * g.plane,
                                        g.pitch * sizeof(double), g.qy * sizeof(double), g.qx,
                                        hipMemcpyDeviceToHost, c->stream));
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < g.qx; i++)
            for (int j = 0; j < g.qy; j++) {
                const int ii = i - si, jj = j - sj;
                out[((size_t)i * g.qy + j) * 4 + n] =
                    (ii >= 0 && jj >= 0) ? tmp[(size_t)ii * g.qy + jj] : 0.0;
            }
    }
    return 0;
}

}  // extern "C"
#endif   // !PYRO_FAST
