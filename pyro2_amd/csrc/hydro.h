// Per-cell / per-face Euler arithmetic for the compressible CTU update.
// Written once, used by the staged kernels and by the fused LDS kernels.
// Operation order follows the reference expression by expression (cited
// below) so that the -ffp-contract=off build is bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>

#include "stencil.h"

#ifndef PYRO_FAST
#define PYRO_FAST 0
#endif

namespace pyro {
// The functions below differ between the bit-faithful and the fast build of a unit.  They
// are inline functions of namespace pyro in BOTH, so in a host build (the emulator of
// tests/emu links comp_exact.o and comp_fast.o into one library) the linker would keep ONE
// copy of every function that was not inlined and both builds would run it.  The inline
// namespace gives the two families distinct symbol names; callers do not see it.
#if PYRO_FAST
inline namespace hydro_fast {
#else
inline namespace hydro_exact {
#endif

// Division policy.  Bit-faithful build: IEEE division, exactly as written in
// the reference.  fast_math build: v_rcp_f64 (~2^-23 relative) + ONE Newton
// step (~2^-45 = 3e-14 relative) and a multiply: 4 VALU instructions instead of
// the ~12 of the IEEE sequence; divisions are ~40 % of this solver's VALU work
// (profiles/r01_fused_8192_pmc.json).  Parity of the fast build is
// tolerance-tested (north_star: 1e-10; 606-step quad and 945-step rt
// regressions included), not bit-tested.  A second Newton step (full double
// precision) costs 3 % of the step time: -DPYRO_RCP_NEWTON2.
#if PYRO_FAST && !defined(PYRO_EMU)
__device__ __forceinline__ double prcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
#ifdef PYRO_RCP_NEWTON2
    r = fma(fma(-b, r, 1.0), r, r);
#endif
    return r;
}
__device__ __forceinline__ double pdiv(double a, double b) { return a * prcp(b); }
// a / b where rb = prcp(b) was computed once for several quotients
__device__ __forceinline__ double pdivr(double a, double b, double rb) { (void)b; return a * rb; }
// sqrt of a strictly positive, normal argument (sound speeds, 1 + ...):
// v_rsq_f64 + one Goldschmidt step (~2^-45 relative, like prcp), without the
// denormal scaling / special-case code of the IEEE expansion
__device__ __forceinline__ double psqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
#ifdef PYRO_RCP_NEWTON2
    const double d = fma(-g, g, x);   // residual correction: full double precision
    g = fma(d, h, g);
#endif
    return (x > 0.0) ? g : 0.0;
}
// the same without the x > 0 select, for arguments that are positive by
// construction (1 + positive, A / (p + B)) or whose result goes through
// fmax(smallc, .) anyway: fmax returns smallc for the NaN a zero / negative
// argument produces, exactly what the selected 0 gives
__device__ __forceinline__ double psqrt_nc(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
#ifdef PYRO_RCP_NEWTON2
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
#endif
    return g;
}
// sqrt(x) and 1/sqrt(x) of a strictly positive, normal argument from ONE
// v_rsq_f64 (a transcendental issues at quarter rate: 16 cycles per wave)
__device__ __forceinline__ double psqrt_r(double x, double &rinv)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    rinv = h + h;
    return g;
}
// same for arguments that may be exactly 0 (velocity magnitudes): psqrt already
// returns 0 there
__device__ __forceinline__ double psqrt0(double x) { return psqrt(x); }
// 1 / sqrt(x) of a strictly positive, normal argument: v_rsq_f64 + one Goldschmidt step
__device__ __forceinline__ double prsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * y;
    return fma(y, fma(-h, x * y, 0.5), y);
}
#elif !PYRO_FAST && !defined(PYRO_EMU) && !defined(PYRO_IEEE_LIBCALLS)
// Bit-faithful build on the GPU: IEEE-correct division and square root WITHOUT the
// exponent scaling and the special-case fix-up of the compiler's expansion
// (v_div_scale x 2, v_div_fmas, v_div_fixup; v_cmp_class / v_ldexp around the square
// root).  The arithmetic core is the same Newton / Markstein chain that expansion runs
// -- v_rcp_f64, two Newton steps, q = a r, one correction of q with the exact residual
// fma(-b, q, a) -- so the result is the correctly rounded quotient, bit for bit, whenever
// the scaling would have been the identity: operands, reciprocal and quotient normal and
// away from the ends of the exponent range (|.| in 2^-700 .. 2^700 is ample), which is
// every quantity of a valid hydrodynamic state.  Outside that (denormal quotients, b = inf)
// the expansion's fix-up is missing: results of invalid states, flagged by the kernels'
// positivity checks.  8 instead of 11 instructions per division, 10 instead of 15 per
// square root; bit-identity with `/` and sqrt() is probed on the GPU over 2^26 operand
// pairs (tools/div_probe.hip -> profiles/r03_div_probe.txt) and pinned by the parity tests
// of the bit-faithful build.  -DPYRO_IEEE_LIBCALLS restores the plain expressions.
__device__ __forceinline__ double prcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    // 1 / b is the quotient 1 / b: one Markstein correction makes it the rounded one
    return fma(fma(-b, r, 1.0), r, r);
}
__device__ __forceinline__ double pdiv(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double pdivr(double a, double b, double rb) { (void)rb; return pdiv(a, b); }
__device__ __forceinline__ double psqrt(double x)
{
    // the compiler's expansion of sqrt(double) without its scaling: rsq seed, one coupled
    // Goldschmidt step, two residual corrections; zero stays zero (0 * inf is avoided)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return (x == 0.0) ? x : g;
}
__device__ __forceinline__ double psqrt0(double x) { return psqrt(x); }
__device__ __forceinline__ double psqrt_nc(double x) { return psqrt(x); }
__device__ __forceinline__ double prsqrt(double x) { return pdiv(1.0, psqrt(x)); }
__device__ __forceinline__ double psqrt_r(double x, double &rinv)
{
    const double g = psqrt(x);
    rinv = pdiv(1.0, g);
    return g;
}
#elif PYRO_FAST && defined(PYRO_EMU) && defined(PYRO_EMU_FASTSEED)
// Host emulator, developer variant (PYRO_EMU_DEFS=-DPYRO_EMU_FASTSEED): the contracted build's
// reciprocal / root with the PRECISION of the GPU forms above -- a 2^-23 seed and one Newton /
// Goldschmidt step -- so that the sensitivity of a kernel to the ~3e-14 quotients can be looked
// at without a GPU (found with it: the Roe solver's eigenvector sums of the shallow-water build)
__device__ __forceinline__ double prcp(double b)
{
    double r = (double)(1.0f / (float)b);
    return fma(fma(-b, r, 1.0), r, r);
}
__device__ __forceinline__ double pdiv(double a, double b) { return a * prcp(b); }
__device__ __forceinline__ double pdivr(double a, double b, double rb) { (void)b; return a * rb; }
__device__ __forceinline__ double psqrt_r(double x, double &rinv)
{
    const double y = (double)(1.0f / sqrtf((float)x));
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    rinv = h + h;
    return g;
}
__device__ __forceinline__ double psqrt(double x) { double ri; return (x > 0.0) ? psqrt_r(x, ri) : 0.0; }
__device__ __forceinline__ double psqrt0(double x) { return psqrt(x); }
__device__ __forceinline__ double psqrt_nc(double x) { double ri; return psqrt_r(x, ri); }
__device__ __forceinline__ double prsqrt(double x) { double ri; psqrt_r(x, ri); return ri; }
#else
__device__ __forceinline__ double prsqrt(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ double psqrt0(double x) { return sqrt(x); }
__device__ __forceinline__ double psqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double psqrt_nc(double x) { return sqrt(x); }
__device__ __forceinline__ double psqrt_r(double x, double &rinv)
{
    const double g = sqrt(x);
    rinv = 1.0 / g;
    return g;
}
__device__ __forceinline__ double prcp(double b) { return 1.0 / b; }
__device__ __forceinline__ double pdiv(double a, double b) { return a / b; }
__device__ __forceinline__ double pdivr(double a, double b, double rb) { (void)rb; return a / b; }
#endif

// velocity = momentum / density (d > 0) for the primitive variables: the bit-faithful GPU
// quotient above is the IEEE one except for the SIGN of a zero quotient (-0 / d gives +0: the
// Markstein correction adds +0), and the characteristic tracing takes copysign(1, u)
// (interface.py:198-201) -- a cell at rest whose momentum is -0.0 (an odd reflection of +0.0)
// would trace the other way than the reference.  One compare + select, bit-faithful build only.
__device__ __forceinline__ double pvel(double m, double d, double rd)
{
    const double q = pdivr(m, d, rd);
#if !PYRO_FAST && !defined(PYRO_EMU) && !defined(PYRO_IEEE_LIBCALLS)
    return (m == 0.0) ? m : q;
#else
    return q;
#endif
}

struct Cons { double d, E, mx, my; };   // density, energy, x-mom, y-mom
struct Prim { double r, u, v, p; };     // rho, u, v, p

// compressible/simulation.py:49-80 (cons_to_prim) + eos.py:26
// `ok` is cleared when e <= 0 or rho <= 0 (the reference's interior assert)
__device__ __forceinline__ Prim cons_to_prim(const Cons &U, double gamma, bool *ok = nullptr)
{
    Prim q;
    q.r = U.d;
    double u = 0.0, v = 0.0, e = 0.0;
    if (U.d != 0.0) {
        const double rd = PYRO_FAST ? prcp(U.d) : 0.0;
        u = pvel(U.mx, U.d, rd);
        v = pvel(U.my, U.d, rd);
        e = pdivr(U.E - 0.5 * U.d * (u * u + v * v), U.d, rd);
    }
    q.u = u;
    q.v = v;
    q.p = U.d * e * (gamma - 1.0);
    if (ok) *ok = (e > 0.0) && (U.d > 0.0);
    return q;
}

// compressible/simulation.py:83-102 (prim_to_cons) + eos.py:72
__device__ __forceinline__ Cons prim_to_cons(const Prim &q, double gamma)
{
    Cons U;
    U.d = q.r;
    U.mx = q.u * U.d;
    U.my = q.v * U.d;
    double rhoe = pdiv(q.p, gamma - 1.0);
    U.E = rhoe + 0.5 * q.r * (q.u * q.u + q.v * q.v);
    return U;
}

// the same with gamma - 1 (and, fast build, its reciprocal) passed in: kernels
// that cannot keep derived uniforms in registers read them from a table
__device__ __forceinline__ Cons prim_to_cons_g(const Prim &q, double gm1, double rgm1)
{
    Cons U;
    U.d = q.r;
    U.mx = q.u * U.d;
    U.my = q.v * U.d;
#if PYRO_FAST
    (void)gm1;
    U.E = fma(0.5, fma(U.my, q.v, U.mx * q.u), q.p * rgm1);
#else
    double rhoe = pdivr(q.p, gm1, rgm1);
    U.E = rhoe + 0.5 * q.r * (q.u * q.u + q.v * q.v);
#endif
    return U;
}

// CFL quantities, compressible/derives.py:19-25,59 + simulation.py:284-288
// returns min(dx/(|u|+c), dy/(|v|+c)) for one cell
__device__ __forceinline__ double cfl_cell(const Cons &U, double gamma, double dx, double dy)
{
    const double rd = PYRO_FAST ? prcp(U.d) : 0.0;
    double u = pdivr(U.mx, U.d, rd);
    double v = pdivr(U.my, U.d, rd);
    double e = pdivr(U.E - 0.5 * U.d * (u * u + v * v), U.d, rd);
    double p = U.d * e * (gamma - 1.0);
    double cs = psqrt(pdivr(gamma * p, U.d, rd));
    double xt = pdiv(dx, fabs(u) + cs);
    double yt = pdiv(dy, fabs(v) + cs);
    return fmin(xt, yt);
}

// |u| + c and |v| + c of one cell: the divisors of cfl_cell.  A kernel may keep
// the running maxima and divide once at the end: min_i fl(dx / a_i) =
// fl(dx / max_i a_i) because a correctly rounded quotient is monotone in the divisor
__device__ __forceinline__ void cfl_speeds(const Cons &U, double gamma, double &ax, double &ay)
{
    const double rd = PYRO_FAST ? prcp(U.d) : 0.0;
    double u = pdivr(U.mx, U.d, rd);
    double v = pdivr(U.my, U.d, rd);
    double e = pdivr(U.E - 0.5 * U.d * (u * u + v * v), U.d, rd);
    double p = U.d * e * (gamma - 1.0);
    double cs = psqrt(pdivr(gamma * p, U.d, rd));
    ax = fabs(u) + cs;
    ay = fabs(v) + cs;
}

// 1-d flattening coefficient, pyro/mesh/reconstruction.py:123-164
//   pm2..pp2 : pressure at -2..+2,  um1/up1 : normal velocity at -1/+1
// (FT: any type with z0(), z1(), delta() -- read where the evaluation gets that far)
struct FlatK {
    double z0_, z1_, delta_;
    __device__ __forceinline__ double z0() const { return z0_; }
    __device__ __forceinline__ double z1() const { return z1_; }
    __device__ __forceinline__ double delta() const { return delta_; }
};
template <class FT>
__device__ __forceinline__ double flatten_1d_k(double pm2, double pm1, double pp1, double pp2,
                                               double um1, double up1, const FT &K)
{
    // The reference evaluates z and xi everywhere and then selects with
    // np.where(t1 > 0 and t2 > delta, xi, 1).  The selected value only depends
    // on z where the condition holds, so the divisions are evaluated lazily:
    // no compression (u_{-1} - u_{+1} <= 0) -> 1 without any division -- the
    // common case away from shocks.  Same result, bit for bit.
    const double t1b = um1 - up1;
    if (!(t1b > 0.0)) return 1.0;
    const double smallp = 1.e-10;
    const double t1 = fabs(pp1 - pm1);
    const double t2b = pdiv(t1, fmin(pp1, pm1));
    if (!(t2b > K.delta())) return 1.0;
    const double t2 = fabs(pp2 - pm2);
    const double z = pdiv(t1, fmax(t2, smallp));
    const double z0 = K.z0(), z1 = K.z1();
    return fmin(1.0, fmax(0.0, 1.0 - pdiv(z - z0, z1 - z0)));
}
__device__ __forceinline__ double flatten_1d(double pm2, double pm1, double pp1, double pp2,
                                             double um1, double up1, double z0, double z1,
                                             double delta)
{
    return flatten_1d_k(pm2, pm1, pp1, pp2, um1, up1, FlatK{z0, z1, delta});
}

// Characteristic tracing of one cell in one direction,
// compressible/interface.py:5-236.  `un`/`ut` are the normal / transverse
// velocity (u,v for idir=1; v,u for idir=2); d* the limited slopes.
// Outputs primitive states on the cell's lower face (q_r[face-]) and upper
// face (q_l[face+]) with the same (rho, un, ut, p) convention.
struct Trace { double r, un, ut, p; };

#if PYRO_FAST
// fast build: the same tracing with the characteristic projections written out.  With
// e0 = un - cs, e1 = e2 = un, e3 = un + cs the factors (e3 - e_m) (sign(e_m) + 1) and
// (e0 - e_m) (1 - sign(e_m)) of interface.py:193-201 are 0 or a multiple of cs:
//   beta_l[0] = [e0 >= 0] dtdx cs (l_0 . dq)      beta_r[3] = -[e3 < 0] dtdx cs (l_3 . dq)
//   beta_l[1,2] = [un >= 0] (dtdx/2) cs (l_1,2 . dq)   beta_r[1,2] = -[un < 0] (dtdx/2) cs (l_1,2 . dq)
// ([.] by the SIGN BIT, as np.copysign does: -0.0 counts as negative).  beta_l[0] and
// beta_r[3] vanish wherever the flow is subsonic in this direction and are evaluated
// lazily; l_1 . dq = dr - dp / cs^2, l_2 . dq = dut.  Differs from the reference's
// evaluation order by rounding only (tolerance-tested, north_star 1e-10).
__device__ __forceinline__ void trace_states(double r, double un, double ut, double p,
                                             double dr, double dun, double dut, double dp,
                                             double gamma, double dtdx, Trace &lo, Trace &hi)
{
    const double rr = prcp(r);
    double rcs;
    const double cs = psqrt_r(gamma * p * rr, rcs);      // interface.py:122
    const double hh = 0.5 * dtdx;
    const double e0 = un - cs, e3 = un + cs;
    // reference states, :174-191
    const double fl = fma(-hh, fmax(e3, 0.0), 0.5);
    const double fr = fma(hh, fmin(e0, 0.0), 0.5);
    hi.r = fma(fl, dr, r);    hi.un = fma(fl, dun, un);
    hi.ut = fma(fl, dut, ut); hi.p = fma(fl, dp, p);
    lo.r = fma(-fr, dr, r);   lo.un = fma(-fr, dun, un);
    lo.ut = fma(-fr, dut, ut); lo.p = fma(-fr, dp, p);
    // the two waves that move with the fluid
    const double w = hh * cs;
    const double a1 = fma(-(rcs * rcs), dp, dr);
    const bool pos = !__builtin_signbit(un);
    const double gp = pos ? w : 0.0, gm = w - gp;
    hi.r = fma(gp, a1, hi.r);   hi.ut = fma(gp, dut, hi.ut);
    lo.r = fma(-gm, a1, lo.r);  lo.ut = fma(-gm, dut, lo.ut);
    // the acoustic waves that run AGAINST their usual direction: supersonic flow only
    if (__builtin_expect(!__builtin_signbit(e0) || e3 < 0.0, 0)) {
        const double t = r * dun, hrcs = 0.5 * rcs, w2 = w + w;
        const double a0 = hrcs * fma(rcs, dp, -t);
        const double a3 = hrcs * fma(rcs, dp, t);
        const double bl0 = !__builtin_signbit(e0) ? w2 * a0 : 0.0;
        const double br3 = (e3 < 0.0) ? -(w2 * a3) : 0.0;
        const double cr = cs * rr, c2 = cs * cs;
        hi.r += bl0;  hi.un = fma(-bl0, cr, hi.un);  hi.p = fma(bl0, c2, hi.p);
        lo.r += br3;  lo.un = fma(br3, cr, lo.un);   lo.p = fma(br3, c2, lo.p);
    }
}
#else
__device__ __forceinline__ void trace_states(double r, double un, double ut, double p,
                                             double dr, double dun, double dut, double dp,
                                             double gamma, double dtdx, Trace &lo, Trace &hi)
{
    const double dtdx4 = 0.25 * dtdx;                // interface.py:107
#if PYRO_FAST && !defined(PYRO_EMU)
    const double rr = prcp(r);
    double rcs_f;
    const double cs = psqrt_r(gamma * p * rr, rcs_f);   // :122
#else
    const double cs = psqrt(pdiv(gamma * p, r));     // :122
#endif
    const double e0 = un - cs, e1 = un, e3 = un + cs;  // :129 / :151  (e2 == e1)

    // reference states, :174-191
    double fl = 0.5 * (1.0 - dtdx * fmax(e3, 0.0));
    double fr = 0.5 * (1.0 + dtdx * fmin(e0, 0.0));
    hi.r = r + fl * dr;   hi.un = un + fl * dun;
    hi.ut = ut + fl * dut; hi.p = p + fl * dp;
    lo.r = r - fr * dr;   lo.un = un - fr * dun;
    lo.ut = ut - fr * dut; lo.p = p - fr * dp;

    // l_m . dq as 4-term in-order sums with the zero entries dropped
    // (adding 0*x terms never changes a finite sum), :131-139 / :153-161
#if PYRO_FAST && !defined(PYRO_EMU)
    const double rcs = rcs_f;
#else
    const double rcs = PYRO_FAST ? prcp(cs) : 0.0;
#endif
    const double rc2 = PYRO_FAST ? rcs * rcs : 0.0;
    const double a0 = pdivr(-0.5 * r, cs, rcs) * dun + pdivr(0.5, cs * cs, rc2) * dp;
    const double a1 = dr + pdivr(-1.0, cs * cs, rc2) * dp;
    const double a2 = dut;
    const double a3 = pdivr(0.5 * r, cs, rcs) * dun + pdivr(0.5, cs * cs, rc2) * dp;

    // :193-201
    const double s0 = copysign(1.0, e0), s1 = copysign(1.0, e1), s3 = copysign(1.0, e3);
    (void)s3; (void)s0;
    const double bl0 = dtdx4 * (e3 - e0) * (s0 + 1.0) * a0;
    const double bl1 = dtdx4 * (e3 - e1) * (s1 + 1.0) * a1;
    const double bl2 = dtdx4 * (e3 - e1) * (s1 + 1.0) * a2;
    // bl3 carries the factor (e3 - e3) and br0 the factor (e0 - e0): both are
    // exactly zero, and adding a zero never changes the sums below, so the two
    // terms are dropped (bitwise neutral for finite states)
    const double br1 = dtdx4 * (e0 - e1) * (1.0 - s1) * a1;
    const double br2 = dtdx4 * (e0 - e1) * (1.0 - s1) * a2;
    const double br3 = dtdx4 * (e0 - e3) * (1.0 - s3) * a3;

    // sum_m beta_m r_m, in index order with the structural zeros kept where
    // they sit between non-zero terms, :203-213; r_m from :141-144 / :163-166
#if PYRO_FAST && !defined(PYRO_EMU)
    const double cr = cs * rr, c2 = cs * cs;
#else
    const double cr = pdiv(cs, r), c2 = cs * cs;
#endif
    hi.r = hi.r + (bl0 + bl1);
    hi.un = hi.un + bl0 * (-cr);
    hi.ut = hi.ut + bl2;
    hi.p = hi.p + bl0 * c2;
    lo.r = lo.r + (br1 + br3);
    lo.un = lo.un + br3 * cr;
    lo.ut = lo.ut + br2;
    lo.p = lo.p + br3 * c2;
}
#endif

// Wave-speed estimate, compressible/riemann.py:596-678 (quirk: S_r uses
// (gamma+1)/(2/gamma), :675)
// gamma and the uniform quotients of it the HLLC solver uses, evaluated once
// with the reference's expressions (kernels that keep their uniforms in a table
// pass them in; make_gask() is the inline equivalent)
struct GasK {
    double gamma;
    double ksl;    // (gamma + 1) / (2 gamma)        riemann.py:665
    double ksr;    // (gamma + 1) / (2 / gamma)      riemann.py:675 (the quirk)
    double rgp1;   // fast build: 1 / (gamma + 1)
    // (the solvers below take any type with these four accessors: a kernel that keeps its
    // uniforms in a table passes one that reads an entry when -- and only when -- a branch
    // needs it: ksl / ksr on compressed faces, rgp1 in the two-shock estimate)
    __device__ __forceinline__ double g() const { return gamma; }
    __device__ __forceinline__ double sl() const { return ksl; }
    __device__ __forceinline__ double sr() const { return ksr; }
    __device__ __forceinline__ double gp1() const { return rgp1; }
};
__device__ __forceinline__ GasK make_gask(double gamma)
{
    GasK k;
    k.gamma = gamma;
    k.ksl = pdiv(gamma + 1.0, 2.0 * gamma);
    k.ksr = pdiv(gamma + 1.0, pdiv(2.0, gamma));
    k.rgp1 = PYRO_FAST ? prcp(gamma + 1.0) : 0.0;
    return k;
}

template <class KT>
__device__ __forceinline__ void estimate_wave_speed(double rho_l, double u_l, double p_l,
                                                    double c_l, double rho_r, double u_r,
                                                    double p_r, double c_r, const KT &K,
                                                    double &S_l, double &S_r)
{
    const double gamma = K.g();
    double p_max = fmax(p_l, p_r);
    double p_min = fmin(p_l, p_r);
#if PYRO_FAST && !defined(PYRO_EMU)
    const double Q = (p_max > 2.0 * p_min) ? 3.0 : 1.0;   // only "Q > 2" is used below
#else
    double Q = pdiv(p_max, p_min);
#endif
    double rho_avg = 0.5 * (rho_l + rho_r);
    double c_avg = 0.5 * (c_l + c_r);
    double factor = rho_avg * c_avg;
    double pstar = 0.5 * (p_l + p_r) + 0.5 * (u_l - u_r) * factor;
    if (Q > 2 && (pstar < p_min || pstar > p_max)) {
        if (pstar < p_min) {   // two-rarefaction, :626-638
            // (kept inline: as an out-of-line function the call's clobber rules
            // spill more registers around it than the three pow() bodies cost)
            double z = pdiv(gamma - 1.0, 2.0 * gamma);
            double p_lr = pow(pdiv(p_l, p_r), z);
            double ustar = pdiv(pdiv(p_lr * u_l, c_l) + pdiv(u_r, c_r) +
                                    pdiv(2.0 * (p_lr - 1.0), gamma - 1.0),
                                pdiv(p_lr, c_l) + pdiv(1.0, c_r));
            pstar = 0.5 * (p_l * pow(1.0 + pdiv((gamma - 1.0) * (u_l - ustar), 2.0 * c_l),
                                     pdiv(1.0, z)) +
                           p_r * pow(1.0 + pdiv((gamma - 1.0) * (ustar - u_r), 2.0 * c_r),
                                     pdiv(1.0, z)));
        } else {               // two-shock, :640-658
            double A_r = pdiv(2.0, (gamma + 1.0) * rho_r);
            double B_r = pdivr(p_r * (gamma - 1.0), gamma + 1.0, K.gp1());
            double A_l = pdiv(2.0, (gamma + 1.0) * rho_l);
            double B_l = pdivr(p_l * (gamma - 1.0), gamma + 1.0, K.gp1());
            double p_guess = fmax(0.0, pstar);
            double g_l = psqrt_nc(pdiv(A_l, p_guess + B_l));
            double g_r = psqrt_nc(pdiv(A_r, p_guess + B_r));
            pstar = pdiv(g_l * p_l + g_r * p_r - (u_r - u_l), g_l + g_r);
        }
    }
    if (pstar <= p_l)
        S_l = u_l - c_l;
    else
        S_l = u_l - c_l * psqrt_nc(1.0 + K.sl() * (pdiv(pstar, p_l) - 1.0));
    if (pstar <= p_r)
        S_r = u_r + c_r;
    else
        S_r = u_r + c_r * psqrt_nc(1.0 + K.sr() * (pdiv(pstar, p_r) - 1.0));
}

// consFlux in the (normal, transverse) frame, riemann.py:1104-1179.
// State and flux components: d, E, mn (normal momentum), mt (transverse).
struct ConsN { double d, E, mn, mt; };

// with_p = false: SphericalPolar grids keep the pressure out of the momentum
// flux (riemann.py:1156, 1171)
__device__ __forceinline__ ConsN cons_flux_n(const ConsN &U, double gamma, bool normal_is_x,
                                             bool with_p = true)
{
    // u, v in the reference are x/y velocities; (u*u + v*v) is evaluated as
    // written there, i.e. x-velocity squared first
    double un = 0.0, ut = 0.0;
    if (U.d != 0.0) {
        const double rd = PYRO_FAST ? prcp(U.d) : 0.0;
        un = pdivr(U.mn, U.d, rd);
        ut = pdivr(U.mt, U.d, rd);
    }
    double ke = normal_is_x ? (un * un + ut * ut) : (ut * ut + un * un);
    double p = (U.E - 0.5 * U.d * ke) * (gamma - 1.0);
    ConsN F;
    F.d = U.d * un;
    F.mn = U.mn * un;
    if (with_p) F.mn += p;
    F.mt = U.mt * un;
    F.E = (U.E + p) * un;
    return F;
}

// HLLC flux for one face, riemann.py:681-860, in the (normal, transverse)
// frame.  normal_is_x only fixes the order of the two squares in the kinetic
// energy of consFlux.
// (normal velocity, transverse velocity, pressure) of a face state whose
// conserved form is also at hand
struct FaceQ { double un, ut, p; };

// HAVEQ (fast build only): the primitive face states ql / qr are given (the
// characteristic tracing produced them; prim_to_cons turned them into Ul / Ur),
// so velocities and pressure are not recovered from the conserved states again.
// Differs from the reference's round trip by rounding only.
#if PYRO_FAST
// fast build.  Same solver, written for the instruction count (differs from the
// reference's evaluation order by rounding only; tolerance-tested, north_star 1e-10):
//  * a_k = rho_k (S_k - u_k) is formed once and serves the contact speed
//    S_c = (p_r - p_l + a_l u_l - a_r u_r) / (a_l - a_r) and the star state;
//  * the star-region flux F_k + S_k (U*_k - U_k) of riemann.py:784-856 collapses, with
//    f = a_k / (S_k - S_c) (= rho*_k) and p* = p_k + a_k (S_c - u_k), to
//        F = (f S_c,  S_c (E* + p*),  f S_c^2 + p*,  f S_c ut_k),
//        E* = ((S_k - u_k) E_k + (S_c - u_k) (a_k S_c + p_k)) / (S_k - S_c)
//    (insert U*_k: the mass flux is rho u + S (f - rho) = f S - a = f S_c, and so on):
//    one reciprocal, no physical flux of the outer state, 17 instead of 35 instructions;
//  * PVRS pressure as one fma over (u_l - u_r) (rho_l + rho_r) (c_l + c_r) / 8.
template <class KT>
__device__ __forceinline__ void estimate_wave_speed_fast(double rho_l, double u_l, double p_l,
                                                         double c_l, double rho_r, double u_r,
                                                         double p_r, double c_r, const KT &K,
                                                         double &S_l, double &S_r)
{
    const double gamma = K.g();
    const double p_max = fmax(p_l, p_r), p_min = fmin(p_l, p_r);
    double pstar = fma(0.125 * (u_l - u_r), (rho_l + rho_r) * (c_l + c_r), 0.5 * (p_l + p_r));
    if (__builtin_expect(p_max > 2.0 * p_min && (pstar < p_min || pstar > p_max), 0)) {      // riemann.py:621-658
        if (pstar < p_min) {   // two-rarefaction, :626-638
            double z = pdiv(gamma - 1.0, 2.0 * gamma);
            double p_lr = pow(pdiv(p_l, p_r), z);
            double ustar = pdiv(pdiv(p_lr * u_l, c_l) + pdiv(u_r, c_r) +
                                    pdiv(2.0 * (p_lr - 1.0), gamma - 1.0),
                                pdiv(p_lr, c_l) + pdiv(1.0, c_r));
            pstar = 0.5 * (p_l * pow(1.0 + pdiv((gamma - 1.0) * (u_l - ustar), 2.0 * c_l),
                                     pdiv(1.0, z)) +
                           p_r * pow(1.0 + pdiv((gamma - 1.0) * (ustar - u_r), 2.0 * c_r),
                                     pdiv(1.0, z)));
        } else {               // two-shock, :640-658
            double A_r = pdiv(2.0, (gamma + 1.0) * rho_r);
            double B_r = p_r * (gamma - 1.0) * K.gp1();
            double A_l = pdiv(2.0, (gamma + 1.0) * rho_l);
            double B_l = p_l * (gamma - 1.0) * K.gp1();
            double p_guess = fmax(0.0, pstar);
            double g_l = psqrt_nc(pdiv(A_l, p_guess + B_l));
            double g_r = psqrt_nc(pdiv(A_r, p_guess + B_r));
            pstar = pdiv(g_l * p_l + g_r * p_r - (u_r - u_l), g_l + g_r);
        }
    }
    S_l = u_l - c_l;
    S_r = u_r + c_r;
    // the shock branch of a compressed side (riemann.py:660-678).  In smooth flow p* lies between p_l
    // and p_r: every lane has exactly ONE compressed side, but which one differs from lane to lane,
    // and a wavefront that evaluates the two branches one after the other pays two reciprocals and
    // two roots (quarter rate) for one per lane.  So: one evaluation for the lane's (first)
    // compressed side, and a second one only where both sides are compressed.  Same expressions on
    // the same operands: the same bits as the two branches.
    const bool hl = pstar > p_l, hr = pstar > p_r;
    if (hl || hr) {
        const double pk = hl ? p_l : p_r, ks = hl ? K.sl() : K.sr();
        const double z = psqrt_nc(fma(ks, pstar * prcp(pk) - 1.0, 1.0));
        if (hl) S_l = fma(-c_l, z, u_l);
        else S_r = fma(c_r, z, u_r);
        if (__builtin_expect(hl && hr, 0))
            S_r = fma(c_r, psqrt_nc(fma(K.sr(), pstar * prcp(p_r) - 1.0, 1.0)), u_r);
    }
}

template <bool HAVEQ, class KT>
__device__ __forceinline__ ConsN hllc_flux_impl(const ConsN &Ul, const ConsN &Ur, const FaceQ &ql,
                                                const FaceQ &qr, const KT &K, bool normal_is_x)
{
    (void)normal_is_x;
    const double gamma = K.g();
    const double smallc = 1.e-10, smallp = 1.e-10;
    const double rho_l = Ul.d, rho_r = Ur.d;
    double un_l, ut_l, pf_l, un_r, ut_r, pf_r;     // pf: pressure of the physical flux
    double p_l, p_r, c_l, c_r;
    if (HAVEQ) {
        // velocities and pressure given; c = sqrt(gamma p / rho) = gamma p / sqrt(gamma p rho):
        // one v_rsq_f64 per side, no reciprocal of the density (nothing else needs it)
        un_l = ql.un; ut_l = ql.ut; pf_l = ql.p;
        un_r = qr.un; ut_r = qr.ut; pf_r = qr.p;
        p_l = fmax(pf_l, smallp); p_r = fmax(pf_r, smallp);
        const double gp_l = gamma * p_l, gp_r = gamma * p_r;
        c_l = fmax(smallc, gp_l * prsqrt(gp_l * rho_l));
        c_r = fmax(smallc, gp_r * prsqrt(gp_r * rho_r));
    } else {
        const double ril = prcp(rho_l), rir = prcp(rho_r);
        un_l = Ul.mn * ril; ut_l = Ul.mt * ril;
        pf_l = fma(-0.5, fma(Ul.mt, ut_l, Ul.mn * un_l), Ul.E) * (gamma - 1.0);
        un_r = Ur.mn * rir; ut_r = Ur.mt * rir;
        pf_r = fma(-0.5, fma(Ur.mt, ut_r, Ur.mn * un_r), Ur.E) * (gamma - 1.0);
        p_l = fmax(pf_l, smallp); p_r = fmax(pf_r, smallp);
        c_l = fmax(smallc, psqrt_nc(gamma * p_l * ril));
        c_r = fmax(smallc, psqrt_nc(gamma * p_r * rir));
    }
    double S_l, S_r;
    estimate_wave_speed_fast(rho_l, un_l, p_l, c_l, rho_r, un_r, p_r, c_r, K, S_l, S_r);
    const double d_l = S_l - un_l, d_r = S_r - un_r;
    const double a_l = rho_l * d_l, a_r = rho_r * d_r;
    const double S_c = fma(-a_r, un_r, fma(a_l, un_l, p_r - p_l)) * prcp(a_l - a_r);
    // physical flux of an outer state (supersonic faces).  With the primitives given the
    // momenta of the conserved state are never touched (rho u, rho v are formed here, on the
    // rare path), so a caller that moves the state between lanes moves density and energy only
    auto outer = [&](const ConsN &U, double un, double ut, double pf) {
        const double mn = HAVEQ ? U.d * un : U.mn, mt = HAVEQ ? U.d * ut : U.mt;
        return ConsN{U.d * un, (U.E + pf) * un, fma(mn, un, pf), mt * un};
    };
    auto star = [&](double un, double ut, double p, double E, double S, double d, double a) {
        const double dS = S_c - un;
        const double ps = fma(a, dS, p);
        const double rb = prcp(S - S_c);
        const double f = a * rb;
        ConsN F;
        F.d = f * S_c;
        F.mn = fma(F.d, S_c, ps);
        F.mt = F.d * ut;
        const double Es = fma(dS, fma(a, S_c, p), d * E) * rb;
        F.E = S_c * (Es + ps);
        return F;
    };
    // riemann.py:784-856: S_r <= 0 | S_c <= 0 < S_r | S_l < 0 < S_c | else
    if (__builtin_expect(S_r <= 0.0, 0)) return outer(Ur, un_r, ut_r, pf_r);
    if (S_c <= 0.0) return star(un_r, ut_r, p_r, Ur.E, S_r, d_r, a_r);
    if (__builtin_expect(S_l < 0.0, 1)) return star(un_l, ut_l, p_l, Ul.E, S_l, d_l, a_l);
    return outer(Ul, un_l, ut_l, pf_l);
}
#else
template <bool HAVEQ, class KT>
__device__ __forceinline__ ConsN hllc_flux_impl(const ConsN &Ul, const ConsN &Ur, const FaceQ &ql,
                                                const FaceQ &qr, const KT &K, bool normal_is_x)
{
    const double gamma = K.g();
    const double smallc = 1.e-10, smallp = 1.e-10;
    double rho_l = Ul.d;
    const double ril = PYRO_FAST ? prcp(rho_l) : 0.0;
    double un_l, ut_l, p_l;
    if (HAVEQ) {
        un_l = ql.un; ut_l = ql.ut; p_l = ql.p;
    } else {
        un_l = pdivr(Ul.mn, rho_l, ril);
        ut_l = pdivr(Ul.mt, rho_l, ril);
        double rhoe_l = Ul.E - 0.5 * rho_l * (un_l * un_l + ut_l * ut_l);
        p_l = rhoe_l * (gamma - 1.0);
    }
    const double pf_l = p_l;            // pressure of the physical flux (HAVEQ)
    p_l = fmax(p_l, smallp);
    double rho_r = Ur.d;
    const double rir = PYRO_FAST ? prcp(rho_r) : 0.0;
    double un_r, ut_r, p_r;
    if (HAVEQ) {
        un_r = qr.un; ut_r = qr.ut; p_r = qr.p;
    } else {
        un_r = pdivr(Ur.mn, rho_r, rir);
        ut_r = pdivr(Ur.mt, rho_r, rir);
        double rhoe_r = Ur.E - 0.5 * rho_r * (un_r * un_r + ut_r * ut_r);
        p_r = rhoe_r * (gamma - 1.0);
    }
    const double pf_r = p_r;
    p_r = fmax(p_r, smallp);
    double c_l = fmax(smallc, psqrt_nc(pdivr(gamma * p_l, rho_l, ril)));
    double c_r = fmax(smallc, psqrt_nc(pdivr(gamma * p_r, rho_r, rir)));
    double S_l, S_r;
    estimate_wave_speed(rho_l, un_l, p_l, c_l, rho_r, un_r, p_r, c_r, K, S_l, S_r);
    double S_c = pdiv(p_r - p_l + rho_l * un_l * (S_l - un_l) - rho_r * un_r * (S_r - un_r),
                      rho_l * (S_l - un_l) - rho_r * (S_r - un_r));
    // physical flux of one side: consFlux of the conserved state, or directly
    // from the given primitives
    auto side_flux = [&](const ConsN &U, double un, double pf) {
        if (!HAVEQ) return cons_flux_n(U, gamma, normal_is_x);
        ConsN F;
        F.d = U.d * un;
        F.mn = U.mn * un + pf;
        F.mt = U.mt * un;
        F.E = (U.E + pf) * un;
        return F;
    };
    ConsN F;
    if (S_r <= 0.0) {
        F = side_flux(Ur, un_r, pf_r);
    } else if (S_c <= 0.0 && 0.0 < S_r) {
#if PYRO_FAST && !defined(PYRO_EMU)
        const double a = rho_r * (S_r - un_r), b = S_r - S_c;
        const double w = prcp(a * b);          // a/b = a*a*w,  p_r/a = p_r*b*w
        const double f = a * a * w, pa = p_r * b * w;
#else
        double f = pdiv(rho_r * (S_r - un_r), S_r - S_c);
        const double pa = pdiv(p_r, rho_r * (S_r - un_r));
#endif
        ConsN Us;
        Us.d = f;
        Us.mn = f * S_c;
        Us.mt = f * ut_r;
        Us.E = f * (pdivr(Ur.E, rho_r, rir) + (S_c - un_r) * (S_c + pa));
        F = side_flux(Ur, un_r, pf_r);
        F.d = F.d + S_r * (Us.d - Ur.d);
        F.mn = F.mn + S_r * (Us.mn - Ur.mn);
        F.mt = F.mt + S_r * (Us.mt - Ur.mt);
        F.E = F.E + S_r * (Us.E - Ur.E);
    } else if (S_l < 0.0 && 0.0 < S_c) {
#if PYRO_FAST && !defined(PYRO_EMU)
        const double a = rho_l * (S_l - un_l), b = S_l - S_c;
        const double w = prcp(a * b);
        const double f = a * a * w, pa = p_l * b * w;
#else
        double f = pdiv(rho_l * (S_l - un_l), S_l - S_c);
        const double pa = pdiv(p_l, rho_l * (S_l - un_l));
#endif
        ConsN Us;
        Us.d = f;
        Us.mn = f * S_c;
        Us.mt = f * ut_l;
        Us.E = f * (pdivr(Ul.E, rho_l, ril) + (S_c - un_l) * (S_c + pa));
        F = side_flux(Ul, un_l, pf_l);
        F.d = F.d + S_l * (Us.d - Ul.d);
        F.mn = F.mn + S_l * (Us.mn - Ul.mn);
        F.mt = F.mt + S_l * (Us.mt - Ul.mt);
        F.E = F.E + S_l * (Us.E - Ul.E);
    } else {
        F = side_flux(Ul, un_l, pf_l);
    }
    return F;
}
#endif
template <class KT>
__device__ __forceinline__ ConsN hllc_flux(const ConsN &Ul, const ConsN &Ur, const KT &K,
                                           bool normal_is_x)
{
    const FaceQ none{0.0, 0.0, 0.0};
    return hllc_flux_impl<false, KT>(Ul, Ur, none, none, K, normal_is_x);
}


// Low-Mach HLLC variant ("HLLC_lm", riemann_hllc_lowspeed, riemann.py:863-1020):
// the star-region pressure is blended with phi = chi (2 - chi),
// chi = min(1, max|v| / max c), and the star fluxes are written in the
// (S_c (S U - F) + S p* D) / (S - S_c) form with D = (0, S_c, 1, 0) in
// (density, energy, normal momentum, transverse momentum).
template <class KT>
__device__ __forceinline__ ConsN hllc_lm_flux(const ConsN &Ul, const ConsN &Ur, const KT &K,
                                              bool normal_is_x)
{
    const double gamma = K.g();
    const double smallc = 1.e-10, smallp = 1.e-10;
    const double rho_l = Ul.d;
    const double ril = PYRO_FAST ? prcp(rho_l) : 0.0;
    const double un_l = pdivr(Ul.mn, rho_l, ril), ut_l = pdivr(Ul.mt, rho_l, ril);
    const double rhoe_l = Ul.E - 0.5 * rho_l * (un_l * un_l + ut_l * ut_l);
    const double p_l = fmax(rhoe_l * (gamma - 1.0), smallp);
    const double rho_r = Ur.d;
    const double rir = PYRO_FAST ? prcp(rho_r) : 0.0;
    const double un_r = pdivr(Ur.mn, rho_r, rir), ut_r = pdivr(Ur.mt, rho_r, rir);
    const double rhoe_r = Ur.E - 0.5 * rho_r * (un_r * un_r + ut_r * ut_r);
    const double p_r = fmax(rhoe_r * (gamma - 1.0), smallp);
    const double c_l = fmax(smallc, psqrt(pdivr(gamma * p_l, rho_l, ril)));
    const double c_r = fmax(smallc, psqrt(pdivr(gamma * p_r, rho_r, rir)));
    double S_l, S_r;
    estimate_wave_speed(rho_l, un_l, p_l, c_l, rho_r, un_r, p_r, c_r, K, S_l, S_r);
    const double S_c = pdiv(p_r - p_l + rho_l * un_l * (S_l - un_l) - rho_r * un_r * (S_r - un_r),
                            rho_l * (S_l - un_l) - rho_r * (S_r - un_r));
    const double vmag_l = psqrt0(un_l * un_l + ut_l * ut_l);
    const double vmag_r = psqrt0(un_r * un_r + ut_r * ut_r);
    const double cs_max = fmax(c_l, c_r);
    const double chi = fmin(1.0, pdiv(fmax(vmag_l, vmag_r), cs_max));
    const double phi = chi * (2.0 - chi);
    const double pstar = 0.5 * (p_l + p_r) + 0.5 * phi * (rho_l * (S_l - un_l) * (S_c - un_l) +
                                                          rho_r * (S_r - un_r) * (S_c - un_r));
    if (S_r <= 0.0) return cons_flux_n(Ur, gamma, normal_is_x);
    if (S_c <= 0.0 && 0.0 < S_r) {
        const ConsN Fr = cons_flux_n(Ur, gamma, normal_is_x);
        const double den = S_r - S_c, sp = S_r * pstar;
        return ConsN{pdiv(S_c * (S_r * Ur.d - Fr.d) + sp * 0.0, den),
                     pdiv(S_c * (S_r * Ur.E - Fr.E) + sp * S_c, den),
                     pdiv(S_c * (S_r * Ur.mn - Fr.mn) + sp * 1.0, den),
                     pdiv(S_c * (S_r * Ur.mt - Fr.mt) + sp * 0.0, den)};
    }
    if (S_l < 0.0 && 0.0 < S_c) {
        const ConsN Fl = cons_flux_n(Ul, gamma, normal_is_x);
        const double den = S_l - S_c, sp = S_l * pstar;
        return ConsN{pdiv(S_c * (S_l * Ul.d - Fl.d) + sp * 0.0, den),
                     pdiv(S_c * (S_l * Ul.E - Fl.E) + sp * S_c, den),
                     pdiv(S_c * (S_l * Ul.mn - Fl.mn) + sp * 1.0, den),
                     pdiv(S_c * (S_l * Ul.mt - Fl.mt) + sp * 0.0, den)};
    }
    return cons_flux_n(Ul, gamma, normal_is_x);
}

// Two-shock Colella-Glaz-Ferguson solver on one face, riemann.py:8-310,
// followed by consFlux of the resulting interface state (riemann_flux
// :1083-1090), in the (normal, transverse) frame.  wall_zero: this is the
// lower boundary face of a solid wall (riemann.py:274-286; the upper-wall
// test of the reference can never fire).
__device__ __forceinline__ ConsN cgf_state(const ConsN &Ul, const ConsN &Ur, double gamma,
                                           bool wall_zero)
{
    const double smallc = 1.e-10, smallrho = 1.e-10, smallp = 1.e-10;
    const double rho_l = Ul.d;
    const double ril = PYRO_FAST ? prcp(rho_l) : 0.0;
    const double un_l = pdivr(Ul.mn, rho_l, ril), ut_l = pdivr(Ul.mt, rho_l, ril);
    const double rhoe_l = Ul.E - 0.5 * rho_l * (un_l * un_l + ut_l * ut_l);
    const double p_l = fmax(rhoe_l * (gamma - 1.0), smallp);
    const double rho_r = Ur.d;
    const double rir = PYRO_FAST ? prcp(rho_r) : 0.0;
    const double un_r = pdivr(Ur.mn, rho_r, rir), ut_r = pdivr(Ur.mt, rho_r, rir);
    const double rhoe_r = Ur.E - 0.5 * rho_r * (un_r * un_r + ut_r * ut_r);
    const double p_r = fmax(rhoe_r * (gamma - 1.0), smallp);
    const double W_l = fmax(smallrho * smallc, psqrt(gamma * p_l * rho_l));
    const double W_r = fmax(smallrho * smallc, psqrt(gamma * p_r * rho_r));
    const double c_l = fmax(smallc, psqrt(pdivr(gamma * p_l, rho_l, ril)));
    const double c_r = fmax(smallc, psqrt(pdivr(gamma * p_r, rho_r, rir)));
    const double rW = PYRO_FAST ? prcp(W_l + W_r) : 0.0;
    double pstar = pdivr(W_l * p_r + W_r * p_l + W_l * W_r * (un_l - un_r), W_l + W_r, rW);
    pstar = fmax(pstar, smallp);
    const double ustar = pdivr(W_l * un_l + W_r * un_r + (p_l - p_r), W_l + W_r, rW);
    const double rhostar_l = rho_l + pdiv(pstar - p_l, c_l * c_l);
    const double rhostar_r = rho_r + pdiv(pstar - p_r, c_r * c_r);
    const double rhoestar_l =
        rhoe_l + pdiv((pstar - p_l) * (pdivr(rhoe_l, rho_l, ril) + pdivr(p_l, rho_l, ril)), c_l * c_l);
    const double rhoestar_r =
        rhoe_r + pdiv((pstar - p_r) * (pdivr(rhoe_r, rho_r, rir) + pdivr(p_r, rho_r, rir)), c_r * c_r);
    double rho_s, un_s, ut_s, rhoe_s;
    if (ustar > 0.0) {
        ut_s = ut_l;
        const double cstar_l = fmax(smallc, psqrt(pdiv(gamma * pstar, rhostar_l)));
        const double lambda_l = un_l - c_l, lambdastar_l = ustar - cstar_l;
        if (pstar > p_l) {
            const double sigma = (lambda_l + lambdastar_l) / 2.0;
            if (sigma > 0.0) { rho_s = rho_l; un_s = un_l; rhoe_s = rhoe_l; }
            else { rho_s = rhostar_l; un_s = ustar; rhoe_s = rhoestar_l; }
        } else if (lambda_l < 0.0 && lambdastar_l < 0.0) {
            rho_s = rhostar_l; un_s = ustar; rhoe_s = rhoestar_l;
        } else if (lambda_l > 0.0 && lambdastar_l > 0.0) {
            rho_s = rho_l; un_s = un_l; rhoe_s = rhoe_l;
        } else {
            const double alpha = pdiv(lambda_l, lambda_l - lambdastar_l);
            rho_s = alpha * rhostar_l + (1.0 - alpha) * rho_l;
            un_s = alpha * ustar + (1.0 - alpha) * un_l;
            rhoe_s = alpha * rhoestar_l + (1.0 - alpha) * rhoe_l;
        }
    } else if (ustar < 0) {
        ut_s = ut_r;
        const double cstar_r = fmax(smallc, psqrt(pdiv(gamma * pstar, rhostar_r)));
        const double lambda_r = un_r + c_r, lambdastar_r = ustar + cstar_r;
        if (pstar > p_r) {
            const double sigma = (lambda_r + lambdastar_r) / 2.0;
            if (sigma > 0.0) { rho_s = rhostar_r; un_s = ustar; rhoe_s = rhoestar_r; }
            else { rho_s = rho_r; un_s = un_r; rhoe_s = rhoe_r; }
        } else if (lambda_r < 0.0 && lambdastar_r < 0.0) {
            rho_s = rho_r; un_s = un_r; rhoe_s = rhoe_r;
        } else if (lambda_r > 0.0 && lambdastar_r > 0.0) {
            rho_s = rhostar_r; un_s = ustar; rhoe_s = rhoestar_r;
        } else {
            const double alpha = pdiv(lambda_r, lambda_r - lambdastar_r);
            rho_s = alpha * rhostar_r + (1.0 - alpha) * rho_r;
            un_s = alpha * ustar + (1.0 - alpha) * un_r;
            rhoe_s = alpha * rhoestar_r + (1.0 - alpha) * rhoe_r;
        }
    } else {   // ustar == 0
        rho_s = 0.5 * (rhostar_l + rhostar_r);
        un_s = ustar;
        ut_s = 0.5 * (ut_l + ut_r);
        rhoe_s = 0.5 * (rhoestar_l + rhoestar_r);
    }
    if (wall_zero) un_s = 0.0;
    ConsN Uo;
    Uo.d = rho_s;
    Uo.mn = rho_s * un_s;
    Uo.mt = rho_s * ut_s;
    Uo.E = rhoe_s + 0.5 * rho_s * (un_s * un_s + ut_s * ut_s);
    return Uo;
}
__device__ __forceinline__ ConsN cgf_flux(const ConsN &Ul, const ConsN &Ur, double gamma,
                                          bool normal_is_x, bool wall_zero)
{
    return cons_flux_n(cgf_state(Ul, Ur, gamma, wall_zero), gamma, normal_is_x);
}

// compressible.riemann dispatch: SOLVER 0 = HLLC, 1 = CGF, 2 = HLLC_lm
template <int SOLVER, class KT>
__device__ __forceinline__ ConsN riemann_face(const ConsN &Ul, const ConsN &Ur, const KT &K,
                                              bool normal_is_x, bool wall_zero)
{
    if (SOLVER == 1) return cgf_flux(Ul, Ur, K.g(), normal_is_x, wall_zero);
    if (SOLVER == 2) return hllc_lm_flux(Ul, Ur, K, normal_is_x);
    return hllc_flux(Ul, Ur, K, normal_is_x);
}
template <int SOLVER>
__device__ __forceinline__ ConsN riemann_face(const ConsN &Ul, const ConsN &Ur, double gamma,
                                              bool normal_is_x, bool wall_zero)
{
    return riemann_face<SOLVER>(Ul, Ur, make_gask(gamma), normal_is_x, wall_zero);
}

// Sponge, compressible/simulation.py:164-184 and :427-441 (acts on every
// cell of the array, ghost cells included)
__device__ __forceinline__ void sponge_cell(Cons &U, double dt, double rho_begin, double rho_full,
                                            double tau)
{
    const double PI = 3.14159265358979323846;
    const double rho = U.d;
    double f;
    if (rho > rho_begin) f = 0.0;
    else if (rho < rho_full) f = 1.0;
    else f = 0.5 * (1.0 - cos(PI * (rho - rho_begin) / (rho_full - rho_begin)));
    const double kappa = f / tau;
    const double xo = U.mx, yo = U.my;
    U.mx = xo / (1.0 + dt * kappa);
    U.my = yo / (1.0 + dt * kappa);
    const double dke = 0.5 * ((U.mx * U.mx + U.my * U.my) - (xo * xo + yo * yo)) / U.d;
    U.E += dke;
}

// Gravity (Cartesian, acting in y): S[ymom] = rho g, S[E] = ymom g
// (compressible/simulation.py:131-134).
//
// Interface states: every one of the cell's four face states receives
// +(dt/2) S(cell) on its momentum and energy components
// (apply_source_terms, unsplit_fluxes.py:308-326; XP(i) = U_xl[i+1] gets
// S[i], etc.).  The reference ghost-fills S with the BCs of ymom_src
// (odd across a reflecting y wall) and E_src (even), which for a ghost cell
// beyond such a wall is minus the value computed from the ghost state itself:
// pass sgn = -1 there, +1 everywhere else.
// hrate / hprof: problem source S[E] += rho * e_rate * profile of the heating /
// plume / convection problems (simulation.py:156-159); its ghost values are
// the even BC fill of the interior ones, i.e. the profile plane is ghost-filled
// and no sign applies.  0 / 0 when there is no such source.
__device__ __forceinline__ void add_grav_to_state(Cons &S, const Cons &Ucell, double grav,
                                                  double dt, double sgn, double hrate = 0.0,
                                                  double hprof = 0.0)
{
    const double Sy = sgn * (Ucell.d * grav);
    const double SE = sgn * (Ucell.my * grav) + Ucell.d * hrate * hprof;
    S.my += 0.5 * dt * Sy;
    S.E += 0.5 * dt * SE;
}

// Source predictor-corrector after the conservative update
// (compressible/simulation.py:406-423 with get_external_sources :105-161):
// U* = U + dt S(U_old); S_new uses the time-centred momentum; U = U* + dt/2 (S_new - S_old)
__device__ __forceinline__ void grav_update(Cons &U, const Cons &Uold, double grav, double dt,
                                            double hrate = 0.0, double hprof = 0.0)
{
    const double Sy_old = Uold.d * grav;
    const double SE_old = Uold.my * grav + Uold.d * hrate * hprof;
    U.my = U.my + dt * Sy_old;
    U.E = U.E + dt * SE_old;
    const double Sy_new = U.d * grav;
    const double ymom_new = U.my + 0.5 * dt * (Sy_new - Sy_old);
    const double SE_new = ymom_new * grav + U.d * hrate * hprof;
    U.my = U.my + 0.5 * dt * (Sy_new - Sy_old);
    U.E = U.E + 0.5 * dt * (SE_new - SE_old);
}

// vertex-centred velocity divergence at (i-1/2, j-1/2),
// compressible/interface.py:312-330 (Cartesian branch)
__device__ __forceinline__ double div_u_vertex(double u_ij, double u_ijm, double u_imj,
                                               double u_imjm, double v_ij, double v_imj,
                                               double v_ijm, double v_imjm, double dx, double dy)
{
    double ur = 0.5 * (u_ij + u_ijm);
    double ul = 0.5 * (u_imj + u_imjm);
    double vt = 0.5 * (v_ij + v_imj);
    double vb = 0.5 * (v_ijm + v_imjm);
    double ux = pdiv(ur - ul, dx);
    double vy = pdiv(vt - vb, dy);
    return ux + vy;
}

// div_u_vertex with the reciprocals of dx, dy passed in (fast build)
__device__ __forceinline__ double div_u_vertex_r(double u_ij, double u_ijm, double u_imj,
                                                 double u_imjm, double v_ij, double v_imj,
                                                 double v_ijm, double v_imjm, double dx, double dy,
                                                 double rdx, double rdy)
{
    double ur = 0.5 * (u_ij + u_ijm);
    double ul = 0.5 * (u_imj + u_imjm);
    double vt = 0.5 * (v_ij + v_imj);
    double vb = 0.5 * (v_ijm + v_imjm);
    double ux = pdivr(ur - ul, dx, rdx);
    double vy = pdivr(vt - vb, dy, rdy);
    return ux + vy;
}

}  // inline namespace hydro_fast / hydro_exact
}  // namespace pyro
