// Compressible Euler, unsplit CTU (Colella 1990) step with HLLC fluxes.
//
// Replaces (reference file:line)
//   pyro/compressible/simulation.py:267-288   method_compute_timestep
//   pyro/compressible/simulation.py:290-450   Simulation.evolve
//   pyro/compressible/unsplit_fluxes.py:134-549
//   pyro/compressible/interface.py:5-378      states, artificial_viscosity
//   pyro/compressible/riemann.py:596-860,1104-1179  HLLC
//   pyro/mesh/reconstruction.py:9-183         limiters, flattening
//
// kernel_set 0 ("staged"): one kernel per algorithm stage with planar global
// intermediates; every stage can be dumped (pyrohip_comp_stage_dump) and
// compared with the oracle.  Only the cells/faces inside the interior's
// domain of dependence are computed (SURVEY.md 7 "stencil radius 4").
//
// This file is compiled twice: PYRO_FAST=0 (-ffp-contract=off, the
// bit-faithful build) and PYRO_FAST=1 (-ffp-contract=fast); the entry points
// dispatch on pyrohip_comp_params.fast_math.
#include "common.h"
#include "hydro.h"
#include "reduce.h"

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

// work-space plane indices
enum {
    W_Q = 0,      // 4: rho u v p
    W_XI = 4,     // 1
    W_XM = 5,     // 4: lower x-face state of the cell  (= U_xr[i,j])
    W_XP = 9,     // 4: upper x-face state of the cell  (= U_xl[i+1,j])
    W_YM = 13,    // 4
    W_YP = 17,    // 4
    W_FXT = 21,   // 4: transverse Riemann flux on x faces
    W_FYT = 25,   // 4
    W_FX = 29,    // 4: final fluxes incl. artificial viscosity
    W_FY = 33,    // 4
    W_NPLANES = 37
};

struct CP {   // kernel-side parameters
    double gamma, dx, dy, dt;
    double z0, z1, delta, cvisc, small_dens;
    int limiter, use_flattening;
    int avx_hi, avy_hi;   // compute avisc on the upper boundary face
    double grav;          // compressible.grav (0: no source terms)
    int refl_ylo, refl_yhi;   // y-momentum reflects oddly at the lower / upper y wall
    int amb_yhi;              // "ambient" boundary on the upper y side
    int have_src;             // gravity and / or a heating source
    double heat_rate;         // S[E] += rho * heat_rate * heat[i,j] (ghost-filled plane)
    const double *heat;
    const double *ext;        // host-evaluated source S_h(U^n), 4 ghost-filled planes (or NULL):
                              // the step then ends with the predictor U* = U + dt S(U^n)
    int riemann, solid_xl, solid_yl;   // 0 HLLC / 1 CGF / 2 HLLC_lm; CGF wall rule
};

__device__ __forceinline__ ConsN riemann_rt(const ConsN &Ul, const ConsN &Ur, const CP &P, bool x,
                                            bool wall)
{
    if (P.riemann == 2) return riemann_face<2>(Ul, Ur, P.gamma, x, wall);
    return P.riemann == 1 ? riemann_face<1>(Ul, Ur, P.gamma, x, wall)
                          : riemann_face<0>(Ul, Ur, P.gamma, x, wall);
}

__device__ __forceinline__ Cons load_cons(const double *__restrict__ a, size_t plane, size_t k)
{
    Cons U;
    U.d = a[k]; U.E = a[plane + k]; U.mx = a[2 * plane + k]; U.my = a[3 * plane + k];
    return U;
}
__device__ __forceinline__ void store_cons(double *__restrict__ a, size_t plane, size_t k,
                                           const Cons &U)
{
    a[k] = U.d; a[plane + k] = U.E; a[2 * plane + k] = U.mx; a[3 * plane + k] = U.my;
}

// ---- stage 0: clean_state + cons_to_prim over the whole array ------------
// simulation.py:452-456 and :49-80.  flag[0] |= 1 when the interior
// positivity assert (:68-71) would fire.
__global__ __launch_bounds__(256) void k_prim(double *__restrict__ U, double *__restrict__ W,
                                              Geom g, CP P, int *__restrict__ flag,
                                              int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = bx * blockDim.x + threadIdx.x;
    const int i = by;
    if (j >= g.qy) return;
    const size_t k = (size_t)i * g.pitch + j;
    const bool interior = (i >= g.ilo && i <= g.ihi && j >= g.jlo && j <= g.jhi);
    Cons Uc = load_cons(U, g.plane, k);
    if (interior) {
        double dn = fmax(Uc.d, P.small_dens);
        if (dn != Uc.d) U[k] = dn;
        Uc.d = dn;
    }
    bool ok;
    Prim q = cons_to_prim(Uc, P.gamma, &ok);
    double *Q = W + (size_t)W_Q * g.plane;
    Q[k] = q.r; Q[g.plane + k] = q.u; Q[2 * g.plane + k] = q.v; Q[3 * g.plane + k] = q.p;
    if (interior && !ok) atomicOr(flag, 1);
}

// ---- stage 1: multi-dimensional flattening coefficient on R(1) -----------
// reconstruction.py:123-183
__global__ __launch_bounds__(256) void k_xi(const double *__restrict__ W_, double *__restrict__ XI,
                                            Geom g, CP P, int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo - 1 + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    if (!P.use_flattening) { XI[k] = 1.0; return; }
    const double *u = W_ + (size_t)(W_Q + 1) * g.plane;
    const double *v = W_ + (size_t)(W_Q + 2) * g.plane;
    const double *pr = W_ + (size_t)(W_Q + 3) * g.plane;
    // own coefficient and the one of the UPWIND neighbour (w.r.t. the pressure
    // gradient) in each direction; the downwind one is never selected
    const ptrdiff_t sx = (pr[k + p] - pr[k - p] > 0) ? -(ptrdiff_t)p : (ptrdiff_t)p;
    const ptrdiff_t sy = (pr[k + 1] - pr[k - 1] > 0) ? -1 : 1;
    const size_t cx = k + sx, cy = k + sy;
    const double xix = flatten_1d(pr[k - 2 * p], pr[k - p], pr[k + p], pr[k + 2 * p], u[k - p],
                                  u[k + p], P.z0, P.z1, P.delta);
    const double px = flatten_1d(pr[cx - 2 * p], pr[cx - p], pr[cx + p], pr[cx + 2 * p],
                                 u[cx - p], u[cx + p], P.z0, P.z1, P.delta);
    const double xiy = flatten_1d(pr[k - 2], pr[k - 1], pr[k + 1], pr[k + 2], v[k - 1], v[k + 1],
                                  P.z0, P.z1, P.delta);
    const double py = flatten_1d(pr[cy - 2], pr[cy - 1], pr[cy + 1], pr[cy + 2], v[cy - 1],
                                 v[cy + 1], P.z0, P.z1, P.delta);
    XI[k] = fmin(fmin(xix, px), fmin(xiy, py));
}

// ---- stage 2: limited slopes + characteristic tracing on R(1) ------------
// unsplit_fluxes.py:186-242, interface.py:5-236, simulation.py:83-102
__global__ __launch_bounds__(256) void k_states(const double *__restrict__ U,
                                                const double *__restrict__ W_,
                                                double *__restrict__ Wout, Geom g, CP P,
                                                int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo - 1 + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    const double *Q = W_ + (size_t)W_Q * pl;
    const double xi = W_[(size_t)W_XI * pl + k];
    double q0[4], dqx[4], dqy[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *a = Q + (size_t)n * pl;
        q0[n] = a[k];
        dqx[n] = xi * limited_slope(a[k - 2 * p], a[k - p], a[k], a[k + p], a[k + 2 * p],
                                    P.limiter);
        dqy[n] = xi * limited_slope(a[k - 2], a[k - 1], a[k], a[k + 1], a[k + 2], P.limiter);
    }
    Trace lo, hi;
    // x: normal velocity u (1), transverse v (2)
    trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], P.gamma,
                 P.dt / P.dx, lo, hi);
    Cons XM = prim_to_cons(Prim{lo.r, lo.un, lo.ut, lo.p}, P.gamma);
    Cons XP = prim_to_cons(Prim{hi.r, hi.un, hi.ut, hi.p}, P.gamma);
    // y: normal velocity v, transverse u
    trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], P.gamma,
                 P.dt / P.dy, lo, hi);
    Cons YM = prim_to_cons(Prim{lo.r, lo.ut, lo.un, lo.p}, P.gamma);
    Cons YP = prim_to_cons(Prim{hi.r, hi.ut, hi.un, hi.p}, P.gamma);
    if (P.have_src) {   // apply_source_terms, unsplit_fluxes.py:247-330
        // "ambient" upper boundary: the source ghosts are copies of row jhi
        // (BC.py:159-160), not the sources of the ambient ghost state
        const Cons Uc = load_cons(U, pl, (P.amb_yhi && j > g.jhi) ? k - (j - g.jhi) : k);
        const double sgn = ((j < g.jlo && P.refl_ylo) || (j > g.jhi && P.refl_yhi)) ? -1.0 : 1.0;
        const double hp = P.heat ? P.heat[k] : 0.0;
        add_grav_to_state(XM, Uc, P.grav, P.dt, sgn, P.heat_rate, hp);
        add_grav_to_state(XP, Uc, P.grav, P.dt, sgn, P.heat_rate, hp);
        add_grav_to_state(YM, Uc, P.grav, P.dt, sgn, P.heat_rate, hp);
        add_grav_to_state(YP, Uc, P.grav, P.dt, sgn, P.heat_rate, hp);
    }
    if (P.ext) {   // S = gravity + S_heating (simulation.py:157-159), ghost-filled as
                   // aux data (unsplit_fluxes.py:298-306); gravity alone when have_src
        const Cons X = load_cons(P.ext, pl, k);
        Cons *F[4] = {&XM, &XP, &YM, &YP};
        const Cons Uc = load_cons(U, pl, (P.amb_yhi && j > g.jhi) ? k - (j - g.jhi) : k);
        const double sgn = ((j < g.jlo && P.refl_ylo) || (j > g.jhi && P.refl_yhi)) ? -1.0 : 1.0;
        const double Sy = sgn * (Uc.d * P.grav) + X.my;
        const double SE = sgn * (Uc.my * P.grav) + X.E;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            F[f]->mx += 0.5 * P.dt * X.mx;
            F[f]->my += 0.5 * P.dt * Sy;
            F[f]->E += 0.5 * P.dt * SE;
        }
    }
    store_cons(Wout + (size_t)W_XM * pl, pl, k, XM);
    store_cons(Wout + (size_t)W_XP * pl, pl, k, XP);
    store_cons(Wout + (size_t)W_YM * pl, pl, k, YM);
    store_cons(Wout + (size_t)W_YP * pl, pl, k, YP);
}

__device__ __forceinline__ ConsN to_n(const Cons &U, bool x)
{
    return x ? ConsN{U.d, U.E, U.mx, U.my} : ConsN{U.d, U.E, U.my, U.mx};
}
__device__ __forceinline__ Cons from_n(const ConsN &F, bool x)
{
    return x ? Cons{F.d, F.E, F.mn, F.mt} : Cons{F.d, F.E, F.mt, F.mn};
}

// ---- stage 3: transverse Riemann problems --------------------------------
// unsplit_fluxes.py:412-440 -> riemann.py:681-860
// thread (i,j) in [ilo-1, ihi+1] x [jlo-1, jhi+1] solves its lower x face
// (needs i >= ilo) and its lower y face (needs j >= jlo)
__global__ __launch_bounds__(256) void k_riemann_t(const double *__restrict__ W_,
                                                   double *__restrict__ Wout, Geom g, CP P,
                                                   int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo - 1 + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    if (i >= g.ilo) {
        Cons Ul = load_cons(W_ + (size_t)W_XP * pl, pl, k - p);
        Cons Ur = load_cons(W_ + (size_t)W_XM * pl, pl, k);
        ConsN F = riemann_rt(to_n(Ul, true), to_n(Ur, true), P, true, P.solid_xl && i == g.ilo);
        store_cons(Wout + (size_t)W_FXT * pl, pl, k, from_n(F, true));
    }
    if (j >= g.jlo) {
        Cons Ul = load_cons(W_ + (size_t)W_YP * pl, pl, k - 1);
        Cons Ur = load_cons(W_ + (size_t)W_YM * pl, pl, k);
        ConsN F = riemann_rt(to_n(Ul, false), to_n(Ur, false), P, false, P.solid_yl && j == g.jlo);
        store_cons(Wout + (size_t)W_FYT * pl, pl, k, from_n(F, false));
    }
}

__device__ __forceinline__ Cons corrected(const Cons &U, const Cons &Fhi, const Cons &Flo,
                                          double hdtV, double A)
{
    // U += -hdtV*(F_hi*A - F_lo*A), unsplit_fluxes.py:447-471
    Cons r;
    r.d = U.d + (-hdtV * (Fhi.d * A - Flo.d * A));
    r.E = U.E + (-hdtV * (Fhi.E * A - Flo.E * A));
    r.mx = U.mx + (-hdtV * (Fhi.mx * A - Flo.mx * A));
    r.my = U.my + (-hdtV * (Fhi.my * A - Flo.my * A));
    return r;
}

// ---- stage 4: transverse correction + final Riemann + art. viscosity ----
// unsplit_fluxes.py:442-471, simulation.py:349-365, interface.py:239-378,
// unsplit_fluxes.py:525-547.
// thread (i,j) in [ilo, ihi+1] x [jlo, jhi+1]: F_x on its lower x face when
// j <= jhi, F_y on its lower y face when i <= ihi.
__global__ __launch_bounds__(256) void k_final(const double *__restrict__ U,
                                               const double *__restrict__ W_,
                                               double *__restrict__ Wout, Geom g, CP P,
                                               int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    const double hdt = 0.5 * P.dt;
    const double hdtV = hdt / (P.dx * P.dy);   // hdt / V, patch.py:232
    const double Ax = P.dy, Ay = P.dx;         // patch.py:219-222
    const double *FXT = W_ + (size_t)W_FXT * pl, *FYT = W_ + (size_t)W_FYT * pl;
    const double *u = W_ + (size_t)(W_Q + 1) * pl, *v = W_ + (size_t)(W_Q + 2) * pl;

    // vertex divergences needed by this thread: divU[i,j], [i,j+1], [i+1,j]
    const double d00 = div_u_vertex(u[k], u[k - 1], u[k - p], u[k - p - 1], v[k], v[k - p],
                                    v[k - 1], v[k - p - 1], P.dx, P.dy);
    const Cons Uc = load_cons(U, pl, k);

    if (j <= g.jhi) {   // x face (i,j)
        Cons Uxl = corrected(load_cons(W_ + (size_t)W_XP * pl, pl, k - p),
                             load_cons(FYT, pl, k - p + 1), load_cons(FYT, pl, k - p), hdtV, Ay);
        Cons Uxr = corrected(load_cons(W_ + (size_t)W_XM * pl, pl, k), load_cons(FYT, pl, k + 1),
                             load_cons(FYT, pl, k), hdtV, Ay);
        Cons F = from_n(riemann_rt(to_n(Uxl, true), to_n(Uxr, true), P, true, P.solid_xl && i == g.ilo), true);
        double avx = 0.0;
        if (i <= g.ihi || P.avx_hi) {
            size_t kk = k + 1;
            double d01 = div_u_vertex(u[kk], u[kk - 1], u[kk - p], u[kk - p - 1], v[kk],
                                      v[kk - p], v[kk - 1], v[kk - p - 1], P.dx, P.dy);
            double divU_x = 0.5 * (d00 + d01);
            avx = P.cvisc * fmax(-divU_x * P.dx, 0.0);
        }
        const Cons Um = load_cons(U, pl, k - p);
        F.d += avx * (Um.d - Uc.d);
        F.E += avx * (Um.E - Uc.E);
        F.mx += avx * (Um.mx - Uc.mx);
        F.my += avx * (Um.my - Uc.my);
        store_cons(Wout + (size_t)W_FX * pl, pl, k, F);
    }
    if (i <= g.ihi) {   // y face (i,j)
        Cons Uyl = corrected(load_cons(W_ + (size_t)W_YP * pl, pl, k - 1),
                             load_cons(FXT, pl, k + p - 1), load_cons(FXT, pl, k - 1), hdtV, Ax);
        Cons Uyr = corrected(load_cons(W_ + (size_t)W_YM * pl, pl, k), load_cons(FXT, pl, k + p),
                             load_cons(FXT, pl, k), hdtV, Ax);
        Cons F = from_n(riemann_rt(to_n(Uyl, false), to_n(Uyr, false), P, false, P.solid_yl && j == g.jlo), false);
        double avy = 0.0;
        if (j <= g.jhi || P.avy_hi) {
            size_t kk = k + p;
            double d10 = div_u_vertex(u[kk], u[kk - 1], u[kk - p], u[kk - p - 1], v[kk],
                                      v[kk - p], v[kk - 1], v[kk - p - 1], P.dx, P.dy);
            double divU_y = 0.5 * (d00 + d10);
            avy = P.cvisc * fmax(-divU_y * P.dy, 0.0);
        }
        const Cons Um = load_cons(U, pl, k - 1);
        F.d += avy * (Um.d - Uc.d);
        F.E += avy * (Um.E - Uc.E);
        F.mx += avy * (Um.mx - Uc.mx);
        F.my += avy * (Um.my - Uc.my);
        store_cons(Wout + (size_t)W_FY * pl, pl, k, F);
    }
}

// ---- stage 5: conservative update + CFL minimum of the new state ---------
// simulation.py:377-384; the per-block minimum of dx/(|u|+c), dy/(|v|+c) of
// the UPDATED cells is the next step's method_compute_timestep (interior min
// == full-array min for outflow / reflect / periodic ghost fills).
__global__ __launch_bounds__(256) void k_update(double *__restrict__ U,
                                                const double *__restrict__ W_, Geom g, CP P,
                                                double *__restrict__ partial, int gx, int gy,
                                                double *__restrict__ keep)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;   // whole block leaves together
    const int j = g.jlo + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo + by;
    const int p = g.pitch;
    const size_t pl = g.plane;
    double cfl = INFINITY;
    if (j <= g.jhi) {
        const size_t k = (size_t)i * p + j;
        const double dtdV = P.dt / (P.dx * P.dy);
        const double Ax = P.dy, Ay = P.dx;
        const double *FX = W_ + (size_t)W_FX * pl, *FY = W_ + (size_t)W_FY * pl;
        double Un[4], Uo[4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const double *fx = FX + (size_t)n * pl, *fy = FY + (size_t)n * pl;
            Uo[n] = U[(size_t)n * pl + k];
            Un[n] = Uo[n] + dtdV * (fx[k] * Ax - fx[k + p] * Ax + fy[k] * Ay - fy[k + 1] * Ay);
        }
        Cons Uc{Un[0], Un[1], Un[2], Un[3]};
        if (P.ext) {
            // predictor only (simulation.py:406-412): U* = U + dt (S_grav(U^n) + S_h(U^n));
            // U^n's density and y momentum stay behind for the corrector
            const Cons X = load_cons(P.ext, pl, k);
            Uc.d = Uc.d + P.dt * X.d;
            Uc.mx = Uc.mx + P.dt * X.mx;
            Uc.my = Uc.my + P.dt * (Uo[0] * P.grav + X.my);
            Uc.E = Uc.E + P.dt * (Uo[3] * P.grav + X.E);
            keep[k] = Uo[0];
            keep[pl + k] = Uo[3];
        } else if (P.have_src)
            grav_update(Uc, Cons{Uo[0], Uo[1], Uo[2], Uo[3]}, P.grav, P.dt, P.heat_rate,
                        P.heat ? P.heat[k] : 0.0);
        store_cons(U, pl, k, Uc);
        cfl = cfl_cell(Uc, P.gamma, P.dx, P.dy);
    }
    cfl = block_reduce_min(cfl);
    if (threadIdx.x == 0) partial[by * gx + bx] = cfl;
}

// ---- corrector of a host-evaluated source (simulation.py:414-423) ---------
// U = U* + dt/2 (S(U*) - S(U^n)), S = gravity (with the time-centred y momentum,
// :148-155) + S_h; U^n's density and y momentum come from k_update (keep)
__global__ __launch_bounds__(256) void k_source_correct(double *__restrict__ U,
                                                        const double *__restrict__ keep,
                                                        const double *__restrict__ Xo_,
                                                        const double *__restrict__ Xn_, Geom g,
                                                        double grav, double dt)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const size_t pl = g.plane, k = (size_t)i * g.pitch + j;
    Cons Us = load_cons(U, pl, k);
    const Cons Xo = load_cons(Xo_, pl, k), Xn = load_cons(Xn_, pl, k);
    const double d_old = keep[k], my_old = keep[pl + k];
    const double Sy_old_g = d_old * grav;
    const double Sy_old = Sy_old_g + Xo.my;
    const double SE_old = my_old * grav + Xo.E;
    const double Sy_new_g = Us.d * grav;
    const double ymom_new = Us.my + 0.5 * dt * (Sy_new_g - Sy_old_g);
    const double SE_new = ymom_new * grav + Xn.E;
    const double Sy_new = Sy_new_g + Xn.my;
    Us.d = Us.d + 0.5 * dt * (Xn.d - Xo.d);
    Us.mx = Us.mx + 0.5 * dt * (Xn.mx - Xo.mx);
    Us.my = Us.my + 0.5 * dt * (Sy_new - Sy_old);
    Us.E = Us.E + 0.5 * dt * (SE_new - SE_old);
    store_cons(U, pl, k, Us);
}

// ---- sponge over the whole array (simulation.py:427-441) -----------------
__global__ __launch_bounds__(256) void k_sponge(double *__restrict__ U, Geom g, double dt,
                                                double rho_begin, double rho_full, double tau)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= g.qy) return;
    const size_t k = (size_t)i * g.pitch + j;
    Cons Uc = load_cons(U, g.plane, k);
    sponge_cell(Uc, dt, rho_begin, rho_full, tau);
    store_cons(U, g.plane, k, Uc);
}

int comp_sponge(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    const Geom &g = s->g;
    hipLaunchKernelGGL(k_sponge, dim3((g.qy + 255) / 256, g.qx), dim3(256), 0, s->ctx->stream,
                       s->d, g, dt, p->sponge_rho_begin, p->sponge_rho_full,
                       p->sponge_timescale);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;   // the cached minimum is from before the sponge
    return 0;
}

// ---- CFL over the whole array (ghost cells included), derives.py ---------
__global__ __launch_bounds__(256) void k_cfl(const double *__restrict__ U, Geom g, double gamma,
                                             double dx, double dy, double *__restrict__ partial)
{
    double m = INFINITY;
    for (int i = blockIdx.y; i < g.qx; i += gridDim.y)
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < g.qy; j += gridDim.x * blockDim.x)
            m = fmin(m, cfl_cell(load_cons(U, g.plane, (size_t)i * g.pitch + j), gamma, dx, dy));
    m = block_reduce_min(m);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
}

static CP make_cp(const pyrohip_comp_params *p, double dt, const pyrohip_state *s)
{
    CP c;
    c.gamma = p->gamma; c.dx = p->dx; c.dy = p->dy; c.dt = dt;
    c.z0 = p->z0; c.z1 = p->z1; c.delta = p->delta; c.cvisc = p->cvisc;
    c.small_dens = p->small_dens;
    c.limiter = p->limiter; c.use_flattening = p->use_flattening;
    c.avx_hi = p->avisc_xhi_interior; c.avy_hi = p->avisc_yhi_interior;
    c.grav = p->grav;
    c.refl_ylo = (s->bc[3 * 4 + 2] == PYROHIP_BC_REFLECT_ODD);
    c.refl_yhi = (s->bc[3 * 4 + 3] == PYROHIP_BC_REFLECT_ODD);
    c.amb_yhi = (s->bc[3 * 4 + 3] == PYROHIP_BC_AMBIENT);
    c.heat = s->heat; c.heat_rate = s->heat ? p->heat_rate : 0.0;
    c.have_src = (p->grav != 0.0 || s->heat != nullptr);
    c.ext = s->ext_old;
    if (c.ext) c.have_src = 0;     // the ext branch adds gravity itself
    c.riemann = p->riemann; c.solid_xl = p->solid_xl; c.solid_yl = p->solid_yl;
    return c;
}

static int ensure_work(pyrohip_state *s, size_t planes)
{
    if (s->work_planes >= planes) return 0;
    if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
    s->work = nullptr; s->work_planes = 0;
    size_t n = s->g.plane * planes + 16;
    PYRO_CHECK_HIP(hipMalloc((void **)&s->work, n * sizeof(double)));
    // zero once: stage dumps of never-written cells then read as 0
    PYRO_CHECK_HIP(hipMemsetAsync(s->work, 0, n * sizeof(double), s->ctx->stream));
    s->work_planes = planes;
    return 0;
}

// CFL minimum over the whole array, left in device memory (no read-back)
int comp_cfl_min_device(pyrohip_state *s, const pyrohip_comp_params *p, const double **dmin)
{
    pyrohip_ctx *c = s->ctx;
    dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    // behind the largest partial array a step kernel may use, so that both fit
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_cfl, grid, block, 0, c->stream, (const double *)s->d, s->g, p->gamma, p->dx,
                       p->dy, part);
    *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int comp_dt(pyrohip_state *s, const pyrohip_comp_params *p, double cfl, double *dt_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    // the hse / ambient boundaries put states into the y ghost rows that no
    // interior cell holds (they do enter the reference's full-array minimum)
    if (s->next_cfl_min > 0.0 && s->cfl_kind == 0 && !s->user_bc && !s->ramp_bc) {   // cached by the last k_update
        *dt_out = cfl * s->next_cfl_min;
        return 0;
    }
    dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_cfl, grid, block, 0, c->stream, (const double *)s->d, g, p->gamma, p->dx,
                       p->dy, part);
    const double *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *dt_out = cfl * ((double *)c->reduce_host)[0];
    return 0;
}

int comp_step_staged(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    PYRO_TRY(ensure_work(s, W_NPLANES));
    const CP P = make_cp(p, dt, s);
    double *U = s->d;
    double *W = s->work + geom_lead(g);
    const dim3 block(256);
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    // all stage kernels: one block = 256 consecutive j of one row; 1-d launch
    // with the XCD-band remap of stencil.h
    int gx = (g.qy + 255) / 256, gy = g.qx;
    PYRO_LAUNCH(c, "k_prim", k_prim, dim3(xcd_grid_1d(gx, gy)), block, 0, U, W, g, P, s->d_flag,
                gx, gy);
    gx = (g.ny + 2 + 255) / 256; gy = g.nx + 2;
    const dim3 gridR1(xcd_grid_1d(gx, gy));
    PYRO_LAUNCH(c, "k_xi", k_xi, gridR1, block, 0, (const double *)W,
                W + (size_t)W_XI * g.plane, g, P, gx, gy);
    PYRO_LAUNCH(c, "k_states", k_states, gridR1, block, 0, (const double *)U, (const double *)W, W, g,
                P, gx, gy);
    PYRO_LAUNCH(c, "k_riemann_t", k_riemann_t, gridR1, block, 0, (const double *)W, W, g, P, gx,
                gy);
    gx = (g.ny + 1 + 255) / 256; gy = g.nx + 1;
    PYRO_LAUNCH(c, "k_final", k_final, dim3(xcd_grid_1d(gx, gy)), block, 0, (const double *)U,
                (const double *)W, W, g, P, gx, gy);
    gx = (g.ny + 255) / 256; gy = g.nx;
    const int nb = gx * gy;
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    // the primitive planes are dead by now: U^n's density and y momentum go there
    // when the step stops at the predictor of a host-evaluated source
    PYRO_LAUNCH(c, "k_update", k_update, dim3(xcd_grid_1d(gx, gy)), block, 0, U,
                (const double *)W, g, P, part, gx, gy, W + (size_t)W_Q * g.plane);
    if (P.ext) s->ext_pending = 1;
    const double *dmin = launch_min_reduce(c->stream, part, nb);
    s->cfl_is_global = false;
    if (c->global_cfl) {   // multi-GPU: the next dt needs the minimum over all slabs
        PYRO_TRY(comm_allreduce_min_device(c, const_cast<double *>(dmin)));
        s->cfl_is_global = true;
    }
    PYRO_CHECK_HIP(hipGetLastError());
    // one 16-byte D2H per step: next step's CFL minimum + positivity flag
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync((char *)c->reduce_host + 8, s->d_flag, sizeof(int),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    s->next_cfl_min = P.ext ? -1.0 : ((double *)c->reduce_host)[0];   // U* is not the new state
    s->cfl_kind = 0;
    int flag = *(int *)((char *)c->reduce_host + 8);
    if (flag & 1) {
        s->next_cfl_min = -1.0;
        s->ext_pending = 0;       // no predictor state to correct: the step did not happen
        set_error("invalid state: min(rho) <= 0 or min(e) <= 0 on the interior "
                  "(compressible/simulation.py:68-71)");
        return PYROHIP_ERR_STATE;
    }
    return 0;
}

int comp_source_correct(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const double *keep = s->work + geom_lead(g) + (size_t)W_Q * g.plane;
    hipLaunchKernelGGL(k_source_correct, dim3((g.ny + 255) / 256, g.nx), dim3(256), 0, c->stream,
                       s->d, keep, s->ext_old, s->ext_new, g, p->grav, dt);
    PYRO_CHECK_HIP(hipGetLastError());
    s->ext_pending = 0;
    s->next_cfl_min = -1.0;
    return 0;
}

// ===========================================================================
// SphericalPolar grids (x = r, y = theta; mesh/patch.py:242-312): the
// coord_type == 1 branches of the reference.  Reuses k_prim / k_xi; the other
// stages get the geometry arrays: per-cell dt/Lx, dt/Ly and the geometric
// source in the tracing (interface.py:106, 215-234), radial gravity and the
// geometric source terms (simulation.py:117-124, 135-147), CGF interface
// states whose pressure enters as a gradient outside the area-weighted flux
// difference (riemann.py:1092-1096, 1156-1171; unsplit_fluxes.py:411-488;
// simulation.py:330-398), the divergence of interface.py:331-364.
// ===========================================================================
enum {
    W_PXT = W_NPLANES,   // pressure of the CGF interface state, transverse solve, x faces
    W_PYT,
    W_PX,                // ... final solve
    W_PY,
    W_SRC,               // 4: external sources in the state's order (density unused)
    W_NPLANES_SPH = W_SRC + 4
};

struct SG {   // kernel-side geometry
    const double *Lx, *Ly, *Ax, *Ay, *V, *dlAx, *dlAy, *x2d, *sint, *sinb, *sinc;
    double xmin, ymin;
};

// get_external_sources, simulation.py:117-124 (U_old is None) on the interior
__global__ __launch_bounds__(256) void k_sph_src(const double *__restrict__ U,
                                                 double *__restrict__ S, Geom g, CP P, SG G)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const size_t k = (size_t)i * g.pitch + j, pl = g.plane;
    Cons Uc = load_cons(U, pl, k);
    Uc.d = fmax(Uc.d, P.small_dens);   // clean_state ran before (k_prim stores it too)
    double Sx = Uc.d * P.grav;
    const double SE = Uc.mx * P.grav;
    Sx += pdiv(Uc.my * Uc.my, Uc.d * G.x2d[k]);
    const double Sy = pdiv(-Uc.mx * Uc.my, Uc.d);
    S[k] = 0.0; S[pl + k] = SE; S[2 * pl + k] = Sx; S[3 * pl + k] = Sy;
}

__global__ __launch_bounds__(256) void k_sph_states(const double *__restrict__ W_,
                                                    double *__restrict__ Wout, Geom g, CP P, SG G,
                                                    int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo - 1 + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    const double *Q = W_ + (size_t)W_Q * pl;
    const double xi = W_[(size_t)W_XI * pl + k];
    double q0[4], dqx[4], dqy[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *a = Q + (size_t)n * pl;
        q0[n] = a[k];
        dqx[n] = xi * limited_slope(a[k - 2 * p], a[k - p], a[k], a[k + p], a[k + 2 * p],
                                    P.limiter);
        dqy[n] = xi * limited_slope(a[k - 2], a[k - 1], a[k], a[k + 1], a[k + 2], P.limiter);
    }
    const double cs = psqrt(pdiv(P.gamma * q0[3], q0[0]));   // interface.py:122
    Trace lo, hi;
    trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], P.gamma,
                 pdiv(P.dt, G.Lx[k]), lo, hi);
    {   // :216-224
        const double rs = -0.5 * P.dt * G.dlAx[k] * q0[0] * q0[1];
        hi.r += rs; lo.r += rs;
        hi.p += rs * cs * cs; lo.p += rs * cs * cs;
    }
    Cons XM = prim_to_cons(Prim{lo.r, lo.un, lo.ut, lo.p}, P.gamma);
    Cons XP = prim_to_cons(Prim{hi.r, hi.un, hi.ut, hi.p}, P.gamma);
    trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], P.gamma,
                 pdiv(P.dt, G.Ly[k]), lo, hi);
    {   // :226-234
        const double rs = -0.5 * P.dt * G.dlAy[k] * q0[0] * q0[2];
        hi.r += rs; lo.r += rs;
        hi.p += rs * cs * cs; lo.p += rs * cs * cs;
    }
    Cons YM = prim_to_cons(Prim{lo.r, lo.ut, lo.un, lo.p}, P.gamma);
    Cons YP = prim_to_cons(Prim{hi.r, hi.ut, hi.un, hi.p}, P.gamma);
    // apply_source_terms with the ghost-filled source planes, unsplit_fluxes.py:308-326
    const double *S = W_ + (size_t)W_SRC * pl;
    const double h = 0.5 * P.dt;
    const double sE = h * S[pl + k], sx = h * S[2 * pl + k], sy = h * S[3 * pl + k];
    XM.mx += sx; XM.my += sy; XM.E += sE;
    XP.mx += sx; XP.my += sy; XP.E += sE;
    YM.mx += sx; YM.my += sy; YM.E += sE;
    YP.mx += sx; YP.my += sy; YP.E += sE;
    store_cons(Wout + (size_t)W_XM * pl, pl, k, XM);
    store_cons(Wout + (size_t)W_XP * pl, pl, k, XP);
    store_cons(Wout + (size_t)W_YM * pl, pl, k, YM);
    store_cons(Wout + (size_t)W_YP * pl, pl, k, YP);
}

// CGF interface state, its flux without the pressure and its pressure
// (riemann_flux(return_cons=True) + cons_to_prim, unsplit_fluxes.py:411-423)
__device__ __forceinline__ Cons sph_face(const Cons &Ul, const Cons &Ur, const CP &P, bool x,
                                         bool wall, double &pface)
{
    const ConsN Uo = cgf_state(to_n(Ul, x), to_n(Ur, x), P.gamma, wall);
    pface = cons_to_prim(from_n(Uo, x), P.gamma).p;
    return from_n(cons_flux_n(Uo, P.gamma, x, false), x);
}

__global__ __launch_bounds__(256) void k_sph_riemann_t(const double *__restrict__ W_,
                                                       double *__restrict__ Wout, Geom g, CP P,
                                                       int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo - 1 + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    if (i >= g.ilo) {
        double pf;
        const Cons F = sph_face(load_cons(W_ + (size_t)W_XP * pl, pl, k - p),
                                load_cons(W_ + (size_t)W_XM * pl, pl, k), P, true,
                                P.solid_xl && i == g.ilo, pf);
        store_cons(Wout + (size_t)W_FXT * pl, pl, k, F);
        Wout[(size_t)W_PXT * pl + k] = pf;
    }
    if (j >= g.jlo) {
        double pf;
        const Cons F = sph_face(load_cons(W_ + (size_t)W_YP * pl, pl, k - 1),
                                load_cons(W_ + (size_t)W_YM * pl, pl, k), P, false,
                                P.solid_yl && j == g.jlo, pf);
        store_cons(Wout + (size_t)W_FYT * pl, pl, k, F);
        Wout[(size_t)W_PYT * pl + k] = pf;
    }
}

__device__ __forceinline__ Cons corrected_g(const Cons &U, const Cons &Fhi, double Ahi,
                                            const Cons &Flo, double Alo, double hv)
{
    Cons r;   // U += -hdtV*(F_hi*A_hi - F_lo*A_lo), unsplit_fluxes.py:447-471
    r.d = U.d + (-hv * (Fhi.d * Ahi - Flo.d * Alo));
    r.E = U.E + (-hv * (Fhi.E * Ahi - Flo.E * Alo));
    r.mx = U.mx + (-hv * (Fhi.mx * Ahi - Flo.mx * Alo));
    r.my = U.my + (-hv * (Fhi.my * Ahi - Flo.my * Alo));
    return r;
}

// vertex divergence at (i-1/2, j-1/2) on the spherical grid, interface.py:331-364
__device__ __forceinline__ double div_u_vertex_sph(const double *__restrict__ u,
                                                   const double *__restrict__ v, size_t k, int p,
                                                   int i, int j, const Geom &g, const CP &P,
                                                   const SG &G)
{
    const double rr = (i + 0.5 - g.ng) * P.dx + G.xmin;
    const double rl = (i - 0.5 - g.ng) * P.dx + G.xmin;
    const double rc = (i - g.ng) * P.dx + G.xmin;
    const double ur = 0.5 * (u[k] + u[k - 1]);
    const double ul = 0.5 * (u[k - p] + u[k - p - 1]);
    const double ux = pdiv(ur * rr * rr - ul * rl * rl, rc * rc * P.dx);
    const double sint = G.sint[j], sinb = G.sinb[j], sinc = G.sinc[j];
    double vy = 0.0;
    if (sinc != 0.0) {
        const double vt = 0.5 * (v[k] + v[k - p]);
        const double vb = 0.5 * (v[k - 1] + v[k - p - 1]);
        vy = pdiv(sint * vt - sinb * vb, rc * sinc * P.dy);
    }
    return ux + vy;
}

__global__ __launch_bounds__(256) void k_sph_final(const double *__restrict__ U,
                                                   const double *__restrict__ W_,
                                                   double *__restrict__ Wout, Geom g, CP P, SG G,
                                                   int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    const double hdt = 0.5 * P.dt;
    const double hv = pdiv(hdt, G.V[k]);
    const double *FXT = W_ + (size_t)W_FXT * pl, *FYT = W_ + (size_t)W_FYT * pl;
    const double *PXT = W_ + (size_t)W_PXT * pl, *PYT = W_ + (size_t)W_PYT * pl;
    const double *u = W_ + (size_t)(W_Q + 1) * pl, *v = W_ + (size_t)(W_Q + 2) * pl;
    const double d00 = div_u_vertex_sph(u, v, k, p, i, j, g, P, G);
    const Cons Uc = load_cons(U, pl, k);

    if (j <= g.jhi) {   // x face (i,j)
        Cons Uxl = corrected_g(load_cons(W_ + (size_t)W_XP * pl, pl, k - p),
                               load_cons(FYT, pl, k - p + 1), G.Ay[k - p + 1],
                               load_cons(FYT, pl, k - p), G.Ay[k - p], hv);
        Cons Uxr = corrected_g(load_cons(W_ + (size_t)W_XM * pl, pl, k), load_cons(FYT, pl, k + 1),
                               G.Ay[k + 1], load_cons(FYT, pl, k), G.Ay[k], hv);
        // pressure gradient on the transverse momentum, :476-481 (Ly of cell (i,j))
        Uxl.my += pdiv(-hdt * (PYT[k - p + 1] - PYT[k - p]), G.Ly[k]);
        Uxr.my += pdiv(-hdt * (PYT[k + 1] - PYT[k]), G.Ly[k]);
        double pf;
        Cons F = sph_face(Uxl, Uxr, P, true, P.solid_xl && i == g.ilo, pf);
        double avx = 0.0;
        if (i <= g.ihi) {
            const double d01 = div_u_vertex_sph(u, v, k + 1, p, i, j + 1, g, P, G);
            const double divU_x = 0.5 * (d00 + d01);
            avx = P.cvisc * fmax(-divU_x * G.Lx[k], 0.0);
        }
        const Cons Um = load_cons(U, pl, k - p);
        F.d += avx * (Um.d - Uc.d);
        F.E += avx * (Um.E - Uc.E);
        F.mx += avx * (Um.mx - Uc.mx);
        F.my += avx * (Um.my - Uc.my);
        store_cons(Wout + (size_t)W_FX * pl, pl, k, F);
        Wout[(size_t)W_PX * pl + k] = pf;
    }
    if (i <= g.ihi) {   // y face (i,j)
        Cons Uyl = corrected_g(load_cons(W_ + (size_t)W_YP * pl, pl, k - 1),
                               load_cons(FXT, pl, k + p - 1), G.Ax[k + p - 1],
                               load_cons(FXT, pl, k - 1), G.Ax[k - 1], hv);
        Cons Uyr = corrected_g(load_cons(W_ + (size_t)W_YM * pl, pl, k), load_cons(FXT, pl, k + p),
                               G.Ax[k + p], load_cons(FXT, pl, k), G.Ax[k], hv);
        Uyl.mx += pdiv(-hdt * (PXT[k + p - 1] - PXT[k - 1]), G.Lx[k]);
        Uyr.mx += pdiv(-hdt * (PXT[k + p] - PXT[k]), G.Lx[k]);
        double pf;
        Cons F = sph_face(Uyl, Uyr, P, false, P.solid_yl && j == g.jlo, pf);
        double avy = 0.0;
        if (j <= g.jhi) {
            const double d10 = div_u_vertex_sph(u, v, k + p, p, i + 1, j, g, P, G);
            const double divU_y = 0.5 * (d00 + d10);
            avy = P.cvisc * fmax(-divU_y * G.Ly[k], 0.0);
        }
        const Cons Um = load_cons(U, pl, k - 1);
        F.d += avy * (Um.d - Uc.d);
        F.E += avy * (Um.E - Uc.E);
        F.mx += avy * (Um.mx - Uc.mx);
        F.my += avy * (Um.my - Uc.my);
        store_cons(Wout + (size_t)W_FY * pl, pl, k, F);
        Wout[(size_t)W_PY * pl + k] = pf;
    }
}

// conservative update with the area / volume arrays, the pressure gradients
// and the source predictor-corrector, simulation.py:375-423
__global__ __launch_bounds__(256) void k_sph_update(double *__restrict__ U,
                                                    const double *__restrict__ W_, Geom g, CP P,
                                                    SG G, int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo + by;
    if (j > g.jhi) return;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const size_t k = (size_t)i * p + j;
    const double dtdV = pdiv(P.dt, G.V[k]);
    const double *FX = W_ + (size_t)W_FX * pl, *FY = W_ + (size_t)W_FY * pl;
    const double *PX = W_ + (size_t)W_PX * pl, *PY = W_ + (size_t)W_PY * pl;
    double Un[4], Uo[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *fx = FX + (size_t)n * pl, *fy = FY + (size_t)n * pl;
        Uo[n] = U[(size_t)n * pl + k];
        Un[n] = Uo[n] + dtdV * (fx[k] * G.Ax[k] - fx[k + p] * G.Ax[k + p] + fy[k] * G.Ay[k] -
                                fy[k + 1] * G.Ay[k + 1]);
    }
    Un[2] -= pdiv(P.dt * (PX[k + p] - PX[k]), G.Lx[k]);
    Un[3] -= pdiv(P.dt * (PY[k + 1] - PY[k]), G.Ly[k]);
    // S_old = S(U_old); U += dt S_old; S_new (time-centred x-momentum); U += dt/2 (S_new - S_old)
    const double r = G.x2d[k], grav = P.grav, dt = P.dt;
    const double Sx_g_old = Uo[0] * grav;
    const double SE_old = Uo[2] * grav;
    const double Sx_old = Sx_g_old + pdiv(Uo[3] * Uo[3], Uo[0] * r);
    const double Sy_old = pdiv(-Uo[2] * Uo[3], Uo[0]);
    Un[1] = Un[1] + dt * SE_old;
    Un[2] = Un[2] + dt * Sx_old;
    Un[3] = Un[3] + dt * Sy_old;
    const double Sx_g_new = Un[0] * grav;
    const double xmom_new = Un[2] + 0.5 * dt * (Sx_g_new - Sx_g_old);
    const double SE_new = xmom_new * grav;
    const double Sx_new = Sx_g_new + pdiv(Un[3] * Un[3], Un[0] * r);
    const double Sy_new = pdiv(-Un[2] * Un[3], Un[0]);
    U[k] = Un[0];   // the density source is zero: U += dt * 0, U += dt/2 * (0 - 0)
    U[pl + k] = Un[1] + 0.5 * dt * (SE_new - SE_old);
    U[2 * pl + k] = Un[2] + 0.5 * dt * (Sx_new - Sx_old);
    U[3 * pl + k] = Un[3] + 0.5 * dt * (Sy_new - Sy_old);
}

// method_compute_timestep with Lx, Ly arrays (simulation.py:284-288), whole array
__global__ __launch_bounds__(256) void k_sph_cfl(const double *__restrict__ U, Geom g,
                                                 double gamma, const double *__restrict__ Lx,
                                                 const double *__restrict__ Ly,
                                                 double *__restrict__ partial)
{
    double m = INFINITY;
    for (int i = blockIdx.y; i < g.qx; i += gridDim.y)
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < g.qy; j += gridDim.x * blockDim.x) {
            const size_t k = (size_t)i * g.pitch + j;
            m = fmin(m, cfl_cell(load_cons(U, g.plane, k), gamma, Lx[k], Ly[k]));
        }
    m = block_reduce_min(m);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
}

static SG make_sg(const pyrohip_state *s)
{
    const SphGeom &h = *s->sph;
    return SG{h.Lx, h.Ly, h.Ax, h.Ay, h.V, h.dlAx, h.dlAy, h.x2d, h.sint, h.sinb, h.sinc, h.xmin,
              h.ymin};
}

// ... left in device memory (no read-back): the first step of a device-side run
int comp_cfl_min_device_sph(pyrohip_state *s, const pyrohip_comp_params *p, const double **dmin)
{
    pyrohip_ctx *c = s->ctx;
    dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_sph_cfl, grid, block, 0, c->stream, (const double *)s->d, s->g, p->gamma,
                       s->sph->Lx, s->sph->Ly, part);
    *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int comp_dt_sph(pyrohip_state *s, const pyrohip_comp_params *p, double cfl, double *dt_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_sph_cfl, grid, block, 0, c->stream, (const double *)s->d, g, p->gamma,
                       s->sph->Lx, s->sph->Ly, part);
    const double *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *dt_out = cfl * ((double *)c->reduce_host)[0];
    return 0;
}

int comp_step_sph(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    PYRO_TRY(ensure_work(s, W_NPLANES_SPH));
    const CP P = make_cp(p, dt, s);
    const SG G = make_sg(s);
    double *U = s->d;
    double *W = s->work + geom_lead(g);
    const dim3 block(256);
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    int gx = (g.qy + 255) / 256, gy = g.qx;
    PYRO_LAUNCH(c, "k_prim", k_prim, dim3(xcd_grid_1d(gx, gy)), block, 0, U, W, g, P, s->d_flag,
                gx, gy);
    // external sources on the interior, then their ghost cells like the
    // reference's dens_src .. E_src variables (boundary types of the state)
    double *S = W + (size_t)W_SRC * g.plane;
    PYRO_LAUNCH(c, "k_sph_src", k_sph_src, dim3((g.ny + 255) / 256, g.nx), block, 0,
                (const double *)U, S, g, P, G);
    PYRO_TRY(fill_bc_planes(s, S, 0, 4));
    gx = (g.ny + 2 + 255) / 256; gy = g.nx + 2;
    const dim3 gridR1(xcd_grid_1d(gx, gy));
    PYRO_LAUNCH(c, "k_xi", k_xi, gridR1, block, 0, (const double *)W,
                W + (size_t)W_XI * g.plane, g, P, gx, gy);
    PYRO_LAUNCH(c, "k_sph_states", k_sph_states, gridR1, block, 0, (const double *)W, W, g, P, G,
                gx, gy);
    PYRO_LAUNCH(c, "k_sph_riemann_t", k_sph_riemann_t, gridR1, block, 0, (const double *)W, W, g,
                P, gx, gy);
    gx = (g.ny + 1 + 255) / 256; gy = g.nx + 1;
    PYRO_LAUNCH(c, "k_sph_final", k_sph_final, dim3(xcd_grid_1d(gx, gy)), block, 0,
                (const double *)U, (const double *)W, W, g, P, G, gx, gy);
    gx = (g.ny + 255) / 256; gy = g.nx;
    PYRO_LAUNCH(c, "k_sph_update", k_sph_update, dim3(xcd_grid_1d(gx, gy)), block, 0, U,
                (const double *)W, g, P, G, gx, gy);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, s->d_flag, sizeof(int), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    s->next_cfl_min = -1.0;
    s->cfl_is_global = false;
    if (*(int *)c->reduce_host & 1) {
        set_error("invalid state: min(rho) <= 0 or min(e) <= 0 on the interior "
                  "(compressible/simulation.py:68-71)");
        return PYROHIP_ERR_STATE;
    }
    return 0;
}

// ===========================================================================
// compressible_rk: method-of-lines right-hand side k = -div F + S
// (pyro/compressible_rk/fluxes.py:28-180, simulation.py:10-44).  Reuses
// k_prim / k_xi; the face states are piecewise linear (no characteristic
// tracing, no transverse terms), then one Riemann problem per face plus the
// artificial viscosity, then the flux divergence.
// ===========================================================================
// R1: face states of the cells of R(1) (the faces of the interior need no more)
__global__ __launch_bounds__(256) void k_rk_states(const double *__restrict__ W_,
                                                   double *__restrict__ Wout, Geom g, CP P,
                                                   int gx, int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo - 1 + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    const double *Q = W_ + (size_t)W_Q * pl;
    const double xi = W_[(size_t)W_XI * pl + k];
    double q0[4], dqx[4], dqy[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *a = Q + (size_t)n * pl;
        q0[n] = a[k];
        dqx[n] = xi * limited_slope(a[k - 2 * p], a[k - p], a[k], a[k + p], a[k + 2 * p],
                                    P.limiter);
        dqy[n] = xi * limited_slope(a[k - 2], a[k - 1], a[k], a[k + 1], a[k + 2], P.limiter);
    }
    // fluxes.py:107-140: V_l[i+1] = q + ld/2 (the cell's upper face), V_r[i] = q - ld/2
    auto face = [&](const double *d, double sgn) {
        return prim_to_cons(Prim{q0[0] + sgn * 0.5 * d[0], q0[1] + sgn * 0.5 * d[1],
                                 q0[2] + sgn * 0.5 * d[2], q0[3] + sgn * 0.5 * d[3]}, P.gamma);
    };
    store_cons(Wout + (size_t)W_XM * pl, pl, k, face(dqx, -1.0));
    store_cons(Wout + (size_t)W_XP * pl, pl, k, face(dqx, 1.0));
    store_cons(Wout + (size_t)W_YM * pl, pl, k, face(dqy, -1.0));
    store_cons(Wout + (size_t)W_YP * pl, pl, k, face(dqy, 1.0));
}

// R2: Riemann problems (fluxes.py:145-160) + artificial viscosity (:162-168,
// unsplit_fluxes.py:525-547) on the lower faces of thread (i,j) in
// [ilo, ihi+1] x [jlo, jhi+1]
__global__ __launch_bounds__(256) void k_rk_flux(const double *__restrict__ U,
                                                 const double *__restrict__ W_,
                                                 double *__restrict__ Wout, Geom g, CP P, int gx,
                                                 int gy)
{
    int bx, by;
    if (!xcd_block_2d(gx, gy, bx, by)) return;
    const int j = g.jlo + bx * blockDim.x + threadIdx.x;
    const int i = g.ilo + by;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j;
    const size_t pl = g.plane;
    const double *u = W_ + (size_t)(W_Q + 1) * pl, *v = W_ + (size_t)(W_Q + 2) * pl;
    const double d00 = div_u_vertex(u[k], u[k - 1], u[k - p], u[k - p - 1], v[k], v[k - p],
                                    v[k - 1], v[k - p - 1], P.dx, P.dy);
    const Cons Uc = load_cons(U, pl, k);
    if (j <= g.jhi) {
        const Cons Ul = load_cons(W_ + (size_t)W_XP * pl, pl, k - p);
        const Cons Ur = load_cons(W_ + (size_t)W_XM * pl, pl, k);
        Cons F = from_n(riemann_rt(to_n(Ul, true), to_n(Ur, true), P, true,
                                   P.solid_xl && i == g.ilo), true);
        double avx = 0.0;
        if (i <= g.ihi || P.avx_hi) {
            const size_t kk = k + 1;
            const double d01 = div_u_vertex(u[kk], u[kk - 1], u[kk - p], u[kk - p - 1], v[kk],
                                            v[kk - p], v[kk - 1], v[kk - p - 1], P.dx, P.dy);
            avx = P.cvisc * fmax(-(0.5 * (d00 + d01)) * P.dx, 0.0);
        }
        const Cons Um = load_cons(U, pl, k - p);
        F.d += avx * (Um.d - Uc.d);
        F.E += avx * (Um.E - Uc.E);
        F.mx += avx * (Um.mx - Uc.mx);
        F.my += avx * (Um.my - Uc.my);
        store_cons(Wout + (size_t)W_FX * pl, pl, k, F);
    }
    if (i <= g.ihi) {
        const Cons Ul = load_cons(W_ + (size_t)W_YP * pl, pl, k - 1);
        const Cons Ur = load_cons(W_ + (size_t)W_YM * pl, pl, k);
        Cons F = from_n(riemann_rt(to_n(Ul, false), to_n(Ur, false), P, false,
                                   P.solid_yl && j == g.jlo), false);
        double avy = 0.0;
        if (j <= g.jhi || P.avy_hi) {
            const size_t kk = k + p;
            const double d10 = div_u_vertex(u[kk], u[kk - 1], u[kk - p], u[kk - p - 1], v[kk],
                                            v[kk - p], v[kk - 1], v[kk - p - 1], P.dx, P.dy);
            avy = P.cvisc * fmax(-(0.5 * (d00 + d10)) * P.dy, 0.0);
        }
        const Cons Um = load_cons(U, pl, k - 1);
        F.d += avy * (Um.d - Uc.d);
        F.E += avy * (Um.E - Uc.E);
        F.mx += avy * (Um.mx - Uc.mx);
        F.my += avy * (Um.my - Uc.my);
        store_cons(Wout + (size_t)W_FY * pl, pl, k, F);
    }
}

// R3: k = (Fx[i] - Fx[i+1])/dx + (Fy[j] - Fy[j+1])/dy + S (gravity) - sponge
// (compressible_rk/simulation.py:16-42) into 4 planes of the k state
struct RkSponge { int on; double rho_begin, rho_full, tau; };
__global__ __launch_bounds__(256) void k_rk_rhs(const double *__restrict__ U,
                                                const double *__restrict__ W_,
                                                double *__restrict__ K, Geom g, CP P,
                                                RkSponge sp)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const size_t k = (size_t)i * p + j;
    const double *FX = W_ + (size_t)W_FX * pl, *FY = W_ + (size_t)W_FY * pl;
    const Cons Uc = load_cons(U, pl, k);
    double kk[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *fx = FX + (size_t)n * pl, *fy = FY + (size_t)n * pl;
        kk[n] = (fx[k] - fx[k + p]) / P.dx + (fy[k] - fy[k + 1]) / P.dy;
    }
    // planes: density, energy, x-momentum, y-momentum; S = (0, ymom g, 0, rho g)
    kk[0] = kk[0] + 0.0;
    kk[1] = kk[1] + (Uc.my * P.grav + Uc.d * P.heat_rate * (P.heat ? P.heat[k] : 0.0));
    kk[2] = kk[2] + 0.0;
    kk[3] = kk[3] + Uc.d * P.grav;
    if (sp.on) {
        const double PI = 3.14159265358979323846;
        double f;
        if (Uc.d > sp.rho_begin) f = 0.0;
        else if (Uc.d < sp.rho_full) f = 1.0;
        else f = 0.5 * (1.0 - cos(PI * (Uc.d - sp.rho_begin) / (sp.rho_full - sp.rho_begin)));
        const double kap = f / sp.tau;
        kk[2] -= kap * Uc.mx;
        kk[3] -= kap * Uc.my;
        kk[1] -= kap * (Uc.mx * Uc.mx / Uc.d + Uc.my * Uc.my / Uc.d);
    }
#pragma unroll
    for (int n = 0; n < 4; n++) K[(size_t)n * pl + k] = kk[n];
}

// compressible_rk/simulation.py:46-56: cfl * min 1 / ((|u|+c)/dx + (|v|+c)/dy)
__global__ __launch_bounds__(256) void k_rk_cfl(const double *__restrict__ U, Geom g, double gamma,
                                                double dx, double dy, double *__restrict__ partial)
{
    double m = INFINITY;
    for (int i = blockIdx.y; i < g.qx; i += gridDim.y)
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < g.qy; j += gridDim.x * blockDim.x) {
            const Cons Uc = load_cons(U, g.plane, (size_t)i * g.pitch + j);
            const double u = Uc.mx / Uc.d, v = Uc.my / Uc.d;
            const double e = (Uc.E - 0.5 * Uc.d * (u * u + v * v)) / Uc.d;
            const double pr = Uc.d * e * (gamma - 1.0);
            const double cs = sqrt(gamma * pr / Uc.d);
            m = fmin(m, 1.0 / ((fabs(u) + cs) / dx + (fabs(v) + cs) / dy));
        }
    m = block_reduce_min(m);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
}

// the same minimum left in device memory (first step of pyrohip_comp_rk_evolve)
int comp_rk_cfl_min_device(pyrohip_state *s, const pyrohip_comp_params *p, const double **dmin)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    // (behind the partials of the step kernels, which use the front of the same buffer)
    PYRO_TRY(c->reduce.ensure((size_t)(2 * nb + kMinStageBlocks + 2 + 65536) * sizeof(double)));
    double *part = (double *)c->reduce.p + 65536;
    hipLaunchKernelGGL(k_rk_cfl, grid, block, 0, c->stream, (const double *)s->d, g, p->gamma, p->dx,
                       p->dy, part);
    *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int comp_rk_dt(pyrohip_state *s, const pyrohip_comp_params *p, double cfl, double *dt_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_rk_cfl, grid, block, 0, c->stream, (const double *)s->d, g, p->gamma, p->dx,
                       p->dy, part);
    const double *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *dt_out = cfl * ((double *)c->reduce_host)[0];
    return 0;
}

// y: stage state (ghost cells filled; the density floor is applied in place
// like clean_state); kst / slot: where the 4 planes of k go
int comp_rk_rhs(pyrohip_state *s, const pyrohip_comp_params *p, pyrohip_state *kst, int slot)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    PYRO_TRY(ensure_work(s, W_NPLANES));
    const CP P = make_cp(p, 0.0, s);
    double *U = s->d;
    double *W = s->work + geom_lead(g);
    const dim3 block(256);
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    int gx = (g.qy + 255) / 256, gy = g.qx;
    PYRO_LAUNCH(c, "k_prim", k_prim, dim3(xcd_grid_1d(gx, gy)), block, 0, U, W, g, P, s->d_flag,
                gx, gy);
    gx = (g.ny + 2 + 255) / 256; gy = g.nx + 2;
    const dim3 gridR1(xcd_grid_1d(gx, gy));
    PYRO_LAUNCH(c, "k_xi", k_xi, gridR1, block, 0, (const double *)W, W + (size_t)W_XI * g.plane, g,
                P, gx, gy);
    PYRO_LAUNCH(c, "k_rk_states", k_rk_states, gridR1, block, 0, (const double *)W, W, g, P, gx, gy);
    gx = (g.ny + 1 + 255) / 256; gy = g.nx + 1;
    PYRO_LAUNCH(c, "k_rk_flux", k_rk_flux, dim3(xcd_grid_1d(gx, gy)), block, 0, (const double *)U,
                (const double *)W, W, g, P, gx, gy);
    const RkSponge sp{p->do_sponge, p->sponge_rho_begin, p->sponge_rho_full, p->sponge_timescale};
    PYRO_LAUNCH(c, "k_rk_rhs", k_rk_rhs, dim3((g.ny + 255) / 256, g.nx), block, 0,
                (const double *)U, (const double *)W, kst->d + (size_t)(4 * slot) * g.plane, g, P,
                sp);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, s->d_flag, sizeof(int), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    s->next_cfl_min = -1.0;
    if (*(int *)c->reduce_host & 1) {
        set_error("invalid state: min(rho) <= 0 or min(e) <= 0 on the interior "
                  "(compressible/simulation.py:68-71)");
        return PYROHIP_ERR_STATE;
    }
    return 0;
}

int comp_stage_dump(pyrohip_state *s, int stage_id, double *out)
{
    static const int first[10] = {W_Q, W_XI, W_XM, W_XP, W_YM, W_YP, W_FXT, W_FYT, W_FX, W_FY};
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    PYRO_REQUIRE(stage_id >= 0 && stage_id < 10, "stage id out of range");
    PYRO_REQUIRE(s->work_planes >= W_NPLANES, "no staged step has been run");
    const int ncomp = (stage_id == 1) ? 1 : 4;
    const double *W = s->work + geom_lead(g) + (size_t)first[stage_id] * g.plane;
    std::vector<double> tmp((size_t)g.qx * g.qy);
    for (int n = 0; n < ncomp; n++) {
        PYRO_CHECK_HIP(hipMemcpy2DAsync(tmp.data(), g.qy * sizeof(double), W + (size_t)n * g.plane,
                                        g.pitch * sizeof(double), g.qy * sizeof(double), g.qx,
                                        hipMemcpyDeviceToHost, c->stream));
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < tmp.size(); k++) out[k * ncomp + n] = tmp[k];
    }
    return 0;
}

}  // namespace PYRO_NS
}  // namespace pyro
