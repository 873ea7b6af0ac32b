// SphericalPolar pieces shared by the two one-launch kernels of the compressible step on such a
// grid: the 2-d LDS tile kernel (comp_fused.hip: k_ctu_fused_sph) and the row-marching kernel
// (comp_sph_wave.hip: k_sph_wave).  Included inside namespace pyro::PYRO_NS after fused_common.h.
#pragma once

struct SphG {   // kernel-side geometry (pyrohip_state_set_geometry)
    const double *Lx, *Ly, *Ax, *Ay, *V, *dlAx, *dlAy, *x2d, *sint, *sinb, *sinc;
    double xmin;
    // FAC instance (round 6): the geometry rebuilt from its 1-d factors -- rowf: A D F G Ly dlogAx x
    // (kSphRowStride doubles per row, common.h), colf: B C E T (stride qyp) -- with the bits of the planes (include/pyrohip.h:
    // pyrohip_geom; mesh/patch.py checks the factorisation when it hands the tables over).  The
    // plane-reading instance moved 312 B of fabric traffic per cell update for 64 algorithmic ones
    // (profiles/r05_sph2048_pmc.json): eight planes, most of them read for two or three cells.
    const double *rowf, *colf;
    int qxp, qyp;
};
// value of a geometry array at (row r, column c) -- FAC: from the factors
template <bool FAC> struct SphAt {
    const SphG &G; int p; double dx;
    __device__ __forceinline__ double rf(int k, int r) const { return G.rowf[r * kSphRowStride + k]; }
    __device__ __forceinline__ double cf(int k, int c) const { return G.colf[k * G.qyp + c]; }
    __device__ __forceinline__ double Lx(int r, int c) const { return FAC ? dx : G.Lx[(size_t)r * p + c]; }
    __device__ __forceinline__ double Ly(int r, int c) const { return FAC ? rf(4, r) : G.Ly[(size_t)r * p + c]; }
    __device__ __forceinline__ double Ax(int r, int c) const { return FAC ? fabs(rf(0, r) * cf(0, c)) : G.Ax[(size_t)r * p + c]; }
    __device__ __forceinline__ double Ay(int r, int c) const { return FAC ? fabs(cf(1, c) * rf(1, r)) : G.Ay[(size_t)r * p + c]; }
    __device__ __forceinline__ double V(int r, int c) const { return FAC ? fabs((cf(2, c) * rf(2, r)) * rf(3, r)) : G.V[(size_t)r * p + c]; }
    __device__ __forceinline__ double dlAx(int r, int c) const { return FAC ? rf(5, r) : G.dlAx[(size_t)r * p + c]; }
    __device__ __forceinline__ double dlAy(int r, int c) const { return FAC ? pdiv(1.0, cf(3, c) * rf(6, r)) : G.dlAy[(size_t)r * p + c]; }
    __device__ __forceinline__ double x(int r, int c) const { return FAC ? rf(6, r) : G.x2d[(size_t)r * p + c]; }
};
// CGF interface state, its flux without the pressure and its pressure
// (riemann_flux(return_cons=True) + cons_to_prim, unsplit_fluxes.py:411-423)
#if PYRO_FAST && !defined(PYRO_SPHF_RESTATED)      // (PYRO_SPHF_RESTATED: developer A/B)
// Contracted build (round 6): the two-shock solver of riemann.py:8-310 written for the instruction
// count -- the bit-faithful restatement (hydro.h: cgf_state, then cons_to_prim and cons_flux_n of
// the conserved interface state) issues 13 quarter-rate reciprocals / roots per face, this one 5
// in gas at rest and 7 elsewhere:
//  * one reciprocal root per side: W = sqrt(gamma p rho) = x rsq(x), c = W / rho, 1 / c^2 = (rho rsq(x))^2;
//  * the star state of ONE side only -- the one the contact's velocity selects -- with the normal
//    velocity of the right side mirrored, so that the wave-pattern tests of the two sides
//    (riemann.py:142-262) are one piece of code (sigma == 0 keeps the reference's choice per side);
//  * the interface state stays primitive: its pressure is rho e (gamma - 1), the pressure-free flux
//    is built from (rho, u_n, u_t, rho e) -- no round trip through the conserved state.
// Differs from the restatement by rounding only (tolerance-tested like the rest of this build).
__device__ __forceinline__ Cons sphf_face(const Cons &Ul, const Cons &Ur, double gamma, bool x,
                                          bool wall, double &pface)
{
    const ConsN l = to_nf(Ul, x), r = to_nf(Ur, x);
    const double smallc = 1.e-10, smallrho = 1.e-10, smallp = 1.e-10;
    const double gm1 = gamma - 1.0;
    const double ril = prcp(l.d), rir = prcp(r.d);
    const double un_l = l.mn * ril, ut_l = l.mt * ril, un_r = r.mn * rir, ut_r = r.mt * rir;
    const double rhoe_l = fma(-0.5, fma(l.mt, ut_l, l.mn * un_l), l.E);
    const double rhoe_r = fma(-0.5, fma(r.mt, ut_r, r.mn * un_r), r.E);
    const double p_l = fmax(rhoe_l * gm1, smallp), p_r = fmax(rhoe_r * gm1, smallp);
    double rs_l, rs_r;
    const double W_l = fmax(smallrho * smallc, psqrt_r(gamma * p_l * l.d, rs_l));
    const double W_r = fmax(smallrho * smallc, psqrt_r(gamma * p_r * r.d, rs_r));
    const double c_l = fmax(smallc, W_l * ril), c_r = fmax(smallc, W_r * rir);
    const double rW = prcp(W_l + W_r);
    const double pstar = fmax((W_l * p_r + W_r * p_l + W_l * W_r * (un_l - un_r)) * rW, smallp);
    const double ustar = (W_l * un_l + W_r * un_r + (p_l - p_r)) * rW;
    // 1 / c, with the floor of c as a ceiling -- which also keeps the states of ghost faces that are
    // no gas at all (a density reflected ODDLY is negative: x < 0, the root is NaN and fmin / fmax
    // return their other operand, as in the restatement) finite: what they feed is never stored
    const double qc_l = fmin(l.d * rs_l, 1.0 / smallc), qc_r = fmin(r.d * rs_r, 1.0 / smallc);
    const double t_l = (pstar - p_l) * (qc_l * qc_l), t_r = (pstar - p_r) * (qc_r * qc_r);
    double rho_s, un_s, ut_s, rhoe_s;
    if (ustar != 0.0) {
        const bool L = ustar > 0.0;
        const double rho_k = L ? l.d : r.d, w_k = L ? un_l : -un_r, rhoe_k = L ? rhoe_l : rhoe_r;
        const double p_k = L ? p_l : p_r, c_k = L ? c_l : c_r, t_k = L ? t_l : t_r, ri_k = L ? ril : rir;
        const double rhostar = rho_k + t_k;
        const double rhoestar = fma(t_k * ri_k, rhoe_k + p_k, rhoe_k);
        const double wstar = fabs(ustar);
        const double cstar = fmax(smallc, psqrt(gamma * pstar * prcp(rhostar)));
        const double lam = w_k - c_k, lamstar = wstar - cstar;
        double a;            // weight of the star state
        if (pstar > p_k) {
            const double sigma = 0.5 * (lam + lamstar);
            a = (L ? sigma > 0.0 : sigma >= 0.0) ? 0.0 : 1.0;
        } else if (lam < 0.0 && lamstar < 0.0) {
            a = 1.0;
        } else if (lam > 0.0 && lamstar > 0.0) {
            a = 0.0;
        } else {
            a = lam * prcp(lam - lamstar);
        }
        rho_s = fma(a, rhostar - rho_k, rho_k);
        const double w_s = fma(a, wstar - w_k, w_k);
        un_s = L ? w_s : -w_s;
        rhoe_s = fma(a, rhoestar - rhoe_k, rhoe_k);
        ut_s = L ? ut_l : ut_r;
    } else {
        rho_s = 0.5 * ((l.d + t_l) + (r.d + t_r));
        un_s = ustar;
        ut_s = 0.5 * (ut_l + ut_r);
        rhoe_s = 0.5 * (fma(t_l * ril, rhoe_l + p_l, rhoe_l) + fma(t_r * rir, rhoe_r + p_r, rhoe_r));
    }
    if (wall) un_s = 0.0;
    pface = rhoe_s * gm1;
    ConsN F;
    F.d = rho_s * un_s;
    F.mn = F.d * un_s;
    F.mt = F.d * ut_s;
    F.E = (fma(0.5 * rho_s, fma(un_s, un_s, ut_s * ut_s), rhoe_s) + pface) * un_s;
#if defined(PYRO_EMU) && defined(SPHF_DEBUG)
    if (!(pface == pface) || !(F.d == F.d) || !(F.E == F.E))
        printf("NAN face: l %.17g %.17g %.17g %.17g r %.17g %.17g %.17g %.17g x %d wall %d ustar %g pstar %g rho_s %g rhoe_s %g un_s %g\n",
               l.d, l.E, l.mn, l.mt, r.d, r.E, r.mn, r.mt, (int)x, (int)wall, ustar, pstar, rho_s, rhoe_s, un_s);
#endif
    return from_nf(F, x);
}
#else
__device__ __forceinline__ Cons sphf_face(const Cons &Ul, const Cons &Ur, double gamma, bool x,
                                          bool wall, double &pface)
{
    const ConsN Uo = cgf_state(to_nf(Ul, x), to_nf(Ur, x), gamma, wall);
    pface = cons_to_prim(from_nf(Uo, x), gamma).p;
    return from_nf(cons_flux_n(Uo, gamma, x, false), x);
}
#endif

__device__ __forceinline__ Cons sphf_corrected(const Cons &U, const Cons &Fhi, double Ahi,
                                               const Cons &Flo, double Alo, double hv)
{
    Cons r;   // U += -hdtV*(F_hi*A_hi - F_lo*A_lo), unsplit_fluxes.py:447-471
    r.d = U.d + (-hv * (Fhi.d * Ahi - Flo.d * Alo));
    r.E = U.E + (-hv * (Fhi.E * Ahi - Flo.E * Alo));
    r.mx = U.mx + (-hv * (Fhi.mx * Ahi - Flo.mx * Alo));
    r.my = U.my + (-hv * (Fhi.my * Ahi - Flo.my * Alo));
    return r;
}

// method_compute_timestep takes its minimum over the WHOLE array (simulation.py:284-288), and
// on this grid a ghost cell's Lx, Ly are its own: the minimum over the ghost cells that take
// their value from interior cell (i, j) -- the cell's new state with the signs of the boundary
// rule, the lengths of the ghost cell -- folded into `cfl`, so that the next dt needs neither a
// ghost fill nor a reduction launch
template <bool FAC>
__device__ __forceinline__ double sphf_ghost_cfl(const Cons &U, double gamma, const Geom &g,
                                                 const FP &P, const SphAt<FAC> &GA, int i, int j, double cfl)
{
    const int ng = g.ng;
    const bool ei = (i < g.ilo + ng) || (i > g.ihi - ng), ej = (j < g.jlo + ng) || (j > g.jhi - ng);
    if (!ei && !ej) return cfl;
    for (int a = -1; a < 2 * ng; a++) {       // a = -1: the cell's own row
        int r = i;
        if (a >= 0) {
            r = a < ng ? a : g.ihi + 1 + (a - ng);
            if (!ei || bc_src(P.mr, r, g.ilo, g.ihi) != i) continue;
        }
        for (int b = -1; b < 2 * ng; b++) {
            int c = j;
            if (b >= 0) {
                c = b < ng ? b : g.jhi + 1 + (b - ng);
                if (!ej || bc_src(P.mc, c, g.jlo, g.jhi) != j) continue;
            }
            if (a < 0 && b < 0) continue;
            const unsigned sd = (r < g.ilo ? 1u : 0u) | (r > g.ihi ? 2u : 0u) | (c < g.jlo ? 4u : 0u) |
                                (c > g.jhi ? 8u : 0u);
            Cons Ug = U;
            Ug.d = odd_sides(P.odd & sd) ? -Ug.d : Ug.d;
            Ug.E = odd_sides((P.odd >> 4) & sd) ? -Ug.E : Ug.E;
            Ug.mx = odd_sides((P.odd >> 8) & sd) ? -Ug.mx : Ug.mx;
            Ug.my = odd_sides((P.odd >> 12) & sd) ? -Ug.my : Ug.my;
            cfl = fmin(cfl, cfl_cell(Ug, gamma, GA.Lx(r, c), GA.Ly(r, c)));
        }
    }
    return cfl;
}

