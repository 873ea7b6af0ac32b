// SphericalPolar pieces shared by the two one-launch kernels of the compressible step on such a
// grid: the 2-d LDS tile kernel (comp_fused.hip: k_ctu_fused_sph) and the row-marching kernel
// (comp_sph_wave.hip: k_sph_wave).  Included inside namespace pyro::PYRO_NS after fused_common.h.
#pragma once

struct SphG {   // kernel-side geometry (pyrohip_state_set_geometry)
    const double *Lx, *Ly, *Ax, *Ay, *V, *dlAx, *dlAy, *x2d, *sint, *sinb, *sinc;
    double xmin;
    // FAC instance (round 6): the geometry rebuilt from its 1-d factors -- rowf: A D F G Ly dlogAx x
    // (stride qxp), colf: B C E T (stride qyp) -- with the bits of the planes (include/pyrohip.h:
    // pyrohip_geom; mesh/patch.py checks the factorisation when it hands the tables over).  The
    // plane-reading instance moved 312 B of fabric traffic per cell update for 64 algorithmic ones
    // (profiles/r05_sph2048_pmc.json): eight planes, most of them read for two or three cells.
    const double *rowf, *colf;
    int qxp, qyp;
};
// value of a geometry array at (row r, column c) -- FAC: from the factors
template <bool FAC> struct SphAt {
    const SphG &G; int p; double dx;
    __device__ __forceinline__ double rf(int k, int r) const { return G.rowf[k * G.qxp + r]; }
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
__device__ __forceinline__ Cons sphf_face(const Cons &Ul, const Cons &Ur, double gamma, bool x,
                                          bool wall, double &pface)
{
    const ConsN Uo = cgf_state(to_nf(Ul, x), to_nf(Ur, x), gamma, wall);
    pface = cons_to_prim(from_nf(Uo, x), gamma).p;
    return from_nf(cons_flux_n(Uo, gamma, x, false), x);
}

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

