/*
 * pyro_oracle.c -- CPU restatement of pyro2's per-timestep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pyro2_amd/) may
 * import, link or call this file; it is the checker used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * It follows the reference statement-by-statement, including the reference's
 * "full scratch array, zero outside the region that was written" semantics,
 * so stage arrays can be compared index-for-index (ghost cells included)
 * with arrays dumped from the reference itself (oracle/gen_golden.py).
 * Parity is PINNED: tests/test_oracle_golden.py checks this file against
 * (i) stage dumps of the reference run in this container and (ii) the
 * reference's own golden files (smooth_0040, sod_x_0076, quad_unsplit_0606,
 * mg_poisson_dirichlet), exported to tests/golden/ by oracle/gen_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the reference is NumPy/numba without FMA
 * contraction and we want bit-level agreement.
 *
 * Array conventions (same as the reference, pyro/mesh/patch.py:450-452):
 *   scalar fields  double a[qx][qy]        C order, j (y) fastest
 *   vector fields  double U[qx][qy][nvar]  C order, component fastest
 *   conserved order: density(0) energy(1) x-momentum(2) y-momentum(3)
 *                    (pyro/compressible/simulation.py:223-226)
 *   primitive order: rho(0) u(1) v(2) p(3)   (simulation.py:38-41)
 * Python grid attrs: ilo=ng, ihi=ng+nx-1 (inclusive).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDENS 0
#define IENER 1
#define IXMOM 2
#define IYMOM 3
#define IRHO 0
#define IU 1
#define IV 2
#define IP 3

/* BC codes shared with include/pyrohip.h */
#define BC_OUTFLOW 0      /* also homogeneous neumann */
#define BC_REFLECT_EVEN 1
#define BC_REFLECT_ODD 2  /* also homogeneous dirichlet */
#define BC_PERIODIC 3
#define BC_HSE 5          /* compressible/BC.py "hse" (y sides only)   */
#define BC_AMBIENT 6      /* compressible/BC.py "ambient" (yr only)    */
#define BC_RAMP 7         /* compressible/BC.py "ramp" (xl, yl, yr)    */
#define BC_CONST 8        /* ghost cells = a constant ("moving_lid" of  */
                          /* incompressible_viscous/BC.py; 0 in the MG) */

typedef struct {
    int nx, ny, ng;
    double dx, dy;
    double gamma;
    int limiter;         /* 0 none, 1 MC2, 2 MC4 */
    int use_flattening;
    double z0, z1, delta;
    double cvisc;
    double grav;
    double small_dens;
    /* bc[var][side]: var in conserved order, side = xl,xr,yl,yr.  Only used
       for the source-term ghost fill when grav != 0. */
    int bc[4][4];
    /* domain-decomposition hook: when non-zero the artificial
       viscosity coefficient on the upper x / y boundary face is computed
       instead of being left at the reference's 0 (interface.py:366-367),
       because that face is an interior interface of the global grid. */
    int avisc_xhi_interior, avisc_yhi_interior;
    /* 0 = HLLC, 1 = CGF (compressible.riemann); solid_*: bc_is_solid of the
       mesh boundaries, used by CGF only (riemann.py:274-286) */
    int riemann;
    int solid_xl, solid_xr, solid_yl, solid_yr;
    /* sponge (compressible/simulation.py:164-184, 427-441) */
    int do_sponge;
    double sponge_rho_begin, sponge_rho_full, sponge_timescale;
    /* problem source of the heating / plume / convection problems
       (compressible/problems/{heating,plume,convection}.py source_terms):
       S[energy] += rho * heat_rate * heat_prof[i,j]; heat_prof: (qx,qy) profile
       exp(-(dist/r)^2) on the whole grid incl. ghost coordinates, or NULL */
    double heat_rate;
    const double *heat_prof;
    /* SphericalPolar grid (mesh/patch.py:242-312; x = r, y = theta), NULL for
       Cartesian2d: see orc_geom */
    const struct orc_geom_s *geom;
} orc_comp_params;

/* geometry arrays of patch.SphericalPolar, all (qx,qy) and evaluated by the
   caller with the reference's NumPy expressions (cos / sin / tan included), so
   that restatement and product see the same bits.  sint/sinb/sinc: (qy) sines
   of artificial_viscosity (interface.py:345-347): sin((j +- 1/2 - ng) dy + ymin)
   and sin((j - ng) dy + ymin). */
typedef struct orc_geom_s {
    const double *Lx, *Ly, *Ax, *Ay, *V, *dlogAx, *dlogAy, *x2d;
    const double *sint, *sinb, *sinc;
    double xmin, ymin;
} orc_geom;

/* optional stage outputs; any pointer may be NULL */
typedef struct {
    double *q;                       /* (qx,qy,4) */
    double *xi;                      /* (qx,qy)   */
    double *ldx, *ldy;               /* (qx,qy,4) */
    double *Uxl0, *Uxr0, *Uyl0, *Uyr0; /* states before transverse corr. */
    double *FxT, *FyT;               /* transverse Riemann fluxes */
    double *Uxl, *Uxr, *Uyl, *Uyr;   /* corrected states */
    double *Fx0, *Fy0;               /* final Riemann fluxes before avisc */
    double *avx, *avy;               /* (qx,qy) */
    double *Fx, *Fy;                 /* fluxes incl. artificial viscosity */
} orc_comp_stages;

static double *zalloc(size_t n) { return (double *)calloc(n, sizeof(double)); }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
/* Python's builtin max(a, b) / min(a, b) on scalars, as the njit kernels of
   the reference call them: the first argument wins unless the second one is
   strictly larger / smaller, so max(smallc, nan) = smallc but max(nan, small)
   = nan.  Only matters in ghost faces whose states are not physical (the
   reflect-odd density ghosts of inputs.sedov.spherical). */
static inline double pymax(double a, double b) { return b > a ? b : a; }
static inline double pymin(double a, double b) { return b < a ? b : a; }

/* ------------------------------------------------------------------ */
/* a1: ghost fill, pyro/mesh/array_indexer.py:150-274                  */
/* operates on component n of a (qx,qy,nvar) array; bc = xl,xr,yl,yr   */
/* ------------------------------------------------------------------ */
void orc_fill_ghost(double *a, int nx, int ny, int ng, int nvar, int n,
                    const int *bc)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
#define A(i, j) a[((size_t)(i) * qy + (j)) * nvar + n]
    /* -x : all j including ghosts (array_indexer.py:163-190) */
    for (int i = 0; i < ilo; i++)
        for (int j = 0; j < qy; j++) {
            switch (bc[0]) {
            case BC_OUTFLOW: A(i, j) = A(ilo, j); break;
            case BC_REFLECT_EVEN: A(i, j) = A(2 * ng - i - 1, j); break;
            case BC_REFLECT_ODD: A(i, j) = -A(2 * ng - i - 1, j); break;
            case BC_PERIODIC: A(i, j) = A(ihi - ng + i + 1, j); break;
            }
        }
    /* +x (array_indexer.py:192-221) */
    for (int k = 0; k < ng; k++)
        for (int j = 0; j < qy; j++) {
            int i = ihi + 1 + k;
            switch (bc[1]) {
            case BC_OUTFLOW: A(i, j) = A(ihi, j); break;
            case BC_REFLECT_EVEN: A(i, j) = A(ihi - k, j); break;
            case BC_REFLECT_ODD: A(i, j) = -A(ihi - k, j); break;
            case BC_PERIODIC: A(i, j) = A(i - ihi - 1 + ng, j); break;
            }
        }
    /* -y : all i including the x ghosts just filled (:223-244) */
    for (int i = 0; i < qx; i++)
        for (int j = 0; j < jlo; j++) {
            switch (bc[2]) {
            case BC_HSE: /* BC.py:54-62: plain copy for everything but energy */
            case BC_OUTFLOW: A(i, j) = A(i, jlo); break;
            case BC_REFLECT_EVEN: A(i, j) = A(i, 2 * ng - j - 1); break;
            case BC_REFLECT_ODD: A(i, j) = -A(i, 2 * ng - j - 1); break;
            case BC_PERIODIC: A(i, j) = A(i, jhi - ng + j + 1); break;
            }
        }
    /* +y (:246-274) */
    for (int i = 0; i < qx; i++)
        for (int k = 0; k < ng; k++) {
            int j = jhi + 1 + k;
            switch (bc[3]) {
            case BC_HSE: case BC_AMBIENT: /* BC.py:87-93, 159-160 */
            case BC_OUTFLOW: A(i, j) = A(i, jhi); break;
            case BC_REFLECT_EVEN: A(i, j) = A(i, jhi - k); break;
            case BC_REFLECT_ODD: A(i, j) = -A(i, jhi - k); break;
            case BC_PERIODIC: A(i, j) = A(i, j - jhi - 1 + ng); break;
            }
        }
#undef A
}

/* ------------------------------------------------------------------ */
/* a2: limiters, pyro/mesh/reconstruction.py:56-120                    */
/* in : a  (qx,qy) with element stride `as` (so a component of a       */
/*      (qx,qy,nvar) array can be passed)                              */
/* out: lda (qx,qy) contiguous, zero outside buf=2                     */
/* ------------------------------------------------------------------ */
static void limit2_(const double *a, int as, int nx, int ny, int ng,
                    int idir, double *lda)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    memset(lda, 0, sizeof(double) * qx * qy);
    const int di = (idir == 1) ? 1 : 0, dj = (idir == 1) ? 0 : 1;
#define AA(i, j) a[((size_t)(i) * qy + (j)) * as]
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            double ap = AA(i + di, j + dj), am = AA(i - di, j - dj),
                   a0 = AA(i, j);
            double dc = 0.5 * (ap - am);
            double dl = ap - a0;
            double dr = a0 - am;
            double d1 = 2.0 * (fabs(dl) < fabs(dr) ? dl : dr);
            double dt = (fabs(dc) < fabs(d1)) ? dc : d1;
            lda[(size_t)i * qy + j] = (dl * dr > 0.0) ? dt : 0.0;
        }
#undef AA
}

void orc_limit(const double *a, int as, int nx, int ny, int ng, int idir,
               int limiter, double *lda)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const int di = (idir == 1) ? 1 : 0, dj = (idir == 1) ? 0 : 1;
#define AA(i, j) a[((size_t)(i) * qy + (j)) * as]
    if (limiter == 0) { /* nolimit, reconstruction.py:56-66 */
        memset(lda, 0, sizeof(double) * qx * qy);
        for (int i = ilo - 2; i <= ihi + 2; i++)
            for (int j = jlo - 2; j <= jhi + 2; j++)
                lda[(size_t)i * qy + j] =
                    0.5 * (AA(i + di, j + dj) - AA(i - di, j - dj));
        return;
    }
    if (limiter == 1) {
        limit2_(a, as, nx, ny, ng, idir, lda);
        return;
    }
    /* limit4, reconstruction.py:94-120 */
    double *l2 = zalloc((size_t)qx * qy);
    limit2_(a, as, nx, ny, ng, idir, l2);
    memset(lda, 0, sizeof(double) * qx * qy);
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            double ap = AA(i + di, j + dj), am = AA(i - di, j - dj),
                   a0 = AA(i, j);
            double l2p = l2[(size_t)(i + di) * qy + (j + dj)];
            double l2m = l2[(size_t)(i - di) * qy + (j - dj)];
            double dc = (2. / 3.) * (ap - am - 0.25 * (l2p + l2m));
            double dl = ap - a0;
            double dr = a0 - am;
            double d1 = 2.0 * (fabs(dl) < fabs(dr) ? dl : dr);
            double dt = (fabs(dc) < fabs(d1)) ? dc : d1;
            lda[(size_t)i * qy + j] = (dl * dr > 0.0) ? dt : 0.0;
        }
    free(l2);
#undef AA
}

/* ------------------------------------------------------------------ */
/* a4-a6: advection step, pyro/advection/{interface,advective_fluxes,  */
/* simulation}.py.  a is (qx,qy) with ghost cells already filled.      */
/* stage outputs (may be NULL): ldx,ldy,ax,ay,Fx,Fy  all (qx,qy)       */
/* ------------------------------------------------------------------ */
void orc_adv_step(double *a, int nx, int ny, int ng, double dx, double dy,
                  double u, double v, double dt, int limiter, double *o_ldx,
                  double *o_ldy, double *o_ax, double *o_ay, double *o_Fx,
                  double *o_Fy)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    double *ldx = zalloc(N), *ldy = zalloc(N), *ax = zalloc(N),
           *ay = zalloc(N), *Fxt = zalloc(N), *Fyt = zalloc(N),
           *Fx = zalloc(N), *Fy = zalloc(N);
#define I2(i, j) ((size_t)(i) * qy + (j))
    /* interface.py:10-21 */
    double cx = u * dt / dx;
    double cy = v * dt / dy;
    orc_limit(a, 1, nx, ny, ng, 1, limiter, ldx);
    orc_limit(a, 1, nx, ny, ng, 2, limiter, ldy);
    /* interface.py:25-41, region buf=1 */
    for (int i = ilo - 1; i <= ihi + 1; i++)
        for (int j = jlo - 1; j <= jhi + 1; j++) {
            if (u < 0)
                ax[I2(i, j)] = a[I2(i, j)] - 0.5 * (1.0 + cx) * ldx[I2(i, j)];
            else
                ax[I2(i, j)] =
                    a[I2(i - 1, j)] + 0.5 * (1.0 - cx) * ldx[I2(i - 1, j)];
            if (v < 0)
                ay[I2(i, j)] = a[I2(i, j)] - 0.5 * (1.0 + cy) * ldy[I2(i, j)];
            else
                ay[I2(i, j)] =
                    a[I2(i, j - 1)] + 0.5 * (1.0 - cy) * ldy[I2(i, j - 1)];
        }
    /* advective_fluxes.py:62-63, full arrays */
    for (size_t k = 0; k < N; k++) {
        Fxt[k] = u * ax[k];
        Fyt[k] = v * ay[k];
    }
    /* advective_fluxes.py:71-90 */
    int mx = (u <= 0) ? 0 : -1;
    int my = (v <= 0) ? 0 : -1;
    double dtdx2 = 0.5 * dt / dx;
    double dtdy2 = 0.5 * dt / dy;
    for (int i = ilo - 1; i <= ihi + 1; i++)
        for (int j = jlo - 1; j <= jhi + 1; j++) {
            Fx[I2(i, j)] = u * (ax[I2(i, j)] - dtdy2 * (Fyt[I2(i + mx, j + 1)] -
                                                        Fyt[I2(i + mx, j)]));
            Fy[I2(i, j)] = v * (ay[I2(i, j)] - dtdx2 * (Fxt[I2(i + 1, j + my)] -
                                                        Fxt[I2(i, j + my)]));
        }
    /* simulation.py:63-80 */
    double dtdx = dt / dx;
    double dtdy = dt / dy;
    for (int i = ilo; i <= ihi; i++)
        for (int j = jlo; j <= jhi; j++)
            a[I2(i, j)] = a[I2(i, j)] +
                          dtdx * (Fx[I2(i, j)] - Fx[I2(i + 1, j)]) +
                          dtdy * (Fy[I2(i, j)] - Fy[I2(i, j + 1)]);
#undef I2
    if (o_ldx) memcpy(o_ldx, ldx, N * 8);
    if (o_ldy) memcpy(o_ldy, ldy, N * 8);
    if (o_ax) memcpy(o_ax, ax, N * 8);
    if (o_ay) memcpy(o_ay, ay, N * 8);
    if (o_Fx) memcpy(o_Fx, Fx, N * 8);
    if (o_Fy) memcpy(o_Fy, Fy, N * 8);
    free(ldx); free(ldy); free(ax); free(ay);
    free(Fxt); free(Fyt); free(Fx); free(Fy);
}

/* advection/simulation.py:38-54 */
double orc_adv_dt(double dx, double dy, double u, double v, double cfl)
{
    const double SMALL = 1.e-12;
    double xtmp = dx / dmax(fabs(u), SMALL);
    double ytmp = dy / dmax(fabs(v), SMALL);
    return cfl * dmin(xtmp, ytmp);
}

/* ------------------------------------------------------------------ */
/* a7: cons_to_prim / prim_to_cons, compressible/simulation.py:49-102  */
/* returns 0 ok, 1 if the interior positivity assert (:68-71) fails    */
/* ------------------------------------------------------------------ */
int orc_cons_to_prim(const double *U, int nx, int ny, int ng, double gamma,
                     double *q)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    int bad = 0;
    for (int i = 0; i < qx; i++)
        for (int j = 0; j < qy; j++) {
            const double *Uc = U + ((size_t)i * qy + j) * 4;
            double *qc = q + ((size_t)i * qy + j) * 4;
            double rho = Uc[IDENS];
            double u = 0.0, v = 0.0, e = 0.0;
            if (rho != 0.0) {
                u = Uc[IXMOM] / rho;
                v = Uc[IYMOM] / rho;
            }
            if (rho != 0.0)
                e = (Uc[IENER] - 0.5 * rho * (u * u + v * v)) / rho;
            qc[IRHO] = rho;
            qc[IU] = u;
            qc[IV] = v;
            qc[IP] = rho * e * (gamma - 1.0); /* eos.py:26 */
            if (i >= ng && i < ng + nx && j >= ng && j < ng + ny)
                if (!(e > 0.0 && rho > 0.0)) bad = 1;
        }
    return bad;
}

void orc_prim_to_cons(const double *q, size_t ncell, double gamma, double *U)
{
    for (size_t k = 0; k < ncell; k++) {
        const double *qc = q + k * 4;
        double *Uc = U + k * 4;
        Uc[IDENS] = qc[IRHO];
        Uc[IXMOM] = qc[IU] * Uc[IDENS];
        Uc[IYMOM] = qc[IV] * Uc[IDENS];
        double rhoe = qc[IP] / (gamma - 1.0); /* eos.py:72 */
        Uc[IENER] = rhoe + 0.5 * qc[IRHO] * (qc[IU] * qc[IU] + qc[IV] * qc[IV]);
    }
}

/* ------------------------------------------------------------------ */
/* a3: flattening, pyro/mesh/reconstruction.py:123-183                 */
/* ------------------------------------------------------------------ */
static void flatten_(const double *q, int nx, int ny, int ng, int idir,
                     double z0, double z1, double delta, double *xi)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    const double smallp = 1.e-10;
    double *z = zalloc(N), *t1 = zalloc(N), *t2 = zalloc(N);
    const int di = (idir == 1) ? 1 : 0, dj = (idir == 1) ? 0 : 1;
    const int ivel = (idir == 1) ? IU : IV;
#define Q(i, j, n) q[((size_t)(i) * qy + (j)) * 4 + (n)]
#define I2(i, j) ((size_t)(i) * qy + (j))
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            t1[I2(i, j)] = fabs(Q(i + di, j + dj, IP) - Q(i - di, j - dj, IP));
            t2[I2(i, j)] =
                fabs(Q(i + 2 * di, j + 2 * dj, IP) - Q(i - 2 * di, j - 2 * dj, IP));
        }
    for (size_t k = 0; k < N; k++) z[k] = t1[k] / dmax(t2[k], smallp);
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            t2[I2(i, j)] = t1[I2(i, j)] /
                           dmin(Q(i + di, j + dj, IP), Q(i - di, j - dj, IP));
            t1[I2(i, j)] = Q(i - di, j - dj, ivel) - Q(i + di, j + dj, ivel);
        }
    for (size_t k = 0; k < N; k++) {
        double x = dmin(1.0, dmax(0.0, 1.0 - (z[k] - z0) / (z1 - z0)));
        xi[k] = (t1[k] > 0.0 && t2[k] > delta) ? x : 1.0;
    }
#undef Q
#undef I2
    free(z); free(t1); free(t2);
}

void orc_flatten_multid(const double *q, int nx, int ny, int ng, double z0,
                        double z1, double delta, double *xi)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    double *xix = zalloc(N), *xiy = zalloc(N);
    flatten_(q, nx, ny, ng, 1, z0, z1, delta, xix);
    flatten_(q, nx, ny, ng, 2, z0, z1, delta, xiy);
    memset(xi, 0, N * 8);
#define Q(i, j, n) q[((size_t)(i) * qy + (j)) * 4 + (n)]
#define I2(i, j) ((size_t)(i) * qy + (j))
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            double px = (Q(i + 1, j, IP) - Q(i - 1, j, IP) > 0)
                            ? xix[I2(i - 1, j)]
                            : xix[I2(i + 1, j)];
            double py = (Q(i, j + 1, IP) - Q(i, j - 1, IP) > 0)
                            ? xiy[I2(i, j - 1)]
                            : xiy[I2(i, j + 1)];
            xi[I2(i, j)] =
                dmin(dmin(xix[I2(i, j)], px), dmin(xiy[I2(i, j)], py));
        }
#undef Q
#undef I2
    free(xix); free(xiy);
}

/* ------------------------------------------------------------------ */
/* a8: characteristic tracing, pyro/compressible/interface.py:5-236    */
/* q, dq: (qx,qy,4).  q_l, q_r: (qx,qy,4) zeroed here.                 */
/* ------------------------------------------------------------------ */
static void states_impl(int idir, int nx, int ny, int ng, double dx, const double *L,
                        const double *dloga, double dt, double gamma, const double *qv,
                        const double *dqv, double *q_l, double *q_r);
void orc_states(int idir, int nx, int ny, int ng, double dx, double dt,
                double gamma, const double *qv, const double *dqv,
                double *q_l, double *q_r)
{
    states_impl(idir, nx, ny, ng, dx, NULL, NULL, dt, gamma, qv, dqv, q_l, q_r);
}
/* L, dloga: the 2-d cell-size and dlog(area) arrays of a curvilinear grid
   (interface.py:106 dtdx = dt / dx with dx an array; :215-234), NULL: uniform */
static void states_impl(int idir, int nx, int ny, int ng, double dx, const double *L,
                        const double *dloga, double dt, double gamma, const double *qv,
                        const double *dqv, double *q_l, double *q_r)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    memset(q_l, 0, sizeof(double) * qx * qy * 4);
    memset(q_r, 0, sizeof(double) * qx * qy * 4);
    double dtdx = dt / dx;       /* interface.py:106 */
    double dtdx4 = 0.25 * dtdx;  /* :107 */
    double lvec[4][4], rvec[4][4], e_val[4], betal[4], betar[4];
    for (int i = ilo - 2; i < ihi + 2; i++)
        for (int j = jlo - 2; j < jhi + 2; j++) {
            if (L) { dtdx = dt / L[(size_t)i * qy + j]; dtdx4 = 0.25 * dtdx; }
            const double *dq = dqv + ((size_t)i * qy + j) * 4;
            const double *q = qv + ((size_t)i * qy + j) * 4;
            double cs = sqrt(gamma * q[IP] / q[IRHO]);
            memset(lvec, 0, sizeof lvec);
            memset(rvec, 0, sizeof rvec);
            if (idir == 1) {
                e_val[0] = q[IU] - cs; e_val[1] = q[IU];
                e_val[2] = q[IU];      e_val[3] = q[IU] + cs;
                lvec[0][1] = -0.5 * q[IRHO] / cs; lvec[0][3] = 0.5 / (cs * cs);
                lvec[1][0] = 1.0;                 lvec[1][3] = -1.0 / (cs * cs);
                lvec[2][2] = 1.0;
                lvec[3][1] = 0.5 * q[IRHO] / cs;  lvec[3][3] = 0.5 / (cs * cs);
                rvec[0][0] = 1.0; rvec[0][1] = -cs / q[IRHO]; rvec[0][3] = cs * cs;
                rvec[1][0] = 1.0;
                rvec[2][2] = 1.0;
                rvec[3][0] = 1.0; rvec[3][1] = cs / q[IRHO];  rvec[3][3] = cs * cs;
            } else {
                e_val[0] = q[IV] - cs; e_val[1] = q[IV];
                e_val[2] = q[IV];      e_val[3] = q[IV] + cs;
                lvec[0][2] = -0.5 * q[IRHO] / cs; lvec[0][3] = 0.5 / (cs * cs);
                lvec[1][0] = 1.0;                 lvec[1][3] = -1.0 / (cs * cs);
                lvec[2][1] = 1.0;
                lvec[3][2] = 0.5 * q[IRHO] / cs;  lvec[3][3] = 0.5 / (cs * cs);
                rvec[0][0] = 1.0; rvec[0][2] = -cs / q[IRHO]; rvec[0][3] = cs * cs;
                rvec[1][0] = 1.0;
                rvec[2][1] = 1.0;
                rvec[3][0] = 1.0; rvec[3][2] = cs / q[IRHO];  rvec[3][3] = cs * cs;
            }
            double *ql = (idir == 1) ? q_l + ((size_t)(i + 1) * qy + j) * 4
                                     : q_l + ((size_t)i * qy + (j + 1)) * 4;
            double *qr = q_r + ((size_t)i * qy + j) * 4;
            /* reference states, interface.py:174-191 */
            double factor = 0.5 * (1.0 - dtdx * pymax(e_val[3], 0.0));
            for (int m = 0; m < 4; m++) ql[m] = q[m] + factor * dq[m];
            factor = 0.5 * (1.0 + dtdx * pymin(e_val[0], 0.0));
            for (int m = 0; m < 4; m++) qr[m] = q[m] - factor * dq[m];
            /* :193-201 ; np.dot = in-order 4-term sum */
            for (int m = 0; m < 4; m++) {
                double asum = 0.0;
                for (int k = 0; k < 4; k++) asum += lvec[m][k] * dq[k];
                betal[m] = dtdx4 * (e_val[3] - e_val[m]) *
                           (copysign(1.0, e_val[m]) + 1.0) * asum;
                betar[m] = dtdx4 * (e_val[0] - e_val[m]) *
                           (1.0 - copysign(1.0, e_val[m])) * asum;
            }
            /* :203-213 */
            for (int m = 0; m < 4; m++) {
                double sum_l = 0.0, sum_r = 0.0;
                for (int k = 0; k < 4; k++) {
                    sum_l += betal[k] * rvec[k][m];
                    sum_r += betar[k] * rvec[k][m];
                }
                ql[m] = ql[m] + sum_l;
                qr[m] = qr[m] + sum_r;
            }
            /* geometric source (:216-234); vanishes for Cartesian: dloga = 0 */
            if (dloga) {
                const double rho_source = -0.5 * dt * dloga[(size_t)i * qy + j] * q[IRHO] *
                                          q[idir == 1 ? IU : IV];
                ql[IRHO] += rho_source;
                qr[IRHO] += rho_source;
                ql[IP] += rho_source * cs * cs;
                qr[IP] += rho_source * cs * cs;
            }
        }
}

/* ------------------------------------------------------------------ */
/* a10: HLLC, pyro/compressible/riemann.py:596-860, 1104-1179          */
/* ------------------------------------------------------------------ */
static void estimate_wave_speed(double rho_l, double u_l, double p_l,
                                double c_l, double rho_r, double u_r,
                                double p_r, double c_r, double gamma,
                                double *S_l, double *S_r)
{
    double p_max = pymax(p_l, p_r);
    double p_min = pymin(p_l, p_r);
    double Q = p_max / p_min;
    double rho_avg = 0.5 * (rho_l + rho_r);
    double c_avg = 0.5 * (c_l + c_r);
    double factor = rho_avg * c_avg;
    double pstar = 0.5 * (p_l + p_r) + 0.5 * (u_l - u_r) * factor;
    double ustar = 0.5 * (u_l + u_r) + 0.5 * (p_l - p_r) / factor;
    if (Q > 2 && (pstar < p_min || pstar > p_max)) {
        if (pstar < p_min) { /* two-rarefaction, riemann.py:626-638 */
            double z = (gamma - 1.0) / (2.0 * gamma);
            double p_lr = pow(p_l / p_r, z);
            ustar = (p_lr * u_l / c_l + u_r / c_r +
                     2.0 * (p_lr - 1.0) / (gamma - 1.0)) /
                    (p_lr / c_l + 1.0 / c_r);
            pstar = 0.5 * (p_l * pow(1.0 + (gamma - 1.0) * (u_l - ustar) /
                                               (2.0 * c_l),
                                     1.0 / z) +
                           p_r * pow(1.0 + (gamma - 1.0) * (ustar - u_r) /
                                               (2.0 * c_r),
                                     1.0 / z));
        } else { /* two-shock, :640-658 */
            double A_r = 2.0 / ((gamma + 1.0) * rho_r);
            double B_r = p_r * (gamma - 1.0) / (gamma + 1.0);
            double A_l = 2.0 / ((gamma + 1.0) * rho_l);
            double B_l = p_l * (gamma - 1.0) / (gamma + 1.0);
            double p_guess = pymax(0.0, pstar);
            double g_l = sqrt(A_l / (p_guess + B_l));
            double g_r = sqrt(A_r / (p_guess + B_r));
            pstar = (g_l * p_l + g_r * p_r - (u_r - u_l)) / (g_l + g_r);
            ustar = 0.5 * (u_l + u_r) +
                    0.5 * ((pstar - p_r) * g_r - (pstar - p_l) * g_l);
        }
    }
    (void)ustar;
    if (pstar <= p_l)
        *S_l = u_l - c_l;
    else
        *S_l = u_l - c_l * sqrt(1.0 + ((gamma + 1.0) / (2.0 * gamma)) *
                                          (pstar / p_l - 1.0));
    if (pstar <= p_r)
        *S_r = u_r + c_r;
    else /* sic: (gamma+1)/(2/gamma), riemann.py:675 */
        *S_r = u_r + c_r * sqrt(1.0 + ((gamma + 1.0) / (2.0 / gamma)) *
                                          (pstar / p_r - 1.0));
}

/* SphericalPolar: consFlux leaves the pressure out of the momentum flux
   (riemann.py:1156, 1171) and riemann_flux(return_cons=True) also hands back the
   CGF interface state (:1092-1096): g_sph / g_cgf_state are set by
   orc_comp_step around riemann_dispatch */
static int g_sph = 0;
static double *g_cgf_state = NULL;
static void cons_flux(int idir, double gamma, const double *Us, double *F)
{
    double u = 0.0, v = 0.0;
    if (Us[IDENS] != 0.0) {
        u = Us[IXMOM] / Us[IDENS];
        v = Us[IYMOM] / Us[IDENS];
    }
    double p = (Us[IENER] - 0.5 * Us[IDENS] * (u * u + v * v)) * (gamma - 1.0);
    if (idir == 1) {
        F[IDENS] = Us[IDENS] * u;
        F[IXMOM] = Us[IXMOM] * u;
        if (!g_sph) F[IXMOM] += p;
        F[IYMOM] = Us[IYMOM] * u;
        F[IENER] = (Us[IENER] + p) * u;
    } else {
        F[IDENS] = Us[IDENS] * v;
        F[IXMOM] = Us[IXMOM] * v;
        F[IYMOM] = Us[IYMOM] * v;
        if (!g_sph) F[IYMOM] += p;
        F[IENER] = (Us[IENER] + p) * v;
    }
}

void orc_riemann_hllc(int idir, int nx, int ny, int ng, double gamma,
                      const double *U_l, const double *U_r, double *F)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    const double smallc = 1.e-10, smallp = 1.e-10;
    memset(F, 0, sizeof(double) * qx * qy * 4);
    double U_state[4];
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            const double *Ul = U_l + ((size_t)i * qy + j) * 4;
            const double *Ur = U_r + ((size_t)i * qy + j) * 4;
            double *Fc = F + ((size_t)i * qy + j) * 4;
            double rho_l = Ul[IDENS];
            double un_l, ut_l;
            if (idir == 1) {
                un_l = Ul[IXMOM] / rho_l;
                ut_l = Ul[IYMOM] / rho_l;
            } else {
                un_l = Ul[IYMOM] / rho_l;
                ut_l = Ul[IXMOM] / rho_l;
            }
            double rhoe_l = Ul[IENER] - 0.5 * rho_l * (un_l * un_l + ut_l * ut_l);
            double p_l = rhoe_l * (gamma - 1.0);
            p_l = pymax(p_l, smallp);
            double rho_r = Ur[IDENS];
            double un_r, ut_r;
            if (idir == 1) {
                un_r = Ur[IXMOM] / rho_r;
                ut_r = Ur[IYMOM] / rho_r;
            } else {
                un_r = Ur[IYMOM] / rho_r;
                ut_r = Ur[IXMOM] / rho_r;
            }
            double rhoe_r = Ur[IENER] - 0.5 * rho_r * (un_r * un_r + ut_r * ut_r);
            double p_r = rhoe_r * (gamma - 1.0);
            p_r = pymax(p_r, smallp);
            double c_l = pymax(smallc, sqrt(gamma * p_l / rho_l));
            double c_r = pymax(smallc, sqrt(gamma * p_r / rho_r));
            double S_l, S_r;
            estimate_wave_speed(rho_l, un_l, p_l, c_l, rho_r, un_r, p_r, c_r,
                                gamma, &S_l, &S_r);
            double S_c = (p_r - p_l + rho_l * un_l * (S_l - un_l) -
                          rho_r * un_r * (S_r - un_r)) /
                         (rho_l * (S_l - un_l) - rho_r * (S_r - un_r));
            if (S_r <= 0.0) {
                cons_flux(idir, gamma, Ur, Fc);
            } else if (S_c <= 0.0 && 0.0 < S_r) {
                double f = rho_r * (S_r - un_r) / (S_r - S_c);
                U_state[IDENS] = f;
                if (idir == 1) {
                    U_state[IXMOM] = f * S_c;
                    U_state[IYMOM] = f * ut_r;
                } else {
                    U_state[IXMOM] = f * ut_r;
                    U_state[IYMOM] = f * S_c;
                }
                U_state[IENER] =
                    f * (Ur[IENER] / rho_r +
                         (S_c - un_r) * (S_c + p_r / (rho_r * (S_r - un_r))));
                cons_flux(idir, gamma, Ur, Fc);
                for (int n = 0; n < 4; n++)
                    Fc[n] = Fc[n] + S_r * (U_state[n] - Ur[n]);
            } else if (S_l < 0.0 && 0.0 < S_c) {
                double f = rho_l * (S_l - un_l) / (S_l - S_c);
                U_state[IDENS] = f;
                if (idir == 1) {
                    U_state[IXMOM] = f * S_c;
                    U_state[IYMOM] = f * ut_l;
                } else {
                    U_state[IXMOM] = f * ut_l;
                    U_state[IYMOM] = f * S_c;
                }
                U_state[IENER] =
                    f * (Ul[IENER] / rho_l +
                         (S_c - un_l) * (S_c + p_l / (rho_l * (S_l - un_l))));
                cons_flux(idir, gamma, Ul, Fc);
                for (int n = 0; n < 4; n++)
                    Fc[n] = Fc[n] + S_l * (U_state[n] - Ul[n]);
            } else {
                cons_flux(idir, gamma, Ul, Fc);
            }
        }
}


/* ------------------------------------------------------------------ */
/* riemann_cgf, pyro/compressible/riemann.py:8-310 + consFlux          */
/* (riemann_flux :1083-1090).  Solid-wall quirk: the njit-local ihi is */
/* ng+nx, so "i == ihi + 1" never fires.          */
/* ------------------------------------------------------------------ */
/* `x**2` on a scalar: numba compiles it to x*x (the real reference), the
   interpreted identity-njit shim used by gen_golden.py evaluates it with
   libm pow(x, 2.0), which is not always the correctly rounded square.
   orc_set_scalar_pow(1) selects the shim's arithmetic so that goldens
   generated through the shim can be pinned bit for bit; default 0. */
static int g_scalar_pow = 0;
void orc_set_scalar_pow(int on) { g_scalar_pow = on; }
static volatile double g_two = 2.0; /* volatile: gcc would fold pow(x, 2.0) into x*x */
static inline double sq_ref(double x) { return g_scalar_pow ? pow(x, g_two) : x * x; }
#define SQ(x) sq_ref(x)
void orc_riemann_cgf(int idir, int nx, int ny, int ng, double gamma, int lower_solid,
                     int upper_solid, const double *U_l, const double *U_r, double *F)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    const double smallc = 1.e-10, smallrho = 1.e-10, smallp = 1.e-10;
    (void)upper_solid;
    memset(F, 0, sizeof(double) * qx * qy * 4);
    if (g_cgf_state) memset(g_cgf_state, 0, sizeof(double) * qx * qy * 4);
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            const double *Ul = U_l + ((size_t)i * qy + j) * 4;
            const double *Ur = U_r + ((size_t)i * qy + j) * 4;
            double rho_l = Ul[IDENS], un_l, ut_l;
            if (idir == 1) { un_l = Ul[IXMOM] / rho_l; ut_l = Ul[IYMOM] / rho_l; }
            else           { un_l = Ul[IYMOM] / rho_l; ut_l = Ul[IXMOM] / rho_l; }
            double rhoe_l = Ul[IENER] - 0.5 * rho_l * (SQ(un_l) + SQ(ut_l));
            double p_l = pymax(rhoe_l * (gamma - 1.0), smallp);
            double rho_r = Ur[IDENS], un_r, ut_r;
            if (idir == 1) { un_r = Ur[IXMOM] / rho_r; ut_r = Ur[IYMOM] / rho_r; }
            else           { un_r = Ur[IYMOM] / rho_r; ut_r = Ur[IXMOM] / rho_r; }
            double rhoe_r = Ur[IENER] - 0.5 * rho_r * (SQ(un_r) + SQ(ut_r));
            double p_r = pymax(rhoe_r * (gamma - 1.0), smallp);
            double W_l = pymax(smallrho * smallc, sqrt(gamma * p_l * rho_l));
            double W_r = pymax(smallrho * smallc, sqrt(gamma * p_r * rho_r));
            double c_l = pymax(smallc, sqrt(gamma * p_l / rho_l));
            double c_r = pymax(smallc, sqrt(gamma * p_r / rho_r));
            double pstar = (W_l * p_r + W_r * p_l + W_l * W_r * (un_l - un_r)) / (W_l + W_r);
            pstar = pymax(pstar, smallp);
            double ustar = (W_l * un_l + W_r * un_r + (p_l - p_r)) / (W_l + W_r);
            double rhostar_l = rho_l + (pstar - p_l) / SQ(c_l);
            double rhostar_r = rho_r + (pstar - p_r) / SQ(c_r);
            double rhoestar_l = rhoe_l + (pstar - p_l) * (rhoe_l / rho_l + p_l / rho_l) / SQ(c_l);
            double rhoestar_r = rhoe_r + (pstar - p_r) * (rhoe_r / rho_r + p_r / rho_r) / SQ(c_r);
            double cstar_l = pymax(smallc, sqrt(gamma * pstar / rhostar_l));
            double cstar_r = pymax(smallc, sqrt(gamma * pstar / rhostar_r));
            double rho_s, un_s, ut_s, p_s, rhoe_s;
            if (ustar > 0.0) {
                ut_s = ut_l;
                double lambda_l = un_l - c_l, lambdastar_l = ustar - cstar_l;
                if (pstar > p_l) {
                    double sigma = (lambda_l + lambdastar_l) / 2.0;
                    if (sigma > 0.0) { rho_s = rho_l; un_s = un_l; p_s = p_l; rhoe_s = rhoe_l; }
                    else { rho_s = rhostar_l; un_s = ustar; p_s = pstar; rhoe_s = rhoestar_l; }
                } else {
                    if (lambda_l < 0.0 && lambdastar_l < 0.0) {
                        rho_s = rhostar_l; un_s = ustar; p_s = pstar; rhoe_s = rhoestar_l;
                    } else if (lambda_l > 0.0 && lambdastar_l > 0.0) {
                        rho_s = rho_l; un_s = un_l; p_s = p_l; rhoe_s = rhoe_l;
                    } else {
                        double alpha = lambda_l / (lambda_l - lambdastar_l);
                        rho_s = alpha * rhostar_l + (1.0 - alpha) * rho_l;
                        un_s = alpha * ustar + (1.0 - alpha) * un_l;
                        p_s = alpha * pstar + (1.0 - alpha) * p_l;
                        rhoe_s = alpha * rhoestar_l + (1.0 - alpha) * rhoe_l;
                    }
                }
            } else if (ustar < 0) {
                ut_s = ut_r;
                double lambda_r = un_r + c_r, lambdastar_r = ustar + cstar_r;
                if (pstar > p_r) {
                    double sigma = (lambda_r + lambdastar_r) / 2.0;
                    if (sigma > 0.0) { rho_s = rhostar_r; un_s = ustar; p_s = pstar; rhoe_s = rhoestar_r; }
                    else { rho_s = rho_r; un_s = un_r; p_s = p_r; rhoe_s = rhoe_r; }
                } else {
                    if (lambda_r < 0.0 && lambdastar_r < 0.0) {
                        rho_s = rho_r; un_s = un_r; p_s = p_r; rhoe_s = rhoe_r;
                    } else if (lambda_r > 0.0 && lambdastar_r > 0.0) {
                        rho_s = rhostar_r; un_s = ustar; p_s = pstar; rhoe_s = rhoestar_r;
                    } else {
                        double alpha = lambda_r / (lambda_r - lambdastar_r);
                        rho_s = alpha * rhostar_r + (1.0 - alpha) * rho_r;
                        un_s = alpha * ustar + (1.0 - alpha) * un_r;
                        p_s = alpha * pstar + (1.0 - alpha) * p_r;
                        rhoe_s = alpha * rhoestar_r + (1.0 - alpha) * rhoe_r;
                    }
                }
            } else {
                rho_s = 0.5 * (rhostar_l + rhostar_r);
                un_s = ustar;
                ut_s = 0.5 * (ut_l + ut_r);
                p_s = pstar;
                rhoe_s = 0.5 * (rhoestar_l + rhoestar_r);
            }
            (void)p_s;
            if (idir == 1) { if (i == ilo && lower_solid == 1) un_s = 0.0; }
            else           { if (j == jlo && lower_solid == 1) un_s = 0.0; }
            double Uo[4];
            Uo[IDENS] = rho_s;
            if (idir == 1) { Uo[IXMOM] = rho_s * un_s; Uo[IYMOM] = rho_s * ut_s; }
            else           { Uo[IXMOM] = rho_s * ut_s; Uo[IYMOM] = rho_s * un_s; }
            Uo[IENER] = rhoe_s + 0.5 * rho_s * (SQ(un_s) + SQ(ut_s));
            if (g_cgf_state) memcpy(g_cgf_state + ((size_t)i * qy + j) * 4, Uo, 32);
            cons_flux(idir, gamma, Uo, F + ((size_t)i * qy + j) * 4);
        }
}

/* riemann_hllc_lowspeed ("HLLC_lm"), compressible/riemann.py:863-1020: HLLC
   with the low-Mach pressure fix  p* = (p_l+p_r)/2 + phi/2 (...),
   phi = chi (2 - chi), chi = min(1, max|v| / max c)                          */
void orc_riemann_hllc_lm(int idir, int nx, int ny, int ng, double gamma, const double *U_l,
                         const double *U_r, double *F)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    const double smallc = 1.e-10, smallp = 1.e-10;
    const int iun = (idir == 1) ? IXMOM : IYMOM, iut = (idir == 1) ? IYMOM : IXMOM;
    memset(F, 0, sizeof(double) * qx * qy * 4);
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            const double *Ul = U_l + ((size_t)i * qy + j) * 4, *Ur = U_r + ((size_t)i * qy + j) * 4;
            double *Fc = F + ((size_t)i * qy + j) * 4;
            const double rho_l = Ul[IDENS], un_l = Ul[iun] / rho_l, ut_l = Ul[iut] / rho_l;
            const double rhoe_l = Ul[IENER] - 0.5 * rho_l * (SQ(un_l) + SQ(ut_l));
            const double p_l = pymax(rhoe_l * (gamma - 1.0), smallp);
            const double rho_r = Ur[IDENS], un_r = Ur[iun] / rho_r, ut_r = Ur[iut] / rho_r;
            const double rhoe_r = Ur[IENER] - 0.5 * rho_r * (SQ(un_r) + SQ(ut_r));
            const double p_r = pymax(rhoe_r * (gamma - 1.0), smallp);
            const double c_l = pymax(smallc, sqrt(gamma * p_l / rho_l));
            const double c_r = pymax(smallc, sqrt(gamma * p_r / rho_r));
            double S_l, S_r;
            estimate_wave_speed(rho_l, un_l, p_l, c_l, rho_r, un_r, p_r, c_r, gamma, &S_l, &S_r);
            const double S_c = (p_r - p_l + rho_l * un_l * (S_l - un_l) - rho_r * un_r * (S_r - un_r)) /
                               (rho_l * (S_l - un_l) - rho_r * (S_r - un_r));
            double D[4] = {0.0, 0.0, 0.0, 0.0};
            D[iun] = 1.0;
            D[IENER] = S_c;
            double F_l[4], F_r[4];
            cons_flux(idir, gamma, Ul, F_l);
            cons_flux(idir, gamma, Ur, F_r);
            const double vmag_l = sqrt(SQ(un_l) + SQ(ut_l)), vmag_r = sqrt(SQ(un_r) + SQ(ut_r));
            const double cs_max = pymax(c_l, c_r);
            const double chi = pymin(1.0, pymax(vmag_l, vmag_r) / cs_max);
            const double phi = chi * (2.0 - chi);
            const double pstar = 0.5 * (p_l + p_r) +
                                 0.5 * phi * (rho_l * (S_l - un_l) * (S_c - un_l) +
                                              rho_r * (S_r - un_r) * (S_c - un_r));
            for (int n = 0; n < 4; n++) {
                if (S_r <= 0.0) Fc[n] = F_r[n];
                else if (S_c <= 0.0 && 0.0 < S_r)
                    Fc[n] = (S_c * (S_r * Ur[n] - F_r[n]) + S_r * pstar * D[n]) / (S_r - S_c);
                else if (S_l < 0.0 && 0.0 < S_c)
                    Fc[n] = (S_c * (S_l * Ul[n] - F_l[n]) + S_l * pstar * D[n]) / (S_l - S_c);
                else Fc[n] = F_l[n];
            }
        }
}

#undef SQ
static void riemann_dispatch(const orc_comp_params *P, int idir, const double *U_l,
                             const double *U_r, double *F)
{
    if (P->riemann == 2)
        orc_riemann_hllc_lm(idir, P->nx, P->ny, P->ng, P->gamma, U_l, U_r, F);
    else if (P->riemann == 1)
        orc_riemann_cgf(idir, P->nx, P->ny, P->ng, P->gamma,
                        idir == 1 ? P->solid_xl : P->solid_yl,
                        idir == 1 ? P->solid_xr : P->solid_yr, U_l, U_r, F);
    else
        orc_riemann_hllc(idir, P->nx, P->ny, P->ng, P->gamma, U_l, U_r, F);
}

/* ------------------------------------------------------------------ */
/* a11: artificial viscosity, compressible/interface.py:239-378        */
/* u, v: components IU, IV of q (qx,qy,4)                              */
/* ------------------------------------------------------------------ */
void orc_artificial_viscosity(int nx, int ny, int ng, double dx, double dy,
                              double cvisc, const double *q, int xhi_interior,
                              int yhi_interior, double *avx, double *avy)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    const size_t N = (size_t)qx * qy;
    double *divU = zalloc(N);
    memset(avx, 0, N * 8);
    memset(avy, 0, N * 8);
#define UU(i, j) q[((size_t)(i) * qy + (j)) * 4 + IU]
#define VV(i, j) q[((size_t)(i) * qy + (j)) * 4 + IV]
#define I2(i, j) ((size_t)(i) * qy + (j))
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            double ur = 0.5 * (UU(i, j) + UU(i, j - 1));
            double ul = 0.5 * (UU(i - 1, j) + UU(i - 1, j - 1));
            double vt = 0.5 * (VV(i, j) + VV(i - 1, j));
            double vb = 0.5 * (VV(i, j - 1) + VV(i - 1, j - 1));
            double ux = (ur - ul) / dx;
            double vy = (vt - vb) / dy;
            divU[I2(i, j)] = ux + vy;
        }
    /* interface.py:366-376: i in [ilo,ihi), j in [jlo,jhi).  With the
       decomposition hooks the x (y) face loop runs one face further. */
    for (int i = ilo; i < ihi + (xhi_interior ? 1 : 0); i++)
        for (int j = jlo; j < jhi; j++) {
            double divU_x = 0.5 * (divU[I2(i, j)] + divU[I2(i, j + 1)]);
            avx[I2(i, j)] = cvisc * dmax(-divU_x * dx, 0.0);
        }
    for (int i = ilo; i < ihi; i++)
        for (int j = jlo; j < jhi + (yhi_interior ? 1 : 0); j++) {
            double divU_y = 0.5 * (divU[I2(i, j)] + divU[I2(i + 1, j)]);
            avy[I2(i, j)] = cvisc * dmax(-divU_y * dy, 0.0);
        }
#undef UU
#undef VV
#undef I2
    free(divU);
}

/* artificial_viscosity on a SphericalPolar grid, interface.py:331-376 */
static void avisc_sph(int nx, int ny, int ng, double dx, double dy, double cvisc,
                      const double *q, const orc_geom *G, double *avx, double *avy)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    const size_t N = (size_t)qx * qy;
    double *divU = zalloc(N);
    memset(avx, 0, N * 8);
    memset(avy, 0, N * 8);
#define UU(i, j) q[((size_t)(i) * qy + (j)) * 4 + IU]
#define VV(i, j) q[((size_t)(i) * qy + (j)) * 4 + IV]
#define I2(i, j) ((size_t)(i) * qy + (j))
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            const double rr = (i + 0.5 - ng) * dx + G->xmin;
            const double rl = (i - 0.5 - ng) * dx + G->xmin;
            const double rc = (i - ng) * dx + G->xmin;
            const double ur = 0.5 * (UU(i, j) + UU(i, j - 1));
            const double ul = 0.5 * (UU(i - 1, j) + UU(i - 1, j - 1));
            const double ux = (ur * rr * rr - ul * rl * rl) / (rc * rc * dx);
            const double sint = G->sint[j], sinb = G->sinb[j], sinc = G->sinc[j];
            double vy;
            if (sinc == 0.0) vy = 0.0;
            else {
                const double vt = 0.5 * (VV(i, j) + VV(i - 1, j));
                const double vb = 0.5 * (VV(i, j - 1) + VV(i - 1, j - 1));
                vy = (sint * vt - sinb * vb) / (rc * sinc * dy);
            }
            divU[I2(i, j)] = ux + vy;
        }
    for (int i = ilo; i < ihi; i++)
        for (int j = jlo; j < jhi; j++) {
            const double divU_x = 0.5 * (divU[I2(i, j)] + divU[I2(i, j + 1)]);
            const double divU_y = 0.5 * (divU[I2(i, j)] + divU[I2(i + 1, j)]);
            avx[I2(i, j)] = cvisc * dmax(-divU_x * G->Lx[I2(i, j)], 0.0);
            avy[I2(i, j)] = cvisc * dmax(-divU_y * G->Ly[I2(i, j)], 0.0);
        }
#undef UU
#undef VV
#undef I2
    free(divU);
}

/* get_external_sources on a SphericalPolar grid (compressible/simulation.py:
   117-124, 135-147): radial gravity and the geometric terms */
static void ext_sources_sph(const double *U, const double *U_old, size_t ncell, double grav,
                            double dt, double *S, const double *x2d)
{
    memset(S, 0, ncell * 4 * 8);
    for (size_t k = 0; k < ncell; k++) {
        const double *Uc = U + k * 4;
        double *Sc = S + k * 4;
        Sc[IXMOM] = Uc[IDENS] * grav;
        if (!U_old) Sc[IENER] = Uc[IXMOM] * grav;
        else {
            const double S_old_xmom = U_old[k * 4 + IDENS] * grav;
            const double xmom_new = Uc[IXMOM] + 0.5 * dt * (Sc[IXMOM] - S_old_xmom);
            Sc[IENER] = xmom_new * grav;
        }
        Sc[IXMOM] += Uc[IYMOM] * Uc[IYMOM] / (Uc[IDENS] * x2d[k]);
        Sc[IYMOM] += -Uc[IXMOM] * Uc[IYMOM] / Uc[IDENS];
    }
}

/* method_compute_timestep with the grid's Lx, Ly arrays (simulation.py:284-288) */
double orc_comp_dt_geom(const double *U, int nx, int ny, int ng, const double *Lx,
                        const double *Ly, double gamma, double cfl)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    double xmin = INFINITY, ymin = INFINITY;
    for (size_t k = 0; k < (size_t)qx * qy; k++) {
        const double *Uc = U + k * 4;
        double dens = Uc[IDENS];
        double u = Uc[IXMOM] / dens;
        double v = Uc[IYMOM] / dens;
        double e = (Uc[IENER] - 0.5 * dens * (u * u + v * v)) / dens;
        double p = dens * e * (gamma - 1.0);
        double cs = sqrt(gamma * p / dens);
        double xt = Lx[k] / (fabs(u) + cs);
        double yt = Ly[k] / (fabs(v) + cs);
        if (xt < xmin) xmin = xt;
        if (yt < ymin) ymin = yt;
    }
    return cfl * dmin(xmin, ymin);
}

/* compressible/simulation.py:105-161, Cartesian branch; problem_source of the
   form rho * rate * prof (:156-159) */
static void ext_sources_h(const double *U, const double *U_old, size_t ncell, double grav,
                          double dt, double *S, double rate, const double *prof)
{
    memset(S, 0, ncell * 4 * 8);
    for (size_t k = 0; k < ncell; k++) {
        const double *Uc = U + k * 4;
        double *Sc = S + k * 4;
        if (!U_old) {
            Sc[IYMOM] = Uc[IDENS] * grav;
            Sc[IENER] = Uc[IYMOM] * grav;
        } else {
            Sc[IYMOM] = Uc[IDENS] * grav;
            double S_old_ymom = U_old[k * 4 + IDENS] * grav;
            double ymom_new = Uc[IYMOM] + 0.5 * dt * (Sc[IYMOM] - S_old_ymom);
            Sc[IENER] = ymom_new * grav;
        }
        if (prof) Sc[IENER] += Uc[IDENS] * rate * prof[k];
    }
}

/* ------------------------------------------------------------------ */
/* f2: CellCenterData2d.fill_BC_all for the compressible state with the */
/* user boundaries of compressible/BC.py:21-176 ("hse", "ambient").      */
/* Variables are filled one after the other in registration order       */
/* (density, energy, x-momentum, y-momentum; patch.py:575-624), so the   */
/* hse energy fill sees the x ghosts of the momenta as the PREVIOUS      */
/* fill left them.  ambient = rho, u, v, p.                              */
/* ------------------------------------------------------------------ */
#define U4(a, i, j, n) a[((size_t)(i) * qy + (j)) * 4 + (n)]
void orc_comp_fill_bc(double *U, int nx, int ny, int ng, const int *bc /*[4][4]*/,
                      double gamma, double grav, double dy, const double *ambient)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int jlo = ng, jhi = ng + ny - 1;
    for (int n = 0; n < 4; n++) {
        const int *b = bc + 4 * n;
        orc_fill_ghost(U, nx, ny, ng, 4, n, b);
        if (n == IENER && (b[2] == BC_HSE || b[3] == BC_HSE)) {
            for (int side = 0; side < 2; side++) {
                if (b[2 + side] != BC_HSE) continue;
                const int jb = side ? jhi : jlo;
                for (int i = 0; i < qx; i++) {
                    /* BC.py:64-84 (ylb), 95-115 (yrb) */
                    double dens_base = U4(U, i, jb, IDENS);
                    double xm = U4(U, i, jb, IXMOM), ym = U4(U, i, jb, IYMOM);
                    double ke_base = 0.5 * (xm * xm + ym * ym) / dens_base;
                    double eint_base = (U4(U, i, jb, IENER) - ke_base) / dens_base;
                    double pres_base = dens_base * eint_base * (gamma - 1.0); /* eos.py pres */
                    for (int k = 1; k <= ng; k++) {
                        int j = side ? jhi + k : jlo - k;
                        double pres_next = side ? pres_base + grav * dens_base * dy
                                                : pres_base - grav * dens_base * dy;
                        double rhoe = pres_next / (gamma - 1.0);
                        U4(U, i, j, IENER) = rhoe + ke_base;
                        pres_base = pres_next;
                    }
                }
            }
        }
        if (b[3] == BC_AMBIENT) { /* BC.py:147-176 */
            double val;
            if (n == IDENS) val = ambient[0];
            else if (n == IXMOM) val = ambient[0] * ambient[1];
            else if (n == IYMOM) val = ambient[0] * ambient[2];
            else {
                double ke = 0.5 * ambient[0] * (ambient[1] * ambient[1] + ambient[2] * ambient[2]);
                val = ambient[3] / (gamma - 1.0) + ke;
            }
            for (int i = 0; i < qx; i++)
                for (int j = jhi + 1; j < qy; j++) U4(U, i, j, n) = val;
        }
    }
}

/* "ramp" boundary of the double Mach reflection problem, BC.py:178-296, as
   part of fill_BC_all: per variable the standard fill (which knows nothing
   about "ramp"), then xlb, ylb, yrb.  x: cell centres (qx); cxoff = 0.5 dx
   sqrt(3); post/pre: inflow values per variable; sfd/sfu: shock front of the
   ng ghost rows above the upper boundary at the current time (evaluated by
   the caller with the reference's math.* expressions). */
void orc_comp_fill_bc_ramp(double *U, int nx, int ny, int ng, const int *bc,
                           const double *x, double cxoff, const double *post,
                           const double *pre, const double *sfd, const double *sfu)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, jlo = ng, jhi = ng + ny - 1;
    for (int n = 0; n < 4; n++) {
        const int *b = bc + 4 * n;
        orc_fill_ghost(U, nx, ny, ng, 4, n, b);
        if (b[0] == BC_RAMP)                     /* :186-191 */
            for (int i = ilo - 1; i >= 0; i--)
                for (int j = 0; j < qy; j++) U4(U, i, j, n) = post[n];
        if (b[2] == BC_RAMP) {                   /* :199-214 */
            int jj = 0;
            for (int j = jlo - 1; j >= 0; j--, jj++)
                for (int i = 0; i < qx; i++) {
                    if (x[i] < 1.0 / 6.0) U4(U, i, j, n) = post[n];
                    else if (n == IYMOM) U4(U, i, j, n) = -1.0 * U4(U, i, jlo + jj, n);
                    else U4(U, i, j, n) = U4(U, i, jlo + jj, n);
                }
        }
        if (b[3] == BC_RAMP)                     /* :224-246 */
            for (int k = 0; k < ng; k++) {
                const int j = jhi + 1 + k;
                const double sf[2] = {sfd[k], sfu[k]};
                for (int i = 0; i < qx; i++) {
                    const double cx[2] = {x[i] - cxoff, x[i] + cxoff};
                    double v = 0.0;
                    for (int s = 0; s < 2; s++)
                        for (int c = 0; c < 2; c++)
                            v = v + 0.25 * ((cx[c] < sf[s]) ? post[n] : pre[n]);
                    U4(U, i, j, n) = v;
                }
            }
    }
}
#undef U4

/* ------------------------------------------------------------------ */
/* a12: CFL time step, compressible/simulation.py:267-288 +            */
/* derives.py:19-25,59; min over the FULL array including ghosts       */
/* ------------------------------------------------------------------ */
double orc_comp_dt(const double *U, int nx, int ny, int ng, double dx,
                   double dy, double gamma, double cfl)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    double xmin = INFINITY, ymin = INFINITY;
    for (size_t k = 0; k < (size_t)qx * qy; k++) {
        const double *Uc = U + k * 4;
        double dens = Uc[IDENS];
        double u = Uc[IXMOM] / dens;
        double v = Uc[IYMOM] / dens;
        double e = (Uc[IENER] - 0.5 * dens * (u * u + v * v)) / dens;
        double p = dens * e * (gamma - 1.0);
        double cs = sqrt(gamma * p / dens);
        double xt = dx / (fabs(u) + cs);
        double yt = dy / (fabs(v) + cs);
        if (xt < xmin) xmin = xt;
        if (yt < ymin) ymin = yt;
    }
    return cfl * dmin(xmin, ymin);
}

/* ------------------------------------------------------------------ */
/* a9+a12: one compressible step, compressible/simulation.py:290-450   */
/* + unsplit_fluxes.py:134-549.  U: (qx,qy,4), ghost cells filled.     */
/* returns 0 ok, 1 = positivity assert would have fired                */
/* ------------------------------------------------------------------ */
int orc_comp_step(double *U, const orc_comp_params *P, double dt,
                  orc_comp_stages *st)
{
    const int nx = P->nx, ny = P->ny, ng = P->ng;
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    const double gamma = P->gamma, dx = P->dx, dy = P->dy;
    const orc_geom *G = P->geom;   /* SphericalPolar when set */
    int rc = 0;
#define U4(a, i, j, n) a[((size_t)(i) * qy + (j)) * 4 + (n)]
#define I2(i, j) ((size_t)(i) * qy + (j))

    /* clean_state, simulation.py:452-456 */
    for (int i = ilo; i <= ihi; i++)
        for (int j = jlo; j <= jhi; j++)
            U4(U, i, j, IDENS) = dmax(U4(U, i, j, IDENS), P->small_dens);

    double *q = zalloc(N * 4), *xi = zalloc(N), *ldx = zalloc(N * 4),
           *ldy = zalloc(N * 4), *tmp = zalloc(N);
    double *V_l = zalloc(N * 4), *V_r = zalloc(N * 4);
    double *Uxl = zalloc(N * 4), *Uxr = zalloc(N * 4), *Uyl = zalloc(N * 4),
           *Uyr = zalloc(N * 4);
    double *Fx = zalloc(N * 4), *Fy = zalloc(N * 4);
    double *avx = zalloc(N), *avy = zalloc(N);
    /* SphericalPolar: CGF interface states and their primitives (pressure) */
    double *Ux = NULL, *Uy = NULL, *qfx = NULL, *qfy = NULL;
    if (G) { Ux = zalloc(N * 4); Uy = zalloc(N * 4); qfx = zalloc(N * 4); qfy = zalloc(N * 4); }

    /* unsplit_fluxes.py:160-197 */
    rc |= orc_cons_to_prim(U, nx, ny, ng, gamma, q);
    if (P->use_flattening)
        orc_flatten_multid(q, nx, ny, ng, P->z0, P->z1, P->delta, xi);
    else
        for (size_t k = 0; k < N; k++) xi[k] = 1.0;
    for (int n = 0; n < 4; n++) {
        orc_limit(q + n, 4, nx, ny, ng, 1, P->limiter, tmp);
        for (size_t k = 0; k < N; k++) ldx[k * 4 + n] = xi[k] * tmp[k];
        orc_limit(q + n, 4, nx, ny, ng, 2, P->limiter, tmp);
        for (size_t k = 0; k < N; k++) ldy[k * 4 + n] = xi[k] * tmp[k];
    }
    if (st && st->q) memcpy(st->q, q, N * 32);
    if (st && st->xi) memcpy(st->xi, xi, N * 8);
    if (st && st->ldx) memcpy(st->ldx, ldx, N * 32);
    if (st && st->ldy) memcpy(st->ldy, ldy, N * 32);

    /* unsplit_fluxes.py:207-242 */
    states_impl(1, nx, ny, ng, dx, G ? G->Lx : NULL, G ? G->dlogAx : NULL, dt, gamma, q, ldx,
                V_l, V_r);
    orc_prim_to_cons(V_l, N, gamma, Uxl);
    orc_prim_to_cons(V_r, N, gamma, Uxr);
    states_impl(2, nx, ny, ng, dy, G ? G->Ly : NULL, G ? G->dlogAy : NULL, dt, gamma, q, ldy,
                V_l, V_r);
    orc_prim_to_cons(V_l, N, gamma, Uyl);
    orc_prim_to_cons(V_r, N, gamma, Uyr);

    /* apply_source_terms, unsplit_fluxes.py:247-330 */
    const int have_src = (P->grav != 0.0 || P->heat_prof != NULL || G != NULL);
    if (have_src) {
        double *S = zalloc(N * 4);
        if (G) ext_sources_sph(U, NULL, N, P->grav, dt, S, G->x2d);
        else ext_sources_h(U, NULL, N, P->grav, dt, S, P->heat_rate, P->heat_prof);
        for (int n = 0; n < 4; n++) orc_fill_ghost(S, nx, ny, ng, 4, n, P->bc[n]);
        const int comps[3] = {IXMOM, IYMOM, IENER};
        for (int c = 0; c < 3; c++) {
            int n = comps[c];
            for (int i = ilo - 1; i <= ihi + 1; i++)
                for (int j = jlo - 1; j <= jhi + 1; j++) {
                    U4(Uxl, i, j, n) += 0.5 * dt * U4(S, i - 1, j, n);
                    U4(Uxr, i, j, n) += 0.5 * dt * U4(S, i, j, n);
                    U4(Uyl, i, j, n) += 0.5 * dt * U4(S, i, j - 1, n);
                    U4(Uyr, i, j, n) += 0.5 * dt * U4(S, i, j, n);
                }
        }
        free(S);
    }
    if (st && st->Uxl0) memcpy(st->Uxl0, Uxl, N * 32);
    if (st && st->Uxr0) memcpy(st->Uxr0, Uxr, N * 32);
    if (st && st->Uyl0) memcpy(st->Uyl0, Uyl, N * 32);
    if (st && st->Uyr0) memcpy(st->Uyr0, Uyr, N * 32);

    /* apply_transverse_flux, unsplit_fluxes.py:333-494 */
    g_sph = (G != NULL);
    g_cgf_state = Ux;
    riemann_dispatch(P, 1, Uxl, Uxr, Fx);
    g_cgf_state = Uy;
    riemann_dispatch(P, 2, Uyl, Uyr, Fy);
    g_cgf_state = NULL;
    if (G) {   /* :421-423; the interior assert of cons_to_prim is not an issue on faces */
        orc_cons_to_prim(Ux, nx, ny, ng, gamma, qfx);
        orc_cons_to_prim(Uy, nx, ny, ng, gamma, qfy);
    }
    if (st && st->FxT) memcpy(st->FxT, Fx, N * 32);
    if (st && st->FyT) memcpy(st->FyT, Fy, N * 32);
    {
        const double hdt = 0.5 * dt;
        const double V = dx * dy;     /* patch.py:232 */
        const double hdtV = hdt / V;
        const double Ax = dy, Ay = dx; /* patch.py:219-222 */
        if (G) {   /* area / volume arrays and the pressure-gradient terms, :442-488 */
            for (int n = 0; n < 4; n++)
                for (int i = ilo - 2; i <= ihi + 1; i++)
                    for (int j = jlo - 2; j <= jhi + 1; j++) {
                        const double hv = hdt / G->V[I2(i, j)];
                        U4(Uxl, i, j, n) += -hv * (U4(Fy, i - 1, j + 1, n) * G->Ay[I2(i - 1, j + 1)] -
                                                   U4(Fy, i - 1, j, n) * G->Ay[I2(i - 1, j)]);
                        U4(Uxr, i, j, n) += -hv * (U4(Fy, i, j + 1, n) * G->Ay[I2(i, j + 1)] -
                                                   U4(Fy, i, j, n) * G->Ay[I2(i, j)]);
                        U4(Uyl, i, j, n) += -hv * (U4(Fx, i + 1, j - 1, n) * G->Ax[I2(i + 1, j - 1)] -
                                                   U4(Fx, i, j - 1, n) * G->Ax[I2(i, j - 1)]);
                        U4(Uyr, i, j, n) += -hv * (U4(Fx, i + 1, j, n) * G->Ax[I2(i + 1, j)] -
                                                   U4(Fx, i, j, n) * G->Ax[I2(i, j)]);
                    }
            for (int i = ilo - 2; i <= ihi + 1; i++)
                for (int j = jlo - 2; j <= jhi + 1; j++) {
                    U4(Uxl, i, j, IYMOM) += -hdt * (U4(qfy, i - 1, j + 1, IP) - U4(qfy, i - 1, j, IP)) /
                                            G->Ly[I2(i, j)];
                    U4(Uxr, i, j, IYMOM) += -hdt * (U4(qfy, i, j + 1, IP) - U4(qfy, i, j, IP)) /
                                            G->Ly[I2(i, j)];
                    U4(Uyl, i, j, IXMOM) += -hdt * (U4(qfx, i + 1, j - 1, IP) - U4(qfx, i, j - 1, IP)) /
                                            G->Lx[I2(i, j)];
                    U4(Uyr, i, j, IXMOM) += -hdt * (U4(qfx, i + 1, j, IP) - U4(qfx, i, j, IP)) /
                                            G->Lx[I2(i, j)];
                }
        } else
        for (int n = 0; n < 4; n++)
            for (int i = ilo - 2; i <= ihi + 1; i++)
                for (int j = jlo - 2; j <= jhi + 1; j++) {
                    U4(Uxl, i, j, n) += -hdtV * (U4(Fy, i - 1, j + 1, n) * Ay -
                                                 U4(Fy, i - 1, j, n) * Ay);
                    U4(Uxr, i, j, n) += -hdtV * (U4(Fy, i, j + 1, n) * Ay -
                                                 U4(Fy, i, j, n) * Ay);
                    U4(Uyl, i, j, n) += -hdtV * (U4(Fx, i + 1, j - 1, n) * Ax -
                                                 U4(Fx, i, j - 1, n) * Ax);
                    U4(Uyr, i, j, n) += -hdtV * (U4(Fx, i + 1, j, n) * Ax -
                                                 U4(Fx, i, j, n) * Ax);
                }
    }
    if (st && st->Uxl) memcpy(st->Uxl, Uxl, N * 32);
    if (st && st->Uxr) memcpy(st->Uxr, Uxr, N * 32);
    if (st && st->Uyl) memcpy(st->Uyl, Uyl, N * 32);
    if (st && st->Uyr) memcpy(st->Uyr, Uyr, N * 32);

    /* final Riemann solves, simulation.py:330-357 */
    g_cgf_state = Ux;
    riemann_dispatch(P, 1, Uxl, Uxr, Fx);
    g_cgf_state = Uy;
    riemann_dispatch(P, 2, Uyl, Uyr, Fy);
    g_cgf_state = NULL;
    g_sph = 0;
    if (G) {
        orc_cons_to_prim(Ux, nx, ny, ng, gamma, qfx);
        orc_cons_to_prim(Uy, nx, ny, ng, gamma, qfy);
    }
    if (st && st->Fx0) memcpy(st->Fx0, Fx, N * 32);
    if (st && st->Fy0) memcpy(st->Fy0, Fy, N * 32);

    /* artificial viscosity, simulation.py:361-365, unsplit_fluxes.py:525-547 */
    if (G) avisc_sph(nx, ny, ng, dx, dy, P->cvisc, q, G, avx, avy);
    else
    orc_artificial_viscosity(nx, ny, ng, dx, dy, P->cvisc, q,
                             P->avisc_xhi_interior, P->avisc_yhi_interior, avx,
                             avy);
    for (int n = 0; n < 4; n++)
        for (int i = ilo - 2; i <= ihi + 1; i++)
            for (int j = jlo - 2; j <= jhi + 1; j++) {
                U4(Fx, i, j, n) +=
                    avx[I2(i, j)] * (U4(U, i - 1, j, n) - U4(U, i, j, n));
                U4(Fy, i, j, n) +=
                    avy[I2(i, j)] * (U4(U, i, j - 1, n) - U4(U, i, j, n));
            }
    if (st && st->avx) memcpy(st->avx, avx, N * 8);
    if (st && st->avy) memcpy(st->avy, avy, N * 8);
    if (st && st->Fx) memcpy(st->Fx, Fx, N * 32);
    if (st && st->Fy) memcpy(st->Fy, Fy, N * 32);

    /* conservative update, simulation.py:367-384 */
    double *U_old = NULL;
    if (have_src) {
        U_old = zalloc(N * 4);
        memcpy(U_old, U, N * 32);
    }
    {
        const double dtdV = dt / (dx * dy);
        const double Ax = dy, Ay = dx;
        if (G) {   /* :375-398: area / volume arrays, then the pressure gradients */
            for (int n = 0; n < 4; n++)
                for (int i = ilo; i <= ihi; i++)
                    for (int j = jlo; j <= jhi; j++)
                        U4(U, i, j, n) += (dt / G->V[I2(i, j)]) *
                            (U4(Fx, i, j, n) * G->Ax[I2(i, j)] - U4(Fx, i + 1, j, n) * G->Ax[I2(i + 1, j)] +
                             U4(Fy, i, j, n) * G->Ay[I2(i, j)] - U4(Fy, i, j + 1, n) * G->Ay[I2(i, j + 1)]);
            for (int i = ilo; i <= ihi; i++)
                for (int j = jlo; j <= jhi; j++) {
                    U4(U, i, j, IXMOM) -= dt * (U4(qfx, i + 1, j, IP) - U4(qfx, i, j, IP)) / G->Lx[I2(i, j)];
                    U4(U, i, j, IYMOM) -= dt * (U4(qfy, i, j + 1, IP) - U4(qfy, i, j, IP)) / G->Ly[I2(i, j)];
                }
        } else
        for (int n = 0; n < 4; n++)
            for (int i = ilo; i <= ihi; i++)
                for (int j = jlo; j <= jhi; j++)
                    U4(U, i, j, n) +=
                        dtdV * (U4(Fx, i, j, n) * Ax - U4(Fx, i + 1, j, n) * Ax +
                                U4(Fy, i, j, n) * Ay - U4(Fy, i, j + 1, n) * Ay);
    }
    /* source predictor-corrector, simulation.py:406-423 */
    if (have_src) {
        double *S_old = zalloc(N * 4), *S_new = zalloc(N * 4);
        if (G) ext_sources_sph(U_old, NULL, N, P->grav, dt, S_old, G->x2d);
        else
        ext_sources_h(U_old, NULL, N, P->grav, dt, S_old, P->heat_rate, P->heat_prof);
        for (int n = 0; n < 4; n++)
            for (int i = ilo; i <= ihi; i++)
                for (int j = jlo; j <= jhi; j++)
                    U4(U, i, j, n) += dt * U4(S_old, i, j, n);
        if (G) ext_sources_sph(U, U_old, N, P->grav, dt, S_new, G->x2d);
        else
        ext_sources_h(U, U_old, N, P->grav, dt, S_new, P->heat_rate, P->heat_prof);
        for (int n = 0; n < 4; n++)
            for (int i = ilo; i <= ihi; i++)
                for (int j = jlo; j <= jhi; j++)
                    U4(U, i, j, n) +=
                        0.5 * dt * (U4(S_new, i, j, n) - U4(S_old, i, j, n));
        free(S_old); free(S_new); free(U_old);
    }
    /* sponge, simulation.py:164-184, 427-441: the WHOLE array incl. ghosts */
    if (P->do_sponge) {
        const double PI = 3.14159265358979323846;
        for (size_t k = 0; k < N; k++) {
            double *Uc = U + k * 4;
            double rho = Uc[IDENS], f;
            if (rho > P->sponge_rho_begin) f = 0.0;
            else if (rho < P->sponge_rho_full) f = 1.0;
            else f = 0.5 * (1.0 - cos(PI * (rho - P->sponge_rho_begin) /
                                      (P->sponge_rho_full - P->sponge_rho_begin)));
            double kappa = f / P->sponge_timescale;
            double xo = Uc[IXMOM], yo = Uc[IYMOM];
            Uc[IXMOM] = xo / (1.0 + dt * kappa);
            Uc[IYMOM] = yo / (1.0 + dt * kappa);
            double dke = 0.5 * ((Uc[IXMOM] * Uc[IXMOM] + Uc[IYMOM] * Uc[IYMOM]) -
                                (xo * xo + yo * yo)) / Uc[IDENS];
            Uc[IENER] += dke;
        }
    }
#undef U4
#undef I2
    free(q); free(xi); free(ldx); free(ldy); free(tmp); free(V_l); free(V_r);
    free(Uxl); free(Uxr); free(Uyl); free(Uyr); free(Fx); free(Fy);
    free(avx); free(avy);
    free(Ux); free(Uy); free(qfx); free(qfy);
    return rc;
}

/* ================================================================== */
/* Multigrid, pyro/multigrid/MG.py + pyro/mesh/patch.py:640-736        */
/* Levels are planar (n+2)x(n+2) arrays with ng = 1.                   */
/* MG BC codes: BC_REFLECT_ODD = dirichlet, BC_OUTFLOW = neumann,      */
/* BC_PERIODIC.  Inhomogeneous values (finest-level v only) follow     */
/* array_indexer.py:166-183,196-215: first ghost cell only.            */
/* ================================================================== */
#define MG_MAXLEV 20
typedef struct {
    int nlevels, nx;       /* finest is nx x nx */
    double xmin, xmax, ymin, ymax;
    double alpha, beta;
    int nsmooth, nsmooth_bottom;
    int bc[4];
    int n[MG_MAXLEV];
    double dx[MG_MAXLEV];
    double *v[MG_MAXLEV], *f[MG_MAXLEV], *r[MG_MAXLEV];
    double *bcval[4];      /* NULL or length n+2 on the finest level */
    double source_norm;
    int num_cycles;
    double relative_error, residual_error;
    int max_cycles;
} orc_mg;

orc_mg *orc_mg_create(int nx, double xmin, double xmax, double ymin,
                      double ymax, const int *bc, double alpha, double beta,
                      int nsmooth, int nsmooth_bottom)
{
    orc_mg *m = (orc_mg *)calloc(1, sizeof(orc_mg));
    m->nx = nx;
    m->xmin = xmin; m->xmax = xmax; m->ymin = ymin; m->ymax = ymax;
    m->alpha = alpha; m->beta = beta;
    m->nsmooth = nsmooth; m->nsmooth_bottom = nsmooth_bottom;
    m->max_cycles = 100;
    memcpy(m->bc, bc, sizeof(int) * 4);
    m->nlevels = (int)(log((double)nx) / log(2.0)); /* MG.py:207 */
    int nt = 2;
    for (int l = 0; l < m->nlevels; l++) {
        m->n[l] = nt;
        m->dx[l] = (xmax - xmin) / nt; /* patch.py:121 */
        size_t N = (size_t)(nt + 2) * (nt + 2);
        m->v[l] = zalloc(N); m->f[l] = zalloc(N); m->r[l] = zalloc(N);
        nt *= 2;
    }
    m->residual_error = 1.e33; m->relative_error = 1.e33;
    return m;
}

void orc_mg_free(orc_mg *m)
{
    for (int l = 0; l < m->nlevels; l++) { free(m->v[l]); free(m->f[l]); free(m->r[l]); }
    for (int s = 0; s < 4; s++) free(m->bcval[s]);
    free(m);
}

int orc_mg_nlevels(const orc_mg *m) { return m->nlevels; }
double *orc_mg_ptr(orc_mg *m, int level, int var)
{
    return var == 0 ? m->v[level] : var == 1 ? m->f[level] : m->r[level];
}
void orc_mg_set_bcval(orc_mg *m, int side, const double *vals)
{
    int n = m->n[m->nlevels - 1] + 2;
    free(m->bcval[side]);
    m->bcval[side] = (double *)malloc(sizeof(double) * n);
    memcpy(m->bcval[side], vals, sizeof(double) * n);
}
void orc_mg_set_max_cycles(orc_mg *m, int mc) { m->max_cycles = mc; }
double orc_mg_get_scalar(const orc_mg *m, int which)
{
    switch (which) {
    case 0: return m->source_norm;
    case 1: return (double)m->num_cycles;
    case 2: return m->relative_error;
    case 3: return m->residual_error;
    }
    return 0.0;
}

/* fill_BC for an ng=1 level array; vals[side] optional inhomogeneous data */
static void mg_fill_bc(double *a, int n, double dx, const int *bc,
                       double *const *vals)
{
    const int q = n + 2, lo = 1, hi = n;
#define A(i, j) a[(size_t)(i) * q + (j)]
    for (int j = 0; j < q; j++) {
        switch (bc[0]) {
        case BC_OUTFLOW:
            A(0, j) = (vals && vals[0]) ? A(lo, j) - dx * vals[0][j] : A(lo, j);
            break;
        case BC_REFLECT_ODD:
            A(0, j) = (vals && vals[0]) ? 2 * vals[0][j] - A(lo, j) : -A(lo, j);
            break;
        case BC_REFLECT_EVEN: A(0, j) = A(lo, j); break;
        case BC_PERIODIC: A(0, j) = A(hi, j); break;
        case BC_CONST: A(0, j) = 0.0; break;
        }
    }
    for (int j = 0; j < q; j++) {
        switch (bc[1]) {
        case BC_OUTFLOW:
            A(hi + 1, j) = (vals && vals[1]) ? A(hi, j) + dx * vals[1][j] : A(hi, j);
            break;
        case BC_REFLECT_ODD:
            A(hi + 1, j) = (vals && vals[1]) ? 2 * vals[1][j] - A(hi, j) : -A(hi, j);
            break;
        case BC_REFLECT_EVEN: A(hi + 1, j) = A(hi, j); break;
        case BC_PERIODIC: A(hi + 1, j) = A(lo, j); break;
        case BC_CONST: A(hi + 1, j) = 0.0; break;
        }
    }
    for (int i = 0; i < q; i++) {
        switch (bc[2]) {
        case BC_OUTFLOW:
            A(i, 0) = (vals && vals[2]) ? A(i, lo) - dx * vals[2][i] : A(i, lo);
            break;
        case BC_REFLECT_ODD:
            A(i, 0) = (vals && vals[2]) ? 2 * vals[2][i] - A(i, lo) : -A(i, lo);
            break;
        case BC_REFLECT_EVEN: A(i, 0) = A(i, lo); break;
        case BC_PERIODIC: A(i, 0) = A(i, hi); break;
        case BC_CONST: A(i, 0) = 0.0; break;
        }
    }
    for (int i = 0; i < q; i++) {
        switch (bc[3]) {
        case BC_OUTFLOW:
            A(i, hi + 1) = (vals && vals[3]) ? A(i, hi) + dx * vals[3][i] : A(i, hi);
            break;
        case BC_REFLECT_ODD:
            A(i, hi + 1) = (vals && vals[3]) ? 2 * vals[3][i] - A(i, hi) : -A(i, hi);
            break;
        case BC_REFLECT_EVEN: A(i, hi + 1) = A(i, hi); break;
        case BC_PERIODIC: A(i, hi + 1) = A(i, lo); break;
        case BC_CONST: A(i, hi + 1) = 0.0; break; /* BC.user on the MG variable "v" */
        }
    }
#undef A
}

void orc_mg_fill_bc_v(orc_mg *m, int level)
{
    double *const *vals = (level == m->nlevels - 1) ? m->bcval : NULL;
    mg_fill_bc(m->v[level], m->n[level], m->dx[level], m->bc, vals);
}

/* array_indexer.py:98-111 (plain left-to-right sum; NumPy uses pairwise
   summation, so norms agree to ~1e-15 relative, not bitwise) */
static double mg_norm(const double *a, int n, double dx)
{
    const int q = n + 2;
    double s = 0.0;
    for (int i = 1; i <= n; i++)
        for (int j = 1; j <= n; j++) s += a[(size_t)i * q + j] * a[(size_t)i * q + j];
    return sqrt(dx * dx * s);
}
double orc_mg_norm(orc_mg *m, int level, int var)
{
    return mg_norm(orc_mg_ptr(m, level, var), m->n[level], m->dx[level]);
}

/* MG.py:529-542 */
void orc_mg_residual(orc_mg *m, int level)
{
    const int n = m->n[level], q = n + 2;
    const double dx = m->dx[level], dx2 = dx * dx;
    const double *v = m->v[level], *f = m->f[level];
    double *r = m->r[level];
#define I(i, j) ((size_t)(i) * q + (j))
    for (int i = 1; i <= n; i++)
        for (int j = 1; j <= n; j++)
            r[I(i, j)] =
                f[I(i, j)] - m->alpha * v[I(i, j)] +
                m->beta * ((v[I(i - 1, j)] + v[I(i + 1, j)] - 2 * v[I(i, j)]) / dx2 +
                           (v[I(i, j - 1)] + v[I(i, j + 1)] - 2 * v[I(i, j)]) / dx2);
#undef I
}

/* MG.py:544-621 */
void orc_mg_smooth(orc_mg *m, int level, int nsmooth)
{
    const int n = m->n[level], q = n + 2;
    const double dx = m->dx[level];
    double *v = m->v[level];
    const double *f = m->f[level];
    orc_mg_fill_bc_v(m, level);
    const double xcoeff = m->beta / (dx * dx);
    const double ycoeff = m->beta / (dx * dx);
    const double denom = m->alpha + 2.0 * xcoeff + 2.0 * ycoeff;
    static const int grp[4][2] = {{0, 0}, {1, 1}, {1, 0}, {0, 1}};
#define I(i, j) ((size_t)(i) * q + (j))
    for (int it = 0; it < nsmooth; it++)
        for (int g = 0; g < 4; g++) {
            int ix = grp[g][0], iy = grp[g][1];
            for (int i = 1 + ix; i <= n; i += 2)
                for (int j = 1 + iy; j <= n; j += 2)
                    v[I(i, j)] = (f[I(i, j)] +
                                  xcoeff * (v[I(i + 1, j)] + v[I(i - 1, j)]) +
                                  ycoeff * (v[I(i, j + 1)] + v[I(i, j - 1)])) /
                                 denom;
            if (g == 1 || g == 3) orc_mg_fill_bc_v(m, level);
        }
#undef I
}

/* patch.py:640-676 : f_coarse.v() = restrict(r_fine).v() */
void orc_mg_restrict(orc_mg *m, int level /* fine */)
{
    const int nc = m->n[level - 1], qc = nc + 2, qf = m->n[level] + 2;
    const double *fd = m->r[level];
    double *cd = m->f[level - 1];
    for (int i = 0; i < nc; i++)
        for (int j = 0; j < nc; j++) {
            int fi = 1 + 2 * i, fj = 1 + 2 * j;
            cd[(size_t)(1 + i) * qc + (1 + j)] =
                0.25 * (fd[(size_t)fi * qf + fj] + fd[(size_t)(fi + 1) * qf + fj] +
                        fd[(size_t)fi * qf + fj + 1] +
                        fd[(size_t)(fi + 1) * qf + fj + 1]);
        }
}

/* patch.py:678-736 + MG.py:745-751: v_fine.v() += prolong(v_coarse).v() */
void orc_mg_prolong_add(orc_mg *m, int level /* fine */)
{
    const int nc = m->n[level - 1], qc = nc + 2, qf = m->n[level] + 2;
    const double *c = m->v[level - 1];
    double *v = m->v[level];
    for (int i = 1; i <= nc; i++)
        for (int j = 1; j <= nc; j++) {
            double c0 = c[(size_t)i * qc + j];
            double m_x = 0.5 * (c[(size_t)(i + 1) * qc + j] - c[(size_t)(i - 1) * qc + j]);
            double m_y = 0.5 * (c[(size_t)i * qc + j + 1] - c[(size_t)i * qc + j - 1]);
            int fi = 1 + 2 * (i - 1), fj = 1 + 2 * (j - 1);
            v[(size_t)fi * qf + fj] += c0 - 0.25 * m_x - 0.25 * m_y;
            v[(size_t)(fi + 1) * qf + fj] += c0 + 0.25 * m_x - 0.25 * m_y;
            v[(size_t)fi * qf + fj + 1] += c0 - 0.25 * m_x + 0.25 * m_y;
            v[(size_t)(fi + 1) * qf + fj + 1] += c0 + 0.25 * m_x + 0.25 * m_y;
        }
}

/* MG.py:699-778 */
void orc_mg_vcycle(orc_mg *m, int level)
{
    if (level > 0) {
        orc_mg_smooth(m, level, m->nsmooth);
        orc_mg_residual(m, level);
        orc_mg_restrict(m, level);
        orc_mg_vcycle(m, level - 1);
        orc_mg_prolong_add(m, level);
        orc_mg_fill_bc_v(m, level);
        orc_mg_smooth(m, level, m->nsmooth);
    } else {
        orc_mg_smooth(m, level, m->nsmooth_bottom);
        orc_mg_fill_bc_v(m, level);
    }
}

void orc_mg_init_rhs_norm(orc_mg *m)
{
    m->source_norm = orc_mg_norm(m, m->nlevels - 1, 1);
}

/* MG.py:623-697 */
void orc_mg_solve(orc_mg *m, double rtol)
{
    const int L = m->nlevels - 1;
    const int n = m->n[L], q = n + 2;
    const size_t N = (size_t)q * q;
    double *old_phi = (double *)malloc(N * 8);
    memcpy(old_phi, m->v[L], N * 8);
    double residual_error = 1.e33, relative_error = 1.e33;
    int cycle = 1;
    while (residual_error > rtol && cycle <= m->max_cycles) {
        for (int l = 0; l < L; l++) {
            size_t Nl = (size_t)(m->n[l] + 2) * (m->n[l] + 2);
            memset(m->v[l], 0, Nl * 8);
        }
        orc_mg_vcycle(m, L);
        /* diff = (v - old)/(v + small); relative_error = diff.norm() */
        double s = 0.0;
        for (int i = 1; i <= n; i++)
            for (int j = 1; j <= n; j++) {
                size_t k = (size_t)i * q + j;
                double d = (m->v[L][k] - old_phi[k]) / (m->v[L][k] + 1.e-16);
                s += d * d;
            }
        relative_error = sqrt(m->dx[L] * m->dx[L] * s);
        memcpy(old_phi, m->v[L], N * 8);
        orc_mg_residual(m, L);
        double rn = orc_mg_norm(m, L, 2);
        residual_error = (m->source_norm != 0.0) ? rn / m->source_norm : rn;
        cycle++;
    }
    m->num_cycles = cycle - 1;
    m->relative_error = relative_error;
    m->residual_error = residual_error;
    orc_mg_fill_bc_v(m, L);
    free(old_phi);
}

/* ================================================================== */
/* Variable-coefficient multigrid  div(eta grad phi) = f               */
/* pyro/multigrid/variable_coeff_MG.py:23-213, edge_coeffs.py:1-54     */
/* (SURVEY 8 row f1).  Uses the level arrays of orc_mg plus per-level  */
/* edge coefficients eta_x[i,j] = eta_{i-1/2,j}, eta_y[i,j] = eta_{i,j-1/2} */
/* ================================================================== */
typedef struct {
    orc_mg *mg;
    double *c[MG_MAXLEV];     /* cell-centred coefficient, ghost filled */
    double *ex[MG_MAXLEV], *ey[MG_MAXLEV];
    int cbc[4];
    /* general mode (general_MG.py): alpha phi + div(beta grad phi) + gamma . grad phi = f;
       c / ex / ey then hold beta */
    int general;
    double *a[MG_MAXLEV], *gx[MG_MAXLEV], *gy[MG_MAXLEV];
} orc_vcmg;

orc_vcmg *orc_vcmg_create(int nx, double xmin, double xmax, double ymin, double ymax,
                          const int *bc, const int *coeffs_bc, const double *coeffs,
                          int nsmooth, int nsmooth_bottom)
{
    orc_vcmg *V = (orc_vcmg *)calloc(1, sizeof(orc_vcmg));
    V->mg = orc_mg_create(nx, xmin, xmax, ymin, ymax, bc, 0.0, 0.0, nsmooth, nsmooth_bottom);
    memcpy(V->cbc, coeffs_bc, sizeof(int) * 4);
    orc_mg *m = V->mg;
    const int L = m->nlevels - 1;
    for (int l = 0; l <= L; l++) {
        size_t N = (size_t)(m->n[l] + 2) * (m->n[l] + 2);
        V->c[l] = zalloc(N); V->ex[l] = zalloc(N); V->ey[l] = zalloc(N);
    }
    {   /* finest: c.v() = coeffs.v(); fill_BC; EdgeCoeffs (edge_coeffs.py:8-27) */
        const int n = m->n[L], q = n + 2;
        for (int i = 1; i <= n; i++)
            for (int j = 1; j <= n; j++) V->c[L][(size_t)i * q + j] = coeffs[(size_t)i * q + j];
        mg_fill_bc(V->c[L], n, m->dx[L], V->cbc, NULL);
        const double dx2 = m->dx[L] * m->dx[L];
        for (int i = 1; i <= n + 1; i++)
            for (int j = 1; j <= n + 1; j++) {
                V->ex[L][(size_t)i * q + j] =
                    0.5 * (V->c[L][(size_t)(i - 1) * q + j] + V->c[L][(size_t)i * q + j]);
                V->ey[L][(size_t)i * q + j] =
                    0.5 * (V->c[L][(size_t)i * q + j - 1] + V->c[L][(size_t)i * q + j]);
            }
        for (size_t k = 0; k < (size_t)q * q; k++) { V->ex[L][k] /= dx2; V->ey[L][k] /= dx2; }
    }
    for (int l = L - 1; l >= 0; l--) {   /* variable_coeff_MG.py:86-99 */
        const int nc = m->n[l], qc = nc + 2, qf = m->n[l + 1] + 2;
        const double *fc = V->c[l + 1];
        for (int i = 0; i < nc; i++)
            for (int j = 0; j < nc; j++) {
                int fi = 1 + 2 * i, fj = 1 + 2 * j;
                V->c[l][(size_t)(1 + i) * qc + 1 + j] =
                    0.25 * (fc[(size_t)fi * qf + fj] + fc[(size_t)(fi + 1) * qf + fj] +
                            fc[(size_t)fi * qf + fj + 1] + fc[(size_t)(fi + 1) * qf + fj + 1]);
            }
        mg_fill_bc(V->c[l], nc, m->dx[l], V->cbc, NULL);
        /* EdgeCoeffs.restrict, edge_coeffs.py:29-54 */
        const double *fx = V->ex[l + 1], *fy = V->ey[l + 1];
        const double fdx2 = m->dx[l + 1] * m->dx[l + 1], cdx2 = m->dx[l] * m->dx[l];
        for (int i = 0; i <= nc; i++)        /* x edges: i in [ilo, ihi+1], j interior */
            for (int j = 0; j < nc; j++) {
                int fi = 1 + 2 * i, fj = 1 + 2 * j;
                double e = 0.5 * (fx[(size_t)fi * qf + fj] + fx[(size_t)fi * qf + fj + 1]);
                V->ex[l][(size_t)(1 + i) * qc + 1 + j] = e * fdx2 / cdx2;
            }
        for (int i = 0; i < nc; i++)         /* y edges: j in [jlo, jhi+1], i interior */
            for (int j = 0; j <= nc; j++) {
                int fi = 1 + 2 * i, fj = 1 + 2 * j;
                double e = 0.5 * (fy[(size_t)fi * qf + fj] + fy[(size_t)(fi + 1) * qf + fj]);
                V->ey[l][(size_t)(1 + i) * qc + 1 + j] = e * fdx2 / cdx2;
            }
    }
    return V;
}

/* general_MG.py:44-105: alpha, gamma_x, gamma_y (cell centred, own BCs) are
   restricted down the hierarchy like beta; beta goes to the edges as in the
   variable-coefficient solver.  cbcs: BC codes of alpha, beta, gamma_x, gamma_y */
orc_vcmg *orc_genmg_create(int nx, double xmin, double xmax, double ymin, double ymax,
                           const int *bc, const int *cbcs, const double *alpha,
                           const double *beta, const double *gamma_x, const double *gamma_y,
                           int nsmooth, int nsmooth_bottom)
{
    orc_vcmg *V = orc_vcmg_create(nx, xmin, xmax, ymin, ymax, bc, cbcs + 4, beta, nsmooth,
                                  nsmooth_bottom);
    orc_mg *m = V->mg;
    const int L = m->nlevels - 1;
    V->general = 1;
    const double *src[3] = {alpha, gamma_x, gamma_y};
    const int *sbc[3] = {cbcs, cbcs + 8, cbcs + 12};
    double **dst[3] = {V->a, V->gx, V->gy};
    for (int w = 0; w < 3; w++) {
        for (int l = 0; l <= L; l++) dst[w][l] = zalloc((size_t)(m->n[l] + 2) * (m->n[l] + 2));
        const int n = m->n[L], q = n + 2;
        for (int i = 1; i <= n; i++)
            for (int j = 1; j <= n; j++) dst[w][L][(size_t)i * q + j] = src[w][(size_t)i * q + j];
        mg_fill_bc(dst[w][L], n, m->dx[L], sbc[w], NULL);
        for (int l = L - 1; l >= 0; l--) {
            const int nc = m->n[l], qc = nc + 2, qf = m->n[l + 1] + 2;
            const double *fc = dst[w][l + 1];
            for (int i = 0; i < nc; i++)
                for (int j = 0; j < nc; j++) {
                    const int fi = 1 + 2 * i, fj = 1 + 2 * j;
                    dst[w][l][(size_t)(1 + i) * qc + 1 + j] =
                        0.25 * (fc[(size_t)fi * qf + fj] + fc[(size_t)(fi + 1) * qf + fj] +
                                fc[(size_t)fi * qf + fj + 1] + fc[(size_t)(fi + 1) * qf + fj + 1]);
                }
            mg_fill_bc(dst[w][l], nc, m->dx[l], sbc[w], NULL);
        }
    }
    return V;
}

void orc_vcmg_free(orc_vcmg *V)
{
    for (int l = 0; l < V->mg->nlevels; l++) { free(V->c[l]); free(V->ex[l]); free(V->ey[l]); }
    if (V->general)
        for (int l = 0; l < V->mg->nlevels; l++) { free(V->a[l]); free(V->gx[l]); free(V->gy[l]); }
    orc_mg_free(V->mg);
    free(V);
}
orc_mg *orc_vcmg_base(orc_vcmg *V) { return V->mg; }
double *orc_vcmg_ptr(orc_vcmg *V, int level, int which)
{
    switch (which) {
    case 0: return V->c[level];
    case 1: return V->ex[level];
    case 2: return V->ey[level];
    case 3: return V->a[level];
    case 4: return V->gx[level];
    default: return V->gy[level];
    }
}

/* variable_coeff_MG.py:103-168 */
void orc_vcmg_smooth(orc_vcmg *V, int level, int nsmooth)
{
    orc_mg *m = V->mg;
    const int n = m->n[level], q = n + 2;
    double *v = m->v[level];
    const double *f = m->f[level], *ex = V->ex[level], *ey = V->ey[level];
    orc_mg_fill_bc_v(m, level);
    static const int grp[4][2] = {{0, 0}, {1, 1}, {1, 0}, {0, 1}};
#define I(i, j) ((size_t)(i) * q + (j))
    const double dxl = m->dx[level];
    for (int it = 0; it < nsmooth; it++)
        for (int g = 0; g < 4; g++) {
            for (int i = 1 + grp[g][0]; i <= n; i += 2)
                for (int j = 1 + grp[g][1]; j <= n; j += 2) {
                    if (V->general) {   /* general_MG.py:130-160 */
                        const double gxc = 0.5 * V->gx[level][I(i, j)] / dxl;
                        const double gyc = 0.5 * V->gy[level][I(i, j)] / dxl;
                        const double den = V->a[level][I(i, j)] - ex[I(i + 1, j)] - ex[I(i, j)] -
                                           ey[I(i, j + 1)] - ey[I(i, j)];
                        v[I(i, j)] = (f[I(i, j)] - (ex[I(i + 1, j)] + gxc) * v[I(i + 1, j)] -
                                      (ex[I(i, j)] - gxc) * v[I(i - 1, j)] -
                                      (ey[I(i, j + 1)] + gyc) * v[I(i, j + 1)] -
                                      (ey[I(i, j)] - gyc) * v[I(i, j - 1)]) / den;
                        continue;
                    }
                    double denom = ex[I(i + 1, j)] + ex[I(i, j)] + ey[I(i, j + 1)] + ey[I(i, j)];
                    v[I(i, j)] = (-f[I(i, j)] + ex[I(i + 1, j)] * v[I(i + 1, j)] +
                                  ex[I(i, j)] * v[I(i - 1, j)] + ey[I(i, j + 1)] * v[I(i, j + 1)] +
                                  ey[I(i, j)] * v[I(i, j - 1)]) / denom;
                }
            if (g == 1 || g == 3) orc_mg_fill_bc_v(m, level);
        }
#undef I
}

/* variable_coeff_MG.py:191-213 */
void orc_vcmg_residual(orc_vcmg *V, int level)
{
    orc_mg *m = V->mg;
    const int n = m->n[level], q = n + 2;
    const double *v = m->v[level], *f = m->f[level], *ex = V->ex[level], *ey = V->ey[level];
    double *r = m->r[level];
#define I(i, j) ((size_t)(i) * q + (j))
    const double dxl = m->dx[level];
    for (int i = 1; i <= n; i++)
        for (int j = 1; j <= n; j++) {
            if (V->general) {   /* general_MG.py:196-242 */
                const double gxc = 0.5 * V->gx[level][I(i, j)] / dxl;
                const double gyc = 0.5 * V->gy[level][I(i, j)] / dxl;
                const double Lg = V->a[level][I(i, j)] * v[I(i, j)] +
                                  ex[I(i + 1, j)] * (v[I(i + 1, j)] - v[I(i, j)]) -
                                  ex[I(i, j)] * (v[I(i, j)] - v[I(i - 1, j)]) +
                                  ey[I(i, j + 1)] * (v[I(i, j + 1)] - v[I(i, j)]) -
                                  ey[I(i, j)] * (v[I(i, j)] - v[I(i, j - 1)]) +
                                  gxc * (v[I(i + 1, j)] - v[I(i - 1, j)]) +
                                  gyc * (v[I(i, j + 1)] - v[I(i, j - 1)]);
                r[I(i, j)] = f[I(i, j)] - Lg;
                continue;
            }
            double L = ex[I(i + 1, j)] * (v[I(i + 1, j)] - v[I(i, j)]) -
                       ex[I(i, j)] * (v[I(i, j)] - v[I(i - 1, j)]) +
                       ey[I(i, j + 1)] * (v[I(i, j + 1)] - v[I(i, j)]) -
                       ey[I(i, j)] * (v[I(i, j)] - v[I(i, j - 1)]);
            r[I(i, j)] = f[I(i, j)] - L;
        }
#undef I
}

void orc_vcmg_vcycle(orc_vcmg *V, int level)
{
    orc_mg *m = V->mg;
    if (level > 0) {
        orc_vcmg_smooth(V, level, m->nsmooth);
        orc_vcmg_residual(V, level);
        orc_mg_restrict(m, level);
        orc_vcmg_vcycle(V, level - 1);
        orc_mg_prolong_add(m, level);
        orc_mg_fill_bc_v(m, level);
        orc_vcmg_smooth(V, level, m->nsmooth);
    } else {
        orc_vcmg_smooth(V, level, m->nsmooth_bottom);
        orc_mg_fill_bc_v(m, level);
    }
}

void orc_vcmg_solve(orc_vcmg *V, double rtol)
{
    orc_mg *m = V->mg;
    const int L = m->nlevels - 1;
    const int n = m->n[L], q = n + 2;
    const size_t N = (size_t)q * q;
    double *old_phi = (double *)malloc(N * 8);
    memcpy(old_phi, m->v[L], N * 8);
    double residual_error = 1.e33, relative_error = 1.e33;
    int cycle = 1;
    while (residual_error > rtol && cycle <= m->max_cycles) {
        for (int l = 0; l < L; l++)
            memset(m->v[l], 0, (size_t)(m->n[l] + 2) * (m->n[l] + 2) * 8);
        orc_vcmg_vcycle(V, L);
        double s = 0.0;
        for (int i = 1; i <= n; i++)
            for (int j = 1; j <= n; j++) {
                size_t k = (size_t)i * q + j;
                double d = (m->v[L][k] - old_phi[k]) / (m->v[L][k] + 1.e-16);
                s += d * d;
            }
        relative_error = sqrt(m->dx[L] * m->dx[L] * s);
        memcpy(old_phi, m->v[L], N * 8);
        orc_vcmg_residual(V, L);
        double rn = orc_mg_norm(m, L, 2);
        residual_error = (m->source_norm != 0.0) ? rn / m->source_norm : rn;
        cycle++;
    }
    m->num_cycles = cycle - 1;
    m->relative_error = relative_error;
    m->residual_error = residual_error;
    orc_mg_fill_bc_v(m, L);
    free(old_phi);
}

/* ================================================================== */
/* Burgers / incompressible: the callers of the  */
/* multigrid solver named in north_star.                                */
/*   pyro/burgers/burgers_interface.py:4-312                            */
/*   pyro/incompressible/incomp_interface.py:4-254                      */
/*   pyro/incompressible/simulation.py:77-330                           */
/* Planar arrays (qx,qy).  Region "B2" = valid region grown by 2; the   */
/* reference works on zero-initialised scratch arrays and never writes  */
/* outside the stated regions, so reads just outside them see 0.        */
/* ================================================================== */
enum { E_UXL, E_UXR, E_UYL, E_UYR, E_VXL, E_VXR, E_VYL, E_VYR };

/* burgers_interface.py:265-290 */
static inline double bg_riemann(double ql, double qr)
{
    if (ql <= 0.0 && qr >= 0.0) return 0.0;
    return (ql > 0.0 && ql + qr > 0.0) ? ql : qr;
}
/* burgers_interface.py:236-262 */
static inline double bg_upwind(double ql, double qr, double s)
{
    if (s == 0.0) return 0.5 * (ql + qr);
    return (s > 0.0) ? ql : qr;
}

/* get_interface_states (:4-86) + apply_transverse_corrections (:89-175) +
   apply_gradp_corrections (incomp_interface.py:139-183; gpx == NULL: burgers).
   E: 8 planes, zeroed here.  */
static void bg_edge_states_src(const double *u, const double *v, const double *gpx,
                               const double *gpy, const double *sx, const double *sy, int nx,
                               int ny, int ng, double dx, double dy, double dt, int limiter,
                               double *E);
void orc_bg_edge_states(const double *u, const double *v, const double *gpx,
                        const double *gpy, int nx, int ny, int ng, double dx,
                        double dy, double dt, int limiter, double *E)
{
    bg_edge_states_src(u, v, gpx, gpy, NULL, NULL, nx, ny, ng, dx, dy, dt, limiter, E);
}
/* burgers_viscous: + eps dt / 2 * Laplacian on the uncorrected states BEFORE the
   transverse terms (burgers_viscous/interface.py:94-171); 0: off */
static double g_bgv_eps = 0.0;
/* sx, sy: other source terms (apply_other_source_terms, incomp_interface.py:186-254) */
static void bg_edge_states_src(const double *u, const double *v, const double *gpx,
                               const double *gpy, const double *sx, const double *sy, int nx,
                               int ny, int ng, double dx, double dy, double dt, int limiter,
                               double *E)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
#define I2(i, j) ((size_t)(i) * qy + (j))
    double *ldux = zalloc(N), *ldvx = zalloc(N), *lduy = zalloc(N), *ldvy = zalloc(N);
    orc_limit(u, 1, nx, ny, ng, 1, limiter, ldux);
    orc_limit(v, 1, nx, ny, ng, 1, limiter, ldvx);
    orc_limit(u, 1, nx, ny, ng, 2, limiter, lduy);
    orc_limit(v, 1, nx, ny, ng, 2, limiter, ldvy);
    memset(E, 0, 8 * N * sizeof(double));
    double *uxl = E + E_UXL * N, *uxr = E + E_UXR * N, *uyl = E + E_UYL * N,
           *uyr = E + E_UYR * N, *vxl = E + E_VXL * N, *vxr = E + E_VXR * N,
           *vyl = E + E_VYL * N, *vyr = E + E_VYR * N;
    const double dtdx = dt / dx, dtdy = dt / dy;
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            const size_t k = I2(i, j);
            const double uc = u[k], vc = v[k];
            uxl[I2(i + 1, j)] = uc + 0.5 * (1.0 - dtdx * uc) * ldux[k];
            uxr[k] = uc - 0.5 * (1.0 + dtdx * uc) * ldux[k];
            vxl[I2(i + 1, j)] = vc + 0.5 * (1.0 - dtdx * uc) * ldvx[k];
            vxr[k] = vc - 0.5 * (1.0 + dtdx * uc) * ldvx[k];
            uyl[I2(i, j + 1)] = uc + 0.5 * (1.0 - dtdy * vc) * lduy[k];
            uyr[k] = uc - 0.5 * (1.0 + dtdy * vc) * lduy[k];
            vyl[I2(i, j + 1)] = vc + 0.5 * (1.0 - dtdy * vc) * ldvy[k];
            vyr[k] = vc - 0.5 * (1.0 + dtdy * vc) * ldvy[k];
        }
    if (g_bgv_eps != 0.0) {   /* apply_diffusion_corrections, get_lap (buf = 2) */
        const double eps = g_bgv_eps;
        for (int i = ilo - 2; i <= ihi + 2; i++)
            for (int j = jlo - 2; j <= jhi + 2; j++) {
                const size_t k = I2(i, j);
                const double lu = (u[I2(i + 1, j)] - 2.0 * u[k] + u[I2(i - 1, j)]) / (dx * dx) +
                                  (u[I2(i, j + 1)] - 2.0 * u[k] + u[I2(i, j - 1)]) / (dy * dy);
                const double lv = (v[I2(i + 1, j)] - 2.0 * v[k] + v[I2(i - 1, j)]) / (dx * dx) +
                                  (v[I2(i, j + 1)] - 2.0 * v[k] + v[I2(i, j - 1)]) / (dy * dy);
                const double cu = 0.5 * eps * dt * lu, cv = 0.5 * eps * dt * lv;
                uxl[I2(i + 1, j)] += cu;  uyl[I2(i, j + 1)] += cu;  uxr[k] += cu;  uyr[k] += cu;
                vxl[I2(i + 1, j)] += cv;  vyl[I2(i, j + 1)] += cv;  vxr[k] += cv;  vyr[k] += cv;
            }
    }
    /* transverse terms from the uncorrected states */
    double *uhat = zalloc(N), *vhat = zalloc(N), *uxi = zalloc(N), *vxi = zalloc(N),
           *uyi = zalloc(N), *vyi = zalloc(N);
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            const size_t k = I2(i, j);
            uhat[k] = bg_riemann(uxl[k], uxr[k]);
            vhat[k] = bg_riemann(vyl[k], vyr[k]);
            uxi[k] = bg_upwind(uxl[k], uxr[k], uhat[k]);
            vxi[k] = bg_upwind(vxl[k], vxr[k], uhat[k]);
            uyi[k] = bg_upwind(uyl[k], uyr[k], vhat[k]);
            vyi[k] = bg_upwind(vyl[k], vyr[k], vhat[k]);
        }
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            const size_t k = I2(i, j);
            const double ubar = 0.5 * (uhat[k] + uhat[I2(i + 1, j)]);
            const double vbar = 0.5 * (vhat[k] + vhat[I2(i, j + 1)]);
            const double tu = -0.5 * dtdy * vbar * (uyi[I2(i, j + 1)] - uyi[k]);
            const double tv = -0.5 * dtdy * vbar * (vyi[I2(i, j + 1)] - vyi[k]);
            uxl[I2(i + 1, j)] += tu;  uxr[k] += tu;
            vxl[I2(i + 1, j)] += tv;  vxr[k] += tv;
            const double sv = -0.5 * dtdx * ubar * (vxi[I2(i + 1, j)] - vxi[k]);
            const double su = -0.5 * dtdx * ubar * (uxi[I2(i + 1, j)] - uxi[k]);
            vyl[I2(i, j + 1)] += sv;  vyr[k] += sv;
            uyl[I2(i, j + 1)] += su;  uyr[k] += su;
            if (gpx) {
                const double gx = -0.5 * dt * gpx[k], gy = -0.5 * dt * gpy[k];
                uxl[I2(i + 1, j)] += gx;  uxr[k] += gx;
                vxl[I2(i + 1, j)] += gy;  vxr[k] += gy;
                vyl[I2(i, j + 1)] += gy;  vyr[k] += gy;
                uyl[I2(i, j + 1)] += gx;  uyr[k] += gx;
            }
            if (sx) {
                const double ax = 0.5 * dt * sx[k], ay = 0.5 * dt * sy[k];
                uxl[I2(i + 1, j)] += ax;  uxr[k] += ax;
                uyl[I2(i, j + 1)] += ax;  uyr[k] += ax;
                vxl[I2(i + 1, j)] += ay;  vxr[k] += ay;
                vyl[I2(i, j + 1)] += ay;  vyr[k] += ay;
            }
        }
    free(ldux); free(ldvx); free(lduy); free(ldvy);
    free(uhat); free(vhat); free(uxi); free(vxi); free(uyi); free(vyi);
#undef I2
}

/* burgers/simulation.py:37-51 (SMALL = 1e-12, simulation_null.py) */
double orc_bg_dt(const double *u, const double *v, int nx, int ny, int ng,
                 double dx, double dy, double cfl)
{
    const size_t N = (size_t)(nx + 2 * ng) * (ny + 2 * ng);
    double um = 0.0, vm = 0.0;
    for (size_t k = 0; k < N; k++) { um = dmax(um, fabs(u[k])); vm = dmax(vm, fabs(v[k])); }
    return cfl * dmin(dx / dmax(um, 1.e-12), dy / dmax(vm, 1.e-12));
}

/* burgers/simulation.py:53-117 + construct_unsplit_fluxes (:178-233) */
void orc_bg_step(double *u, double *v, int nx, int ny, int ng, double dx,
                 double dy, double dt, int limiter)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
#define I2(i, j) ((size_t)(i) * qy + (j))
    double *E = zalloc(8 * N);
    orc_bg_edge_states(u, v, NULL, NULL, nx, ny, ng, dx, dy, dt, limiter, E);
    double *fux = zalloc(N), *fvx = zalloc(N), *fuy = zalloc(N), *fvy = zalloc(N);
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            const size_t k = I2(i, j);
            const double um = bg_upwind(E[E_UXL * N + k], E[E_UXR * N + k],
                                        bg_riemann(E[E_UXL * N + k], E[E_UXR * N + k]));
            const double vm = bg_upwind(E[E_VYL * N + k], E[E_VYR * N + k],
                                        bg_riemann(E[E_VYL * N + k], E[E_VYR * N + k]));
            fux[k] = 0.5 * bg_upwind(E[E_UXL * N + k], E[E_UXR * N + k], um) * um;
            fvx[k] = 0.5 * bg_upwind(E[E_VXL * N + k], E[E_VXR * N + k], um) * um;
            fuy[k] = 0.5 * bg_upwind(E[E_UYL * N + k], E[E_UYR * N + k], vm) * vm;
            fvy[k] = 0.5 * bg_upwind(E[E_VYL * N + k], E[E_VYR * N + k], vm) * vm;
        }
    const double dtdx = dt / dx, dtdy = dt / dy;
    for (int i = ilo; i <= ihi; i++)
        for (int j = jlo; j <= jhi; j++) {
            const size_t k = I2(i, j);
            u[k] = u[k] + dtdx * (fux[k] - fux[I2(i + 1, j)]) + dtdy * (fuy[k] - fuy[I2(i, j + 1)]);
            v[k] = v[k] + dtdx * (fvx[k] - fvx[I2(i + 1, j)]) + dtdy * (fvy[k] - fvy[I2(i, j + 1)]);
        }
    free(E); free(fux); free(fvx); free(fuy); free(fvy);
#undef I2
}

/* burgers_viscous Simulation.evolve (pyro/burgers_viscous/simulation.py:9-89) with
   interface.diffuse (burgers_viscous/interface.py:27-91): advective terms from
   the unsplit fluxes, then one Crank-Nicolson Helmholtz solve per component.
   u, v: ghost cells filled; bc_u / bc_v: their boundary codes.  ncyc[2]: V-cycles */
void orc_bgv_step(double *u, double *v, int nx, int ng, double xmin, double xmax, double ymin,
                  double ymax, double dt, int limiter, double eps, const int *bc_u,
                  const int *bc_v, int *ncyc)
{
    const int ny = nx, qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    const double dx = (xmax - xmin) / nx, dy = (ymax - ymin) / ny;
#define I2(i, j) ((size_t)(i) * qy + (j))
    double *E = zalloc(8 * N);
    g_bgv_eps = eps;
    orc_bg_edge_states(u, v, NULL, NULL, nx, ny, ng, dx, dy, dt, limiter, E);
    g_bgv_eps = 0.0;
    /* construct_unsplit_fluxes, burgers_interface.py:178-233 */
    double *fux = zalloc(N), *fvx = zalloc(N), *fuy = zalloc(N), *fvy = zalloc(N);
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            const size_t k = I2(i, j);
            const double um = bg_upwind(E[E_UXL * N + k], E[E_UXR * N + k],
                                        bg_riemann(E[E_UXL * N + k], E[E_UXR * N + k]));
            const double vm = bg_upwind(E[E_VYL * N + k], E[E_VYR * N + k],
                                        bg_riemann(E[E_VYL * N + k], E[E_VYR * N + k]));
            fux[k] = 0.5 * bg_upwind(E[E_UXL * N + k], E[E_UXR * N + k], um) * um;
            fvx[k] = 0.5 * bg_upwind(E[E_VXL * N + k], E[E_VXR * N + k], um) * um;
            fuy[k] = 0.5 * bg_upwind(E[E_UYL * N + k], E[E_UYR * N + k], vm) * vm;
            fvy[k] = 0.5 * bg_upwind(E[E_VYL * N + k], E[E_VYR * N + k], vm) * vm;
        }
    double *Au = zalloc(N), *Av = zalloc(N);
    for (int i = ilo; i <= ihi; i++)
        for (int j = jlo; j <= jhi; j++) {
            const size_t k = I2(i, j);
            Au[k] = (fux[I2(i + 1, j)] - fux[k]) / dx + (fuy[I2(i, j + 1)] - fuy[k]) / dy;
            Av[k] = (fvx[I2(i + 1, j)] - fvx[k]) / dx + (fvy[I2(i, j + 1)] - fvy[k]) / dy;
        }
    for (int comp = 0; comp < 2; comp++) {   /* diffuse(), x-velocity first */
        double *a = comp ? v : u;
        const double *A = comp ? Av : Au;
        orc_mg *m = orc_mg_create(nx, xmin, xmax, ymin, ymax, comp ? bc_v : bc_u, 1.0,
                                  0.5 * dt * eps, 10, 50);
        const int L = m->nlevels - 1, n = nx;
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++) {
                const int gi = ilo + i, gj = jlo + j;
                const size_t k = I2(gi, gj);
                const double lap = (a[I2(gi + 1, gj)] - 2.0 * a[k] + a[I2(gi - 1, gj)]) / (dx * dx) +
                                   (a[I2(gi, gj + 1)] - 2.0 * a[k] + a[I2(gi, gj - 1)]) / (dy * dy);
                m->f[L][(size_t)(i + 1) * (n + 2) + j + 1] = a[k] + 0.5 * dt * eps * lap - dt * A[k];
            }
        orc_mg_init_rhs_norm(m);
        orc_mg_solve(m, 1.e-12);   /* init_zeros: the guess is 0 */
        if (ncyc) ncyc[comp] = m->num_cycles;
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++)
                a[I2(ilo + i, jlo + j)] = m->v[L][(size_t)(i + 1) * (n + 2) + j + 1];
        orc_mg_free(m);
    }
    free(E); free(fux); free(fvx); free(fuy); free(fvy); free(Au); free(Av);
#undef I2
}

/* incompressible_viscous (pyro/incompressible_viscous/simulation.py:8-190):
   viscosity nu >= 0 switches orc_incomp_step to the viscous update; lid_u /
   lid_v: ghost values of u / v on sides with code BC_CONST ("moving_lid",
   incompressible_viscous/BC.py:9-50) */
static struct { int on; double nu, lid_u, lid_v; } g_visc = {0, 0.0, 1.0, 0.0};
void orc_incomp_set_viscous(int on, double nu, double lid_u, double lid_v)
{
    g_visc.on = on; g_visc.nu = nu; g_visc.lid_u = lid_u; g_visc.lid_v = lid_v;
}
static void inc_fill(double *a, int nx, int ny, int ng, const int *bc, double cval)
{
    orc_fill_ghost(a, nx, ny, ng, 1, 0, bc);
    if (bc[3] == BC_CONST) {
        const int qx = nx + 2 * ng, qy = ny + 2 * ng;
        for (int i = 0; i < qx; i++)
            for (int j = ng + ny; j < qy; j++) a[(size_t)i * qy + j] = cval;
    }
}
/* Helmholtz solve of one velocity component, incompressible_viscous/
   simulation.py:78-123: (1 - dt nu/2 L) w = w + dt nu/2 L w - dt (advect [+ gradp]) */
static int visc_solve(double *w, const double *adv, const double *gp, int nx, int ng,
                      double xmin, double xmax, double ymin, double ymax, double dt, double nu,
                      int proj_type, const int *bc)
{
    const int ny = nx, qy = ny + 2 * ng, ilo = ng, jlo = ng;
    const double dx = (xmax - xmin) / nx, dy = (ymax - ymin) / ny;
#define I2(i, j) ((size_t)(i) * qy + (j))
    orc_mg *m = orc_mg_create(nx, xmin, xmax, ymin, ymax, bc, 1.0, 0.5 * dt * nu, 10, 50);
    const int L = m->nlevels - 1, n = nx;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++) {
            const int gi = ilo + i, gj = jlo + j;
            const size_t k = I2(gi, gj);
            double f = w[k] + 0.5 * dt * nu *
                                  ((w[I2(gi + 1, gj)] + w[I2(gi - 1, gj)] - 2.0 * w[k]) / (dx * dx) +
                                   (w[I2(gi, gj + 1)] + w[I2(gi, gj - 1)] - 2.0 * w[k]) / (dy * dy));
            if (proj_type == 1) f -= dt * (adv[k] + gp[k]);
            else f -= dt * adv[k];
            m->f[L][(size_t)(i + 1) * (n + 2) + j + 1] = f;
        }
    orc_mg_init_rhs_norm(m);
    for (int i = -1; i <= nx; i++)
        for (int j = -1; j <= ny; j++)
            m->v[L][(size_t)(i + 1) * (n + 2) + j + 1] = w[I2(ilo + i, jlo + j)];
    orc_mg_solve(m, 1.e-12);
    const int nc = m->num_cycles;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++)
            w[I2(ilo + i, jlo + j)] = m->v[L][(size_t)(i + 1) * (n + 2) + j + 1];
    orc_mg_free(m);
#undef I2
    return nc;
}

/* copy (qx,qy) interior [+buf] <-> MG level array (n+2)^2 */
static inline size_t mgk(int n, int i, int j) { return (size_t)i * (n + 2) + j; }

/* incompressible/simulation.py:200-330, one evolve().  D: 6 planes
   u, v, phi-MAC, phi, gradp_x, gradp_y (registration order :35-55); ghost
   cells of u, v filled on entry and on exit.  bc_u/bc_v/bc_phi: 4 codes each.
   Optional stage outputs (any may be NULL): o_umac/o_vmac (after the MAC
   projection), o_adv (2 planes).  Returns the number of V-cycles of the two
   solves in ncyc[2]. */
void orc_incomp_step(double *D, int nx, int ng, double xmin, double xmax,
                     double ymin, double ymax, double dt, int limiter,
                     int proj_type, const int *bc_u, const int *bc_v,
                     const int *bc_phi, int in_preevolve_unused,
                     double *o_umac, double *o_vmac, double *o_adv, int *ncyc)
{
    (void)in_preevolve_unused;
    const int ny = nx;
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    const double dx = (xmax - xmin) / nx, dy = (ymax - ymin) / ny;
    double *u = D, *v = D + N, *phiM = D + 2 * N, *phi = D + 3 * N, *gpx = D + 4 * N,
           *gpy = D + 5 * N;
#define I2(i, j) ((size_t)(i) * qy + (j))
    double *E = zalloc(8 * N);
    double *srcx = NULL, *srcy = NULL;
    if (g_visc.on) {   /* other_source_term, incompressible_viscous/simulation.py:24-41 */
        srcx = zalloc(N); srcy = zalloc(N);
        for (int i = ilo; i <= ihi; i++)
            for (int j = jlo; j <= jhi; j++) {
                const size_t k = I2(i, j);
                srcx[k] = g_visc.nu * ((u[I2(i + 1, j)] + u[I2(i - 1, j)] - 2.0 * u[k]) / (dx * dx) +
                                       (u[I2(i, j + 1)] + u[I2(i, j - 1)] - 2.0 * u[k]) / (dy * dy));
                srcy[k] = g_visc.nu * ((v[I2(i + 1, j)] + v[I2(i - 1, j)] - 2.0 * v[k]) / (dx * dx) +
                                       (v[I2(i, j + 1)] + v[I2(i, j - 1)] - 2.0 * v[k]) / (dy * dy));
            }
    }
    bg_edge_states_src(u, v, gpx, gpy, srcx, srcy, nx, ny, ng, dx, dy, dt, limiter, E);
    free(srcx); free(srcy);
    /* mac_vels: riemann_and_upwind on B2 (incomp_interface.py:62-63) */
    double *um = zalloc(N), *vm = zalloc(N);
    for (int i = ilo - 2; i <= ihi + 2; i++)
        for (int j = jlo - 2; j <= jhi + 2; j++) {
            const size_t k = I2(i, j);
            um[k] = bg_upwind(E[E_UXL * N + k], E[E_UXR * N + k],
                              bg_riemann(E[E_UXL * N + k], E[E_UXR * N + k]));
            vm[k] = bg_upwind(E[E_VYL * N + k], E[E_VYR * N + k],
                              bg_riemann(E[E_VYL * N + k], E[E_VYR * N + k]));
        }
    /* MAC projection (:232-262) */
    orc_mg *m = orc_mg_create(nx, xmin, xmax, ymin, ymax, bc_phi, 0.0, -1.0, 10, 50);
    const int L = m->nlevels - 1, n = nx;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++)
            m->f[L][mgk(n, i + 1, j + 1)] =
                (um[I2(ilo + i + 1, jlo + j)] - um[I2(ilo + i, jlo + j)]) / dx +
                (vm[I2(ilo + i, jlo + j + 1)] - vm[I2(ilo + i, jlo + j)]) / dy;
    orc_mg_init_rhs_norm(m);
    orc_mg_solve(m, 1.e-12);
    if (ncyc) ncyc[0] = m->num_cycles;
    for (int i = -1; i <= nx; i++)
        for (int j = -1; j <= ny; j++)
            phiM[I2(ilo + i, jlo + j)] = m->v[L][mgk(n, i + 1, j + 1)];
    for (int i = ilo; i <= ihi + 1; i++)
        for (int j = jlo; j <= jhi; j++)
            um[I2(i, j)] -= (phiM[I2(i, j)] - phiM[I2(i - 1, j)]) / dx;
    for (int i = ilo; i <= ihi; i++)
        for (int j = jlo; j <= jhi + 1; j++)
            vm[I2(i, j)] -= (phiM[I2(i, j)] - phiM[I2(i, j - 1)]) / dy;
    if (o_umac) memcpy(o_umac, um, N * 8);
    if (o_vmac) memcpy(o_vmac, vm, N * 8);
    /* states (:264-277) + provisional update (:286-304) */
    double *ax = zalloc(N), *ay = zalloc(N);
    {
        double *uxi = zalloc(N), *vxi = zalloc(N), *uyi = zalloc(N), *vyi = zalloc(N);
        for (int i = ilo - 2; i <= ihi + 2; i++)
            for (int j = jlo - 2; j <= jhi + 2; j++) {
                const size_t k = I2(i, j);
                uxi[k] = bg_upwind(E[E_UXL * N + k], E[E_UXR * N + k], um[k]);
                vxi[k] = bg_upwind(E[E_VXL * N + k], E[E_VXR * N + k], um[k]);
                uyi[k] = bg_upwind(E[E_UYL * N + k], E[E_UYR * N + k], vm[k]);
                vyi[k] = bg_upwind(E[E_VYL * N + k], E[E_VYR * N + k], vm[k]);
            }
        for (int i = ilo; i <= ihi; i++)
            for (int j = jlo; j <= jhi; j++) {
                const size_t k = I2(i, j), ki = I2(i + 1, j), kj = I2(i, j + 1);
                ax[k] = 0.5 * (um[k] + um[ki]) * (uxi[ki] - uxi[k]) / dx +
                        0.5 * (vm[k] + vm[kj]) * (uyi[kj] - uyi[k]) / dy;
                ay[k] = 0.5 * (um[k] + um[ki]) * (vxi[ki] - vxi[k]) / dx +
                        0.5 * (vm[k] + vm[kj]) * (vyi[kj] - vyi[k]) / dy;
            }
        free(uxi); free(vxi); free(uyi); free(vyi);
    }
    if (o_adv) { memcpy(o_adv, ax, N * 8); memcpy(o_adv + N, ay, N * 8); }
    if (g_visc.on) {   /* do_other_update_velocity: two parabolic solves */
        const int nu_c = visc_solve(u, ax, gpx, nx, ng, xmin, xmax, ymin, ymax, dt, g_visc.nu,
                                    proj_type, bc_u);
        const int nv_c = visc_solve(v, ay, gpy, nx, ng, xmin, xmax, ymin, ymax, dt, g_visc.nu,
                                    proj_type, bc_v);
        if (ncyc) { ncyc[2] = nu_c; ncyc[3] = nv_c; }
    } else
    for (size_t k = 0; k < N; k++) {
        if (proj_type == 1) {
            u[k] -= (dt * ax[k] + dt * gpx[k]);
            v[k] -= (dt * ay[k] + dt * gpy[k]);
        } else {
            u[k] -= dt * ax[k];
            v[k] -= dt * ay[k];
        }
    }
    inc_fill(u, nx, ny, ng, bc_u, g_visc.lid_u);
    inc_fill(v, nx, ny, ng, bc_v, g_visc.lid_v);
    /* final projection (:306-330) */
    for (int l = 0; l <= L; l++) {
        size_t Nl = (size_t)(m->n[l] + 2) * (m->n[l] + 2);
        memset(m->v[l], 0, Nl * 8); memset(m->f[l], 0, Nl * 8); memset(m->r[l], 0, Nl * 8);
    }
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++) {
            const int gi = ilo + i, gj = jlo + j;
            const double d = 0.5 * (u[I2(gi + 1, gj)] - u[I2(gi - 1, gj)]) / dx +
                             0.5 * (v[I2(gi, gj + 1)] - v[I2(gi, gj - 1)]) / dy;
            m->f[L][mgk(n, i + 1, j + 1)] = d / dt;
        }
    orc_mg_init_rhs_norm(m);
    for (int i = -1; i <= nx; i++)
        for (int j = -1; j <= ny; j++)
            m->v[L][mgk(n, i + 1, j + 1)] = phi[I2(ilo + i, jlo + j)];
    orc_mg_solve(m, 1.e-12);
    if (ncyc) ncyc[1] = m->num_cycles;
    memset(phi, 0, N * 8);
    for (int i = -1; i <= nx; i++)
        for (int j = -1; j <= ny; j++)
            phi[I2(ilo + i, jlo + j)] = m->v[L][mgk(n, i + 1, j + 1)];
    if (proj_type == 2) { memset(gpx, 0, N * 8); memset(gpy, 0, N * 8); }
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++) {
            const size_t k = I2(ilo + i, jlo + j);
            const double gx = 0.5 * (m->v[L][mgk(n, i + 2, j + 1)] - m->v[L][mgk(n, i, j + 1)]) / dx;
            const double gy = 0.5 * (m->v[L][mgk(n, i + 1, j + 2)] - m->v[L][mgk(n, i + 1, j)]) / dy;
            u[k] -= dt * gx;
            v[k] -= dt * gy;
            if (proj_type == 1) { gpx[k] += gx; gpy[k] += gy; }
            else { gpx[k] = gx; gpy[k] = gy; }
        }
    inc_fill(u, nx, ny, ng, bc_u, g_visc.lid_u);
    inc_fill(v, nx, ny, ng, bc_v, g_visc.lid_v);
    orc_mg_free(m);
    free(E); free(um); free(vm); free(ax); free(ay);
#undef I2
}

/* incompressible/simulation.py:77-143: initial projection of (u, v), then one
   throw-away evolve() whose only surviving product is gradp.  Returns the dt
   used for that evolve. */
double orc_incomp_preevolve(double *D, int nx, int ng, double xmin, double xmax,
                            double ymin, double ymax, double cfl, int limiter,
                            int proj_type, const int *bc_u, const int *bc_v,
                            const int *bc_phi)
{
    const int ny = nx;
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, jlo = ng;
    const size_t N = (size_t)qx * qy;
    const double dx = (xmax - xmin) / nx, dy = (ymax - ymin) / ny;
    double *u = D, *v = D + N, *phi = D + 3 * N;
#define I2(i, j) ((size_t)(i) * qy + (j))
    inc_fill(u, nx, ny, ng, bc_u, g_visc.lid_u);
    inc_fill(v, nx, ny, ng, bc_v, g_visc.lid_v);
    const int per[4] = {BC_PERIODIC, BC_PERIODIC, BC_PERIODIC, BC_PERIODIC}; /* :90-97 */
    orc_mg *m = orc_mg_create(nx, xmin, xmax, ymin, ymax, per, 0.0, -1.0, 10, 50);
    const int L = m->nlevels - 1, n = nx;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++) {
            const int gi = ilo + i, gj = jlo + j;
            m->f[L][mgk(n, i + 1, j + 1)] = 0.5 * (u[I2(gi + 1, gj)] - u[I2(gi - 1, gj)]) / dx +
                                            0.5 * (v[I2(gi, gj + 1)] - v[I2(gi, gj - 1)]) / dy;
        }
    orc_mg_init_rhs_norm(m);
    orc_mg_solve(m, 1.e-10);
    memset(phi, 0, N * 8);
    for (int i = -1; i <= nx; i++)
        for (int j = -1; j <= ny; j++)
            phi[I2(ilo + i, jlo + j)] = m->v[L][mgk(n, i + 1, j + 1)];
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++) {
            const size_t k = I2(ilo + i, jlo + j);
            u[k] -= 0.5 * (m->v[L][mgk(n, i + 2, j + 1)] - m->v[L][mgk(n, i, j + 1)]) / dx;
            v[k] -= 0.5 * (m->v[L][mgk(n, i + 1, j + 2)] - m->v[L][mgk(n, i + 1, j)]) / dy;
        }
    orc_mg_free(m);
    inc_fill(u, nx, ny, ng, bc_u, g_visc.lid_u);
    inc_fill(v, nx, ny, ng, bc_v, g_visc.lid_v);
    double *T = (double *)malloc(6 * N * 8);
    memcpy(T, D, 6 * N * 8);
    const double dt = orc_bg_dt(T, T + N, nx, ny, ng, dx, dy, cfl);
    orc_incomp_step(T, nx, ng, xmin, xmax, ymin, ymax, dt, limiter, proj_type, bc_u, bc_v,
                    bc_phi, 1, NULL, NULL, NULL, NULL);
    memcpy(D + 4 * N, T + 4 * N, 2 * N * 8);   /* gradp_x, gradp_y */
    free(T);
#undef I2
    return dt;
}

/* ================================================================== */
/* compressible_rk: method-of-lines right-hand side  */
/*   pyro/compressible_rk/fluxes.py:28-180 (no well-balancing)         */
/*   pyro/compressible_rk/simulation.py:10-44  substep                 */
/* U: (qx,qy,4) stage state with ghost cells filled; density floor is  */
/* applied in place like clean_state.  k: (qx,qy,4), interior written. */
/* Optional dumps Fx, Fy (with artificial viscosity).  Returns 1 when  */
/* the positivity assert of cons_to_prim would fire.                   */
/* ================================================================== */
int orc_comp_rk_rhs(double *U, const orc_comp_params *P, double *kout,
                    double *o_Fx, double *o_Fy)
{
    const int nx = P->nx, ny = P->ny, ng = P->ng;
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    const double gamma = P->gamma, dx = P->dx, dy = P->dy;
    int rc = 0;
#define U4(a, i, j, n) a[((size_t)(i) * qy + (j)) * 4 + (n)]
#define I2(i, j) ((size_t)(i) * qy + (j))
    for (int i = ilo; i <= ihi; i++)            /* clean_state */
        for (int j = jlo; j <= jhi; j++)
            U4(U, i, j, IDENS) = dmax(U4(U, i, j, IDENS), P->small_dens);
    double *S = zalloc(N * 4);                  /* simulation.py:16-19 */
    ext_sources_h(U, NULL, N, P->grav, 0.0, S, P->heat_rate, P->heat_prof);

    double *q = zalloc(N * 4), *xi = zalloc(N), *ldx = zalloc(N * 4), *ldy = zalloc(N * 4),
           *tmp = zalloc(N);
    double *V_l = zalloc(N * 4), *V_r = zalloc(N * 4);
    double *Uxl = zalloc(N * 4), *Uxr = zalloc(N * 4), *Uyl = zalloc(N * 4), *Uyr = zalloc(N * 4);
    double *Fx = zalloc(N * 4), *Fy = zalloc(N * 4), *avx = zalloc(N), *avy = zalloc(N);
    rc |= orc_cons_to_prim(U, nx, ny, ng, gamma, q);           /* fluxes.py:63-88 */
    if (P->use_flattening)
        orc_flatten_multid(q, nx, ny, ng, P->z0, P->z1, P->delta, xi);
    else
        for (size_t k = 0; k < N; k++) xi[k] = 1.0;
    for (int n = 0; n < 4; n++) {
        orc_limit(q + n, 4, nx, ny, ng, 1, P->limiter, tmp);
        for (size_t k = 0; k < N; k++) ldx[k * 4 + n] = xi[k] * tmp[k];
        orc_limit(q + n, 4, nx, ny, ng, 2, P->limiter, tmp);
        for (size_t k = 0; k < N; k++) ldy[k * 4 + n] = xi[k] * tmp[k];
    }
    /* piecewise-linear face states on B2, fluxes.py:107-140 */
    for (int d = 1; d <= 2; d++) {
        const double *ld = (d == 1) ? ldx : ldy;
        memset(V_l, 0, N * 32); memset(V_r, 0, N * 32);
        for (int i = ilo - 2; i <= ihi + 2; i++)
            for (int j = jlo - 2; j <= jhi + 2; j++)
                for (int n = 0; n < 4; n++) {
                    const double qc = U4(q, i, j, n), l = U4(ld, i, j, n);
                    if (d == 1) U4(V_l, i + 1, j, n) = qc + 0.5 * l;
                    else        U4(V_l, i, j + 1, n) = qc + 0.5 * l;
                    U4(V_r, i, j, n) = qc - 0.5 * l;
                }
        orc_prim_to_cons(V_l, N, gamma, d == 1 ? Uxl : Uyl);
        orc_prim_to_cons(V_r, N, gamma, d == 1 ? Uxr : Uyr);
    }
    riemann_dispatch(P, 1, Uxl, Uxr, Fx);                      /* fluxes.py:145-160 */
    riemann_dispatch(P, 2, Uyl, Uyr, Fy);
    orc_artificial_viscosity(nx, ny, ng, dx, dy, P->cvisc, q, P->avisc_xhi_interior,
                             P->avisc_yhi_interior, avx, avy);
    for (int n = 0; n < 4; n++)
        for (int i = ilo - 2; i <= ihi + 1; i++)
            for (int j = jlo - 2; j <= jhi + 1; j++) {
                U4(Fx, i, j, n) += avx[I2(i, j)] * (U4(U, i - 1, j, n) - U4(U, i, j, n));
                U4(Fy, i, j, n) += avy[I2(i, j)] * (U4(U, i, j - 1, n) - U4(U, i, j, n));
            }
    if (o_Fx) memcpy(o_Fx, Fx, N * 32);
    if (o_Fy) memcpy(o_Fy, Fy, N * 32);
    memset(kout, 0, N * 32);
    for (int i = ilo; i <= ihi; i++)            /* simulation.py:26-30 */
        for (int j = jlo; j <= jhi; j++)
            for (int n = 0; n < 4; n++)
                U4(kout, i, j, n) = (U4(Fx, i, j, n) - U4(Fx, i + 1, j, n)) / dx +
                                    (U4(Fy, i, j, n) - U4(Fy, i, j + 1, n)) / dy +
                                    U4(S, i, j, n);
    if (P->do_sponge) {                         /* simulation.py:33-42 */
        const double PI = 3.14159265358979323846;
        for (int i = ilo; i <= ihi; i++)
            for (int j = jlo; j <= jhi; j++) {
                const double rho = U4(U, i, j, IDENS);
                double f;
                if (rho > P->sponge_rho_begin) f = 0.0;
                else if (rho < P->sponge_rho_full) f = 1.0;
                else f = 0.5 * (1.0 - cos(PI * (rho - P->sponge_rho_begin) /
                                          (P->sponge_rho_full - P->sponge_rho_begin)));
                const double kap = f / P->sponge_timescale;
                const double mx = U4(U, i, j, IXMOM), my = U4(U, i, j, IYMOM);
                U4(kout, i, j, IXMOM) -= kap * mx;
                U4(kout, i, j, IYMOM) -= kap * my;
                U4(kout, i, j, IENER) -= kap * (mx * mx / rho + my * my / rho);
            }
    }
    free(S); free(q); free(xi); free(ldx); free(ldy); free(tmp); free(V_l); free(V_r);
    free(Uxl); free(Uxr); free(Uyl); free(Uyr); free(Fx); free(Fy); free(avx); free(avy);
#undef U4
#undef I2
    return rc;
}

/* compressible_rk/simulation.py:46-56 (the dt is NOT the CTU one) */
double orc_comp_rk_dt(const double *U, int nx, int ny, int ng, double dx,
                      double dy, double gamma, double cfl)
{
    const size_t N = (size_t)(nx + 2 * ng) * (ny + 2 * ng);
    double m = INFINITY;
    for (size_t k = 0; k < N; k++) {
        const double *Uc = U + k * 4;
        const double rho = Uc[IDENS], u = Uc[IXMOM] / rho, v = Uc[IYMOM] / rho;
        const double e = (Uc[IENER] - 0.5 * rho * (u * u + v * v)) / rho;
        const double p = rho * e * (gamma - 1.0);
        const double cs = sqrt(gamma * p / rho);
        const double xtmp = (fabs(u) + cs) / dx, ytmp = (fabs(v) + cs) / dy;
        m = dmin(m, 1.0 / (xtmp + ytmp));
    }
    return cfl * m;
}

/* ================================================================== */
/* Shallow water: pyro/swe                           */
/*   simulation.py:48-80 (cons/prim), :143-193 (dt, evolve)            */
/*   unsplit_fluxes.py:132-380                                         */
/*   interface.py:5-578 (states, riemann_roe, riemann_hllc, consFlux)  */
/* 4 variables: height, x-momentum, y-momentum, fuel (h X); primitive  */
/* h, u, v, X.  (qx,qy,4) AoS arrays.                                  */
/* ================================================================== */
#define SQ(x) sq_ref(x)
enum { SH = 0, SMX = 1, SMY = 2, SHX = 3 };

typedef struct {
    int nx, ny, ng;
    double dx, dy, g;
    int limiter;
    int riemann;     /* 0 Roe, 1 HLLC */
} orc_swe_params;

/* interface.py:557-578 */
static void swe_cons_flux(int idir, double g, const double *U, double *F)
{
    const double u = U[SMX] / U[SH], v = U[SMY] / U[SH];
    if (idir == 1) {
        F[SH] = U[SH] * u;
        F[SMX] = U[SMX] * u + 0.5 * g * SQ(U[SH]);
        F[SMY] = U[SMY] * u;
        F[SHX] = U[SHX] * u;
    } else {
        F[SH] = U[SH] * v;
        F[SMX] = U[SMX] * v;
        F[SMY] = U[SMY] * v + 0.5 * g * SQ(U[SH]);
        F[SHX] = U[SHX] * v;
    }
}

/* interface.py:5-213 */
static void swe_states(int idir, int nx, int ny, int ng, double dx, double dt, double g,
                       const double *qv, const double *dqv, double *q_l, double *q_r)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny; /* njit-local */
    memset(q_l, 0, sizeof(double) * qx * qy * 4);
    memset(q_r, 0, sizeof(double) * qx * qy * 4);
    const double dtdx = dt / dx;
    const double dtdx3 = 0.33333 * dtdx;   /* sic, interface.py:100 */
    const int in = (idir == 1) ? 1 : 2;    /* normal velocity */
    double lvec[4][4], rvec[4][4], e_val[4], betal[4], betar[4];
    for (int i = ilo - 2; i < ihi + 2; i++)
        for (int j = jlo - 2; j < jhi + 2; j++) {
            const double *dq = dqv + ((size_t)i * qy + j) * 4;
            const double *q = qv + ((size_t)i * qy + j) * 4;
            const double cs = sqrt(g * q[0]);
            memset(lvec, 0, sizeof lvec);
            memset(rvec, 0, sizeof rvec);
            e_val[0] = q[in] - cs; e_val[1] = q[in]; e_val[2] = q[in] + cs; e_val[3] = q[in];
            if (idir == 1) {
                lvec[0][0] = cs;   lvec[0][1] = -q[0];
                lvec[1][2] = 1.0;
                lvec[2][0] = cs;   lvec[2][1] = q[0];
                rvec[0][0] = q[0]; rvec[0][1] = -cs;
                rvec[1][2] = 1.0;
                rvec[2][0] = q[0]; rvec[2][1] = cs;
            } else {
                lvec[0][0] = cs;   lvec[0][2] = -q[0];
                lvec[1][1] = 1.0;
                lvec[2][0] = cs;   lvec[2][2] = q[0];
                rvec[0][0] = q[0]; rvec[0][2] = -cs;
                rvec[1][1] = 1.0;
                rvec[2][0] = q[0]; rvec[2][2] = cs;
            }
            lvec[3][3] = 1.0; rvec[3][3] = 1.0;
            for (int k = 0; k < 4; k++) {
                lvec[0][k] = lvec[0][k] * 0.50 / (cs * q[0]);
                lvec[2][k] = -lvec[2][k] * 0.50 / (cs * q[0]);
            }
            double *ql = (idir == 1) ? q_l + ((size_t)(i + 1) * qy + j) * 4
                                     : q_l + ((size_t)i * qy + (j + 1)) * 4;
            double *qr = q_r + ((size_t)i * qy + j) * 4;
            double factor = 0.5 * (1.0 - dtdx * dmax(e_val[2], 0.0));
            for (int m = 0; m < 4; m++) ql[m] = q[m] + factor * dq[m];
            factor = 0.5 * (1.0 + dtdx * dmin(e_val[0], 0.0));
            for (int m = 0; m < 4; m++) qr[m] = q[m] - factor * dq[m];
            for (int m = 0; m < 4; m++) {
                double asum = 0.0;
                for (int k = 0; k < 4; k++) asum += lvec[m][k] * dq[k];
                betal[m] = dtdx3 * (e_val[2] - e_val[m]) * (copysign(1.0, e_val[m]) + 1.0) * asum;
                betar[m] = dtdx3 * (e_val[0] - e_val[m]) * (1.0 - copysign(1.0, e_val[m])) * asum;
            }
            for (int m = 0; m < 4; m++) {
                double sum_l = 0.0, sum_r = 0.0;
                for (int k = 0; k < 4; k++) {
                    sum_l += betal[k] * rvec[k][m];
                    sum_r += betar[k] * rvec[k][m];
                }
                ql[m] = ql[m] + sum_l;
                qr[m] = qr[m] + sum_r;
            }
        }
}

/* interface.py:216-385 */
static void swe_riemann_roe(int idir, int nx, int ny, int ng, double g, const double *U_l,
                            const double *U_r, double *F)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny;
    const double smallc = 1.e-10, tol = 0.1e-1;
    const int im = (idir == 1) ? SMX : SMY;
    memset(F, 0, sizeof(double) * qx * qy * 4);
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            const double *Ul = U_l + ((size_t)i * qy + j) * 4, *Ur = U_r + ((size_t)i * qy + j) * 4;
            double *Fo = F + ((size_t)i * qy + j) * 4;
            const double h_l = Ul[SH], un_l = Ul[im] / h_l;
            const double h_r = Ur[SH], un_r = Ur[im] / h_r;
            const double c_l = dmax(smallc, sqrt(g * h_l)), c_r = dmax(smallc, sqrt(g * h_r));
            double U_roe[4], delta[4], lambda[4], alpha[4], K[4][4];
            for (int n = 0; n < 4; n++) {
                U_roe[n] = (Ul[n] / sqrt(h_l) + Ur[n] / sqrt(h_r)) / (sqrt(h_l) + sqrt(h_r));
                delta[n] = Ur[n] / h_r - Ul[n] / h_l;
            }
            U_roe[SH] = sqrt(h_l * h_r);
            const double c_roe = sqrt(0.5 * (SQ(c_l) + SQ(c_r)));
            delta[SH] = h_r - h_l;
            const double un_roe = U_roe[im];
            memset(K, 0, sizeof K);
            lambda[0] = un_roe - c_roe; lambda[1] = un_roe; lambda[2] = un_roe + c_roe;
            if (idir == 1) {
                alpha[0] = 0.5 * (delta[SH] - U_roe[SH] / c_roe * delta[SMX]);
                alpha[1] = U_roe[SH] * delta[SMY];
                alpha[2] = 0.5 * (delta[SH] + U_roe[SH] / c_roe * delta[SMX]);
                K[0][0] = 1.0; K[0][1] = un_roe - c_roe; K[0][2] = U_roe[SMY];
                K[1][2] = 1.0;
                K[2][0] = 1.0; K[2][1] = un_roe + c_roe; K[2][2] = U_roe[SMY];
            } else {
                alpha[0] = 0.5 * (delta[SH] - U_roe[SH] / c_roe * delta[SMY]);
                alpha[1] = U_roe[SH] * delta[SMX];
                alpha[2] = 0.5 * (delta[SH] + U_roe[SH] / c_roe * delta[SMY]);
                K[0][0] = 1.0; K[0][1] = U_roe[SMX]; K[0][2] = un_roe - c_roe;
                K[1][1] = 1.0;
                K[2][0] = 1.0; K[2][1] = U_roe[SMX]; K[2][2] = un_roe + c_roe;
            }
            lambda[3] = un_roe;
            alpha[3] = U_roe[SH] * delta[3];
            K[3][3] = 1.0;
            double Fl[4], Fr[4];
            swe_cons_flux(idir, g, Ul, Fl);
            swe_cons_flux(idir, g, Ur, Fr);
            for (int n = 0; n < 4; n++) Fo[n] = 0.5 * (Fl[n] + Fr[n]);
            const double h_star = 1.0 / g * SQ(0.5 * (c_l + c_r) + 0.25 * (un_l - un_r));
            const double u_star = 0.5 * (un_l + un_r) + c_l - c_r;
            const double c_star = sqrt(g * h_star);
            if (fabs(lambda[0]) < tol)
                lambda[0] = lambda[0] * (u_star - c_star - lambda[0]) /
                            (u_star - c_star - (un_l - c_l));
            if (fabs(lambda[2]) < tol)
                lambda[2] = lambda[2] * (u_star + c_star - lambda[2]) /
                            (u_star + c_star - (un_r + c_r));
            for (int n = 0; n < 4; n++)
                for (int m = 0; m < 4; m++)
                    Fo[n] -= 0.5 * alpha[m] * fabs(lambda[m]) * K[m][n];
        }
}

/* interface.py:388-554 */
static void swe_riemann_hllc(int idir, int nx, int ny, int ng, double g, const double *U_l,
                             const double *U_r, double *F)
{
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx, jlo = ng, jhi = ng + ny;
    const double smallc = 1.e-10;
    const int im = (idir == 1) ? SMX : SMY, it = (idir == 1) ? SMY : SMX;
    memset(F, 0, sizeof(double) * qx * qy * 4);
    for (int i = ilo - 1; i < ihi + 1; i++)
        for (int j = jlo - 1; j < jhi + 1; j++) {
            const double *Ul = U_l + ((size_t)i * qy + j) * 4, *Ur = U_r + ((size_t)i * qy + j) * 4;
            double *Fo = F + ((size_t)i * qy + j) * 4;
            const double h_l = Ul[SH], un_l = Ul[im] / h_l, ut_l = Ul[it] / h_l;
            const double h_r = Ur[SH], un_r = Ur[im] / h_r, ut_r = Ur[it] / h_r;
            const double c_l = dmax(smallc, sqrt(g * h_l)), c_r = dmax(smallc, sqrt(g * h_r));
            const double h_avg = 0.5 * (h_l + h_r), c_avg = 0.5 * (c_l + c_r);
            const double hstar = h_avg - 0.25 * (un_r - un_l) * h_avg / c_avg;
            const double S_l = (hstar <= h_l) ? un_l - c_l
                                              : un_l - c_l * sqrt(0.5 * (hstar + h_l) * hstar) / h_l;
            const double S_r = (hstar <= h_r) ? un_r + c_r
                                              : un_r + c_r * sqrt(0.5 * (hstar + h_r) * hstar) / h_r;
            const double S_c = (S_l * h_r * (un_r - S_r) - S_r * h_l * (un_l - S_l)) /
                               (h_r * (un_r - S_r) - h_l * (un_l - S_l));
            double Us[4];
            if (S_r <= 0.0) {
                swe_cons_flux(idir, g, Ur, Fo);
            } else if (S_c <= 0.0 && 0.0 < S_r) {
                const double fac = h_r * (S_r - un_r) / (S_r - S_c);
                Us[SH] = fac; Us[im] = fac * S_c; Us[it] = fac * ut_r;
                Us[SHX] = fac * Ur[SHX] / h_r;
                swe_cons_flux(idir, g, Ur, Fo);
                for (int n = 0; n < 4; n++) Fo[n] = Fo[n] + S_r * (Us[n] - Ur[n]);
            } else if (S_l < 0.0 && 0.0 < S_c) {
                const double fac = h_l * (S_l - un_l) / (S_l - S_c);
                Us[SH] = fac; Us[im] = fac * S_c; Us[it] = fac * ut_l;
                Us[SHX] = fac * Ul[SHX] / h_l;
                swe_cons_flux(idir, g, Ul, Fo);
                for (int n = 0; n < 4; n++) Fo[n] = Fo[n] + S_l * (Us[n] - Ul[n]);
            } else {
                swe_cons_flux(idir, g, Ul, Fo);
            }
        }
}

static void swe_riemann(const orc_swe_params *P, int idir, const double *Ul, const double *Ur,
                        double *F)
{
    if (P->riemann == 1) swe_riemann_hllc(idir, P->nx, P->ny, P->ng, P->g, Ul, Ur, F);
    else swe_riemann_roe(idir, P->nx, P->ny, P->ng, P->g, Ul, Ur, F);
}

/* simulation.py:48-80 */
static void swe_prim_to_cons(const double *q, size_t N, double *U)
{
    for (size_t k = 0; k < N; k++) {
        const double *qc = q + k * 4;
        double *Uc = U + k * 4;
        Uc[SH] = qc[0];
        Uc[SMX] = qc[1] * Uc[SH];
        Uc[SMY] = qc[2] * Uc[SH];
        Uc[SHX] = qc[3] * qc[0];
    }
}

/* simulation.py:143-153 + derives.py */
double orc_swe_dt(const double *U, int nx, int ny, int ng, double dx, double dy, double g,
                  double cfl)
{
    const size_t N = (size_t)(nx + 2 * ng) * (ny + 2 * ng);
    double xm = INFINITY, ym = INFINITY;
    for (size_t k = 0; k < N; k++) {
        const double h = U[k * 4 + SH], u = U[k * 4 + SMX] / h, v = U[k * 4 + SMY] / h;
        const double cs = sqrt(g * h);
        xm = dmin(xm, dx / (fabs(u) + cs));
        ym = dmin(ym, dy / (fabs(v) + cs));
    }
    return cfl * dmin(xm, ym);
}

typedef struct {
    double *Uxl0, *Uxr0, *Uyl0, *Uyr0, *FxT, *FyT, *Fx, *Fy;   /* any may be NULL */
} orc_swe_stages;

/* Simulation.evolve (simulation.py:155-193) with unsplit_fluxes.py:132-380 */
void orc_swe_step(double *U, const orc_swe_params *P, double dt, orc_swe_stages *st)
{
    const int nx = P->nx, ny = P->ny, ng = P->ng;
    const int qx = nx + 2 * ng, qy = ny + 2 * ng;
    const int ilo = ng, ihi = ng + nx - 1, jlo = ng, jhi = ng + ny - 1;
    const size_t N = (size_t)qx * qy;
    const double dx = P->dx, dy = P->dy;
#define U4(a, i, j, n) a[((size_t)(i) * qy + (j)) * 4 + (n)]
    double *q = zalloc(N * 4), *ldx = zalloc(N * 4), *ldy = zalloc(N * 4), *tmp = zalloc(N);
    double *V_l = zalloc(N * 4), *V_r = zalloc(N * 4);
    double *Uxl = zalloc(N * 4), *Uxr = zalloc(N * 4), *Uyl = zalloc(N * 4), *Uyr = zalloc(N * 4);
    double *Fx = zalloc(N * 4), *Fy = zalloc(N * 4);
    for (size_t k = 0; k < N; k++) {            /* cons_to_prim over the whole array */
        q[k * 4 + 0] = U[k * 4 + SH];
        q[k * 4 + 1] = U[k * 4 + SMX] / U[k * 4 + SH];
        q[k * 4 + 2] = U[k * 4 + SMY] / U[k * 4 + SH];
        q[k * 4 + 3] = U[k * 4 + SHX] / q[k * 4 + 0];
    }
    for (int n = 0; n < 4; n++) {               /* xi = 1 (no flattening) */
        orc_limit(q + n, 4, nx, ny, ng, 1, P->limiter, tmp);
        for (size_t k = 0; k < N; k++) ldx[k * 4 + n] = 1.0 * tmp[k];
        orc_limit(q + n, 4, nx, ny, ng, 2, P->limiter, tmp);
        for (size_t k = 0; k < N; k++) ldy[k * 4 + n] = 1.0 * tmp[k];
    }
    swe_states(1, nx, ny, ng, dx, dt, P->g, q, ldx, V_l, V_r);
    swe_prim_to_cons(V_l, N, Uxl); swe_prim_to_cons(V_r, N, Uxr);
    swe_states(2, nx, ny, ng, dy, dt, P->g, q, ldy, V_l, V_r);
    swe_prim_to_cons(V_l, N, Uyl); swe_prim_to_cons(V_r, N, Uyr);
    if (st && st->Uxl0) memcpy(st->Uxl0, Uxl, N * 32);
    if (st && st->Uxr0) memcpy(st->Uxr0, Uxr, N * 32);
    if (st && st->Uyl0) memcpy(st->Uyl0, Uyl, N * 32);
    if (st && st->Uyr0) memcpy(st->Uyr0, Uyr, N * 32);
    swe_riemann(P, 1, Uxl, Uxr, Fx);
    swe_riemann(P, 2, Uyl, Uyr, Fy);
    if (st && st->FxT) memcpy(st->FxT, Fx, N * 32);
    if (st && st->FyT) memcpy(st->FyT, Fy, N * 32);
    {   /* transverse flux differences, unsplit_fluxes.py:331-352, b = (2, 1) */
        const double dtdx = dt / dx, dtdy = dt / dy;
        for (int n = 0; n < 4; n++)
            for (int i = ilo - 2; i <= ihi + 1; i++)
                for (int j = jlo - 2; j <= jhi + 1; j++) {
                    U4(Uxl, i, j, n) += -0.5 * dtdy * (U4(Fy, i - 1, j + 1, n) - U4(Fy, i - 1, j, n));
                    U4(Uxr, i, j, n) += -0.5 * dtdy * (U4(Fy, i, j + 1, n) - U4(Fy, i, j, n));
                    U4(Uyl, i, j, n) += -0.5 * dtdx * (U4(Fx, i + 1, j - 1, n) - U4(Fx, i, j - 1, n));
                    U4(Uyr, i, j, n) += -0.5 * dtdx * (U4(Fx, i + 1, j, n) - U4(Fx, i, j, n));
                }
    }
    swe_riemann(P, 1, Uxl, Uxr, Fx);
    swe_riemann(P, 2, Uyl, Uyr, Fy);
    if (st && st->Fx) memcpy(st->Fx, Fx, N * 32);
    if (st && st->Fy) memcpy(st->Fy, Fy, N * 32);
    {
        const double dtdx = dt / dx, dtdy = dt / dy;
        for (int n = 0; n < 4; n++)
            for (int i = ilo; i <= ihi; i++)
                for (int j = jlo; j <= jhi; j++)
                    U4(U, i, j, n) += dtdx * (U4(Fx, i, j, n) - U4(Fx, i + 1, j, n)) +
                                      dtdy * (U4(Fy, i, j, n) - U4(Fy, i, j + 1, n));
    }
#undef U4
    free(q); free(ldx); free(ldy); free(tmp); free(V_l); free(V_r);
    free(Uxl); free(Uxr); free(Uyl); free(Uyr); free(Fx); free(Fy);
}
#undef SQ
