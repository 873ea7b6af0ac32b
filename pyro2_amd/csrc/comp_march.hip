// Compressible CTU + Riemann step as ONE kernel per time step, marching along
// the rows (kernel_set 2).
//
// Same arithmetic as the staged kernels (compressible.hip) and the 2-d tile
// kernel (comp_fused.hip) -- the per-cell functions of hydro.h / stencil.h in
// the reference's operation order -- but organised so that NOTHING is computed
// twice: the tile kernel spends 18 % of its threads on apron cells, evaluates
// every limit2 three times and every 1-d flattening coefficient twice
// (1661 VALU lane-instructions per cell update).
//
// Decomposition.  A workgroup of MNT = 256 threads owns MNT - 8 = 248 columns
// (thread = column j, lanes along the contiguous axis -> 512-B row loads per
// wave and plane) and walks down a strip of L rows.  The x direction (index i,
// rows) lives in REGISTERS as rolling windows: limit2_x, the 1-d flattening
// coefficient in x, the x face states and the x fluxes of a row are each
// computed once and handed from one iteration to the next.  The y direction
// (neighbouring lanes, possibly in another wave) goes through LDS rings that
// hold ONE row each.  The row loop is software-pipelined: iteration k runs
//
//   half A   S0  load row k -> primitives (register window rows k-4..k);
//                flatten_x(k-2), limit2_x(k-2); publish Q(k-2)           -> e1
//            S2  row k-3: xi, limited slopes, characteristic tracing ->
//                XM XP YM YP; transverse x flux FxT(k-3); publish YP     -> e3
//                row k-4: transverse correction of the y states; publish -> e5
//            S4  row k-4: transverse correction of the x states (FyT from e4),
//                final x flux Fx(k-4) + artificial viscosity; publish U  -> e7
//            S6  row k-5: conservative update (Fy from e6), store, CFL
//   barrier
//   half B   S1  row k-2: limit2_y, flatten_y (from e1) -> e2; vertex div -> dv
//            S3  row k-3: transverse y flux FyT (YP from e3)             -> e4
//            S5  row k-4: final y flux Fy (YP from e5) + art. viscosity  -> e6
//   barrier
//
// Every ring is written in one half and read in the other, so one slot per
// ring is enough (e1 has two: S2 re-reads the row S1 used); 34 row arrays of
// 260 doubles = 69 KiB, two workgroups (8 waves) per CU.  A strip costs L + 9
// iterations for L rows and 256 lanes for 248 columns: 93-95 % of the lanes do
// work that is needed, against 82 % (phase 0: 61 %) in the tile kernel.  HBM
// traffic: every row is read once per column block (+8 apron columns, +8 apron
// rows per strip) and written once; the second read of a row four iterations
// later (old state for the update and the viscosity terms) is an L2 hit.
//
// Compiled twice like the other compressible units (PYRO_FAST = 0 / 1).
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

#include "fused_common.h"

constexpr int MNT = 256;          // threads = columns per workgroup
constexpr int MOUT = MNT - 8;     // columns a workgroup updates
constexpr int MW = MNT + 4;       // ring width: 2 pad columns on either side
constexpr int MROWS = 8 + 5 + 4 + 4 + 4 + 4 + 4 + 1;
constexpr size_t MLDS_BYTES = (size_t)MROWS * MW * sizeof(double);

__device__ __forceinline__ Cons ring_get(const double *b, int x)
{
    return Cons{b[x], b[MW + x], b[2 * MW + x], b[3 * MW + x]};
}
__device__ __forceinline__ void ring_put(double *b, int x, const Cons &U)
{
    b[x] = U.d; b[MW + x] = U.E; b[2 * MW + x] = U.mx; b[3 * MW + x] = U.my;
}

template <int SOLVER, bool STD>   // as k_ctu_fused
__global__ __launch_bounds__(MNT, 2) void k_ctu_march(const double *__restrict__ Uin,
                                                      double *__restrict__ Uout, Geom g, FP P,
                                                      int *__restrict__ flag,
                                                      double *__restrict__ partial)
{
    HIP_DYNAMIC_SHARED(double, lds)
    double *e1 = lds;              // Q = (rho,u,v,p) of one row, 2 slots
    double *e2 = e1 + 8 * MW;      // limit2_y of (rho,u,v,p), flatten_y
    double *e3 = e2 + 5 * MW;      // YP (upper y face state), uncorrected
    double *e4 = e3 + 4 * MW;      // transverse y flux FyT
    double *e5 = e4 + 4 * MW;      // YP, corrected
    double *e6 = e5 + 4 * MW;      // final y flux Fy
    double *e7 = e6 + 4 * MW;      // old state U (viscosity term of the y faces)
    double *dv = e7 + 4 * MW;      // vertex div(U)

    const int tj = threadIdx.x, x = tj + 2;
    const int cb = blockIdx.x % P.ncb, sb = blockIdx.x / P.ncb;
    const int i0 = g.ilo + sb * P.L;                       // strip rows [i0, i1)
    const int i1 = (i0 + P.L < g.ihi + 1) ? i0 + P.L : g.ihi + 1;
    const int j = g.jlo + cb * MOUT - 4 + tj;              // this thread's column
    const int jc = (j < g.qy) ? j : g.qy - 1;              // ragged last block: clamp, unused
    const bool jin = (j >= g.jlo && j <= g.jhi);
    const bool jout = jin && tj >= 4 && tj <= MNT - 5;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const double gamma = P.gamma;
    const int limiter = STD ? 2 : P.limiter;
    const bool flat = STD || P.use_flattening;
    const double hdtV = P.hdtV, Ax = P.dy, Ay = P.dx;

    auto loadU = [&](int row) {
        row = row < 0 ? 0 : (row > g.qx - 1 ? g.qx - 1 : row);
        const size_t kk = (size_t)row * p + jc;
        return Cons{Uin[kk], Uin[pl + kk], Uin[2 * pl + kk], Uin[3 * pl + kk]};
    };
    auto row_in = [&](int r) { return r >= g.ilo && r <= g.ihi; };

    // state carried from one iteration to the next (rows relative to the
    // iteration k that is about to start)
    double wr[5] = {1, 1, 1, 1, 1}, wu[5] = {0, 0, 0, 0, 0};   // Q window, rows k-4..k
    double wv[5] = {0, 0, 0, 0, 0}, wp[5] = {1, 1, 1, 1, 1};
    double l2a[4] = {0, 0, 0, 0}, l2b[4] = {0, 0, 0, 0};       // limit2_x of rows k-4, k-3
    double fxa = 1.0, fxb = 1.0;                               // flatten_x of rows k-4, k-3
    const Cons one{1.0, 1.0, 0.0, 0.0}, zero{0.0, 0.0, 0.0, 0.0};
    Cons XMs = one, XPs = one, YMs = one, YPs = one;           // uncorrected states, row k-4
    Cons FxTp = zero;                                          // FxT(k-4)
    Cons XPcp = one;                                           // corrected XP(k-5)
    Cons Fxp = zero;                                           // Fx(k-5)
    Cons Ue = one, Uem = one;                                  // old state, rows k-4 / k-5
    Cons YMc = one;                                            // corrected YM(k-4), A -> B
    double avx_c = 0.0, avy_c = 0.0, Dp = 0.0;                 // avx(k-4), avy(k-4), D(k-3)
    double up = 0.0, vp = 0.0;                                 // u, v at (k-3, j-1)
    Cons Upre = loadU(i0 - 4);                                 // row k, in flight
    Cons Urep = loadU(i0 - 8);                                 // row k-4 again, in flight
    bool bad = false;
    double cfl = INFINITY;

    for (int k = i0 - 4; k <= i1 + 4; k++) {
        // ================= half A =================
#pragma unroll
        for (int n = 0; n < 4; n++) {
            wr[n] = wr[n + 1]; wu[n] = wu[n + 1]; wv[n] = wv[n + 1]; wp[n] = wp[n + 1];
        }
        Uem = Ue;
        Ue = Urep;
        if (row_in(k - 4) && jin) Ue.d = fmax(Ue.d, P.small_dens);      // clean_state
        Urep = loadU(k - 3);
        // ---- S0: row k -> primitives
        {
            Cons U = Upre;
            Upre = loadU(k + 1);
            const bool interior = row_in(k) && jin;
            if (interior) U.d = fmax(U.d, P.small_dens);
            bool ok;
            const Prim q = cons_to_prim_nb(U, gamma, ok);
            if (k <= i1 + 3 && interior && !ok) bad = true;
            wr[4] = q.r; wu[4] = q.u; wv[4] = q.v; wp[4] = q.p;
        }
        // ---- flatten_x and limit2_x of row k-2 (window index 2)
        double fxn = 1.0, l2n[4] = {0, 0, 0, 0};
        if (k >= i0 && k <= i1 + 3) {
            if (flat)
                fxn = flatten_1d(wp[0], wp[1], wp[3], wp[4], wu[1], wu[3], P.z0, P.z1, P.delta);
            if (limiter != 0) {
                l2n[0] = limit2(wr[1], wr[2], wr[3]);
                l2n[1] = limit2(wu[1], wu[2], wu[3]);
                l2n[2] = limit2(wv[1], wv[2], wv[3]);
                l2n[3] = limit2(wp[1], wp[2], wp[3]);
            }
        }
        // ---- publish Q(k-2) for the y neighbours
        if (k >= i0 + 1 && k <= i1 + 2) {
            double *q = e1 + ((k - 2) & 1) * 4 * MW;
            q[x] = wr[2]; q[MW + x] = wu[2]; q[2 * MW + x] = wv[2]; q[3 * MW + x] = wp[2];
        }
        // ---- artificial viscosity coefficient of the x faces of row k-3
        // (interface.py:366-376: only faces i in [ilo, ihi], j in [jlo, jhi])
        double avx_n = 0.0;
        if (k >= i0 + 3 && k <= i1 + 3) {
            const int i = k - 3;
            if (i >= g.ilo && (i <= g.ihi || (P.avx_hi && i == g.ihi + 1)) && jin) {
                const double divU_x = 0.5 * (dv[x] + dv[x + 1]);
                avx_n = P.cvisc * fmax(-divU_x * P.dx, 0.0);
            }
        }
        // ---- S2: row c = k-3 (window index 1): xi, slopes, tracing
        Cons XMn = one, XPn = one, YMn = one, YPn = one, FxTn = zero;
        if (k >= i0 + 2 && k <= i1 + 3) {
            const int i = k - 3;
            const double *q = e1 + (i & 1) * 4 * MW;
            double xi = 1.0;
            if (flat) {
                // flatten_multid (reconstruction.py:167-183): own coefficient and the
                // one of the UPWIND neighbour (w.r.t. the pressure gradient)
                const double px = (wp[2] - wp[0] > 0) ? fxa : fxn;
                const double py = (q[3 * MW + x + 1] - q[3 * MW + x - 1] > 0) ? e2[4 * MW + x - 1]
                                                                             : e2[4 * MW + x + 1];
                xi = fmin(fmin(fxb, px), fmin(e2[4 * MW + x], py));
            }
            const double q0[4] = {wr[1], wu[1], wv[1], wp[1]};
            const double qm[4] = {wr[0], wu[0], wv[0], wp[0]};
            const double qp[4] = {wr[2], wu[2], wv[2], wp[2]};
            double dqx[4], dqy[4];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                dqx[n] = xi * slope_shared(l2a[n], l2b[n], l2n[n], qm[n], q0[n], qp[n], limiter);
                dqy[n] = xi * slope_shared(e2[n * MW + x - 1], e2[n * MW + x], e2[n * MW + x + 1],
                                           q[n * MW + x - 1], q0[n], q[n * MW + x + 1], limiter);
            }
            Trace lo, hi;
            trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], gamma,
                         P.dtdx, lo, hi);
            XMn = prim_to_cons(Prim{lo.r, lo.un, lo.ut, lo.p}, gamma);
            XPn = prim_to_cons(Prim{hi.r, hi.un, hi.ut, hi.p}, gamma);
            trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], gamma,
                         P.dtdy, lo, hi);
            YMn = prim_to_cons(Prim{lo.r, lo.ut, lo.un, lo.p}, gamma);
            YPn = prim_to_cons(Prim{hi.r, hi.ut, hi.un, hi.p}, gamma);
            if (P.have_src) {   // apply_source_terms, unsplit_fluxes.py:247-330
                const bool ina = (j < g.qy);
                // "ambient" upper boundary: the source ghosts are copies of row jhi
                // (BC.py:159-160), not the sources of the ambient ghost state
                const int js = (P.amb_yhi && j > g.jhi) ? g.jhi : j;
                const size_t kc = (size_t)(ina ? i : g.qx - 1) * p + (ina ? js : g.qy - 1);
                Cons Ug{Uin[kc], 0.0, 0.0, Uin[3 * pl + kc]};
                if (i >= g.ilo && i <= g.ihi && js >= g.jlo && js <= g.jhi)
                    Ug.d = fmax(Ug.d, P.small_dens);
                const double sgn =
                    ((j < g.jlo && P.refl_ylo) || (j > g.jhi && P.refl_yhi)) ? -1.0 : 1.0;
                const double hp =
                    P.heat ? P.heat[(size_t)(ina ? i : g.qx - 1) * p + (ina ? j : g.qy - 1)] : 0.0;
                add_grav_to_state(XMn, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
                add_grav_to_state(XPn, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
                add_grav_to_state(YMn, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
                add_grav_to_state(YPn, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
            }
            ring_put(e3, x, YPn);
            // transverse x flux on the lower face of row k-3
            if (k >= i0 + 3)
                FxTn = from_nf(riemann_face<SOLVER>(to_nf(XPs, true), to_nf(XMn, true), gamma, true,
                                                    P.solid_xl && i == g.ilo), true);
        }
        // ---- transverse correction of the y states of row k-4 (FxT of rows k-4, k-3)
        if (k >= i0 + 4 && k <= i1 + 3) {
            YMc = corr(YMs, FxTn, FxTp, hdtV, Ax);
            const Cons YPc = corr(YPs, FxTn, FxTp, hdtV, Ax);
            ring_put(e5, x, YPc);
        }
        // ---- S4: row k-4: transverse correction of the x states, final x flux
        Cons Fxn = zero, XPc = one;
        if (k >= i0 + 3) {
            const int i = k - 4;
            const Cons Flo = ring_get(e4, x);          // FyT at (i, j)
            const Cons Fhi = ring_get(e4, x + 1);      // FyT at (i, j+1)
            const Cons XMc = corr(XMs, Fhi, Flo, hdtV, Ay);
            XPc = corr(XPs, Fhi, Flo, hdtV, Ay);
            if (k >= i0 + 4) {
                Fxn = from_nf(riemann_face<SOLVER>(to_nf(XPcp, true), to_nf(XMc, true), gamma, true,
                                                   P.solid_xl && i == g.ilo), true);
                Fxn.d += avx_c * (Uem.d - Ue.d);
                Fxn.E += avx_c * (Uem.E - Ue.E);
                Fxn.mx += avx_c * (Uem.mx - Ue.mx);
                Fxn.my += avx_c * (Uem.my - Ue.my);
            }
            ring_put(e7, x, Ue);
        }
        // ---- S6: conservative update of row k-5 + CFL of the new state
        if (k >= i0 + 5 && jout) {
            const int i = k - 5;
            const double dtdV = P.dtdV;
            const Cons Fy = ring_get(e6, x);
            const Cons Fyh = ring_get(e6, x + 1);
            const Cons &Uc = Uem;
            Cons Un;   // simulation.py:377-384
            Un.d = Uc.d + dtdV * (Fxp.d * Ax - Fxn.d * Ax + Fy.d * Ay - Fyh.d * Ay);
            Un.E = Uc.E + dtdV * (Fxp.E * Ax - Fxn.E * Ax + Fy.E * Ay - Fyh.E * Ay);
            Un.mx = Uc.mx + dtdV * (Fxp.mx * Ax - Fxn.mx * Ax + Fy.mx * Ay - Fyh.mx * Ay);
            Un.my = Uc.my + dtdV * (Fxp.my * Ax - Fxn.my * Ax + Fy.my * Ay - Fyh.my * Ay);
            const size_t ko = (size_t)i * p + j;
            if (P.have_src)   // simulation.py:406-423
                grav_update(Un, Uc, P.grav, P.dt, P.heat_rate, P.heat ? P.heat[ko] : 0.0);
            Uout[ko] = Un.d; Uout[pl + ko] = Un.E; Uout[2 * pl + ko] = Un.mx; Uout[3 * pl + ko] = Un.my;
            cfl = fmin(cfl, cfl_cell(Un, gamma, P.dx, P.dy));
        }
        // hand the rows on
#pragma unroll
        for (int n = 0; n < 4; n++) { l2a[n] = l2b[n]; l2b[n] = l2n[n]; }
        fxa = fxb; fxb = fxn;
        XMs = XMn; XPs = XPn; YMs = YMn; YPs = YPn;
        FxTp = FxTn;
        XPcp = XPc;
        Fxp = Fxn;
        avx_c = avx_n;
        __syncthreads();

        // ================= half B =================
        // ---- S5: final y flux on the lower face of row k-4 (before avy_c moves on)
        if (k >= i0 + 4 && k <= i1 + 3) {
            Cons Fy = from_nf(riemann_face<SOLVER>(to_nf(ring_get(e5, x - 1), false),
                                                   to_nf(YMc, false), gamma, false,
                                                   P.solid_yl && j == g.jlo), false);
            const Cons Umy = ring_get(e7, x - 1);
            Fy.d += avy_c * (Umy.d - Ue.d);
            Fy.E += avy_c * (Umy.E - Ue.E);
            Fy.mx += avy_c * (Umy.mx - Ue.mx);
            Fy.my += avy_c * (Umy.my - Ue.my);
            ring_put(e6, x, Fy);
        }
        // ---- S3: transverse y flux on the lower face of row k-3
        if (k >= i0 + 2 && k <= i1 + 3) {
            const Cons FyT = from_nf(riemann_face<SOLVER>(to_nf(ring_get(e3, x - 1), false),
                                                          to_nf(YMs, false), gamma, false,
                                                          P.solid_yl && j == g.jlo), false);
            ring_put(e4, x, FyT);
        }
        // ---- S1: row r = k-2 (window index 2): limit2_y, flatten_y, vertex div(U)
        if (k >= i0 + 1 && k <= i1 + 2) {
            const int r = k - 2;
            const double *q = e1 + (r & 1) * 4 * MW;
            const double um = q[MW + x - 1], vm = q[2 * MW + x - 1];
            if (limiter != 0) {
                e2[x] = limit2(q[x - 1], wr[2], q[x + 1]);
                e2[MW + x] = limit2(um, wu[2], q[MW + x + 1]);
                e2[2 * MW + x] = limit2(vm, wv[2], q[2 * MW + x + 1]);
                e2[3 * MW + x] = limit2(q[3 * MW + x - 1], wp[2], q[3 * MW + x + 1]);
            }
            if (flat)
                e2[4 * MW + x] = flatten_1d(q[3 * MW + x - 2], q[3 * MW + x - 1], q[3 * MW + x + 1],
                                            q[3 * MW + x + 2], vm, q[2 * MW + x + 1], P.z0, P.z1,
                                            P.delta);
            // vertex divergence at (r-1/2, j-1/2), interface.py:312-330
            const double Dn = div_u_vertex(wu[2], um, wu[1], up, wv[2], wv[1], vm, vp, P.dx, P.dy);
            dv[x] = Dn;
            // coefficient of the y face (r-1, j): only faces j in [jlo, jhi], i in [ilo, ihi]
            double avy_n = 0.0;
            if (j >= g.jlo && (j <= g.jhi || (P.avy_hi && j == g.jhi + 1)) && row_in(r - 1)) {
                const double divU_y = 0.5 * (Dp + Dn);
                avy_n = P.cvisc * fmax(-divU_y * P.dy, 0.0);
            }
            avy_c = avy_n;
            Dp = Dn;
            up = um; vp = vm;
        }
        __syncthreads();
    }
    if (bad) atomicOr(flag, 1);
    cfl = block_reduce_min(cfl);
    if (tj == 0) partial[blockIdx.x] = cfl;
}

// rows per strip: the strip count that minimises (rounds of resident
// workgroups) x (iterations per strip)
static int march_rows(int nx, int ncb, int slots)
{
    int bestL = nx, best = 1 << 30;
    for (int nsb = 1; nsb <= nx; nsb++) {
        const int L = (nx + nsb - 1) / nsb;
        if (L < 16 && nsb > 1) break;
        const int wgs = ncb * ((nx + L - 1) / L);
        const int rounds = (wgs + slots - 1) / slots;
        const int cost = rounds * (L + 9);
        if (cost < best) { best = cost; bestL = L; }
    }
    return bestL;
}

int comp_step_march(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Uout;
    PYRO_TRY(fused_prepare(s, p, dt, P, Uin, Uout));
    P.ncb = (g.ny + MOUT - 1) / MOUT;
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    P.L = march_rows(g.nx, P.ncb, 2 * cus);
    if (p->march_rows > 0) P.L = p->march_rows < g.nx ? p->march_rows : g.nx;
    const int nsb = (g.nx + P.L - 1) / P.L;
    const int nwg = P.ncb * nsb;
    PYRO_TRY(c->reduce.ensure((nwg + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    using KernelT = void (*)(const double *, double *, Geom, FP, int *, double *);
    static const KernelT kernels[3][2] = {
        {k_ctu_march<0, false>, k_ctu_march<0, true>},
        {k_ctu_march<1, false>, k_ctu_march<1, true>},
        {k_ctu_march<2, false>, k_ctu_march<2, true>}};
#ifndef PYRO_EMU
    static bool attr_set = false;
    if (!attr_set) {
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 2; b++)
                PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)kernels[a][b],
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)MLDS_BYTES));
        attr_set = true;
    }
#endif
    const int solver = (p->riemann == 1 || p->riemann == 2) ? p->riemann : 0;
    const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
    PYRO_LAUNCH(c, "k_ctu_march", kernels[solver][std_rec], dim3(nwg), dim3(MNT), MLDS_BYTES,
                (const double *)Uin, Uout, g, P, s->d_flag, part);
    return fused_finish(s, part, nwg);
}

}  // namespace PYRO_NS
}  // namespace pyro
