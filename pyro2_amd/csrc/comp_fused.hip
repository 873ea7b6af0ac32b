// Compressible CTU + HLLC step as ONE kernel per time step (kernel_set 1).
//
// Same arithmetic as the staged kernels of compressible.hip (shared per-cell
// functions in hydro.h), but every intermediate lives in LDS or registers:
// HBM traffic per cell update is the algorithmic 64 B (read 4 + write 4
// conserved doubles) plus apron re-reads that hit L2.
//
// Decomposition: a workgroup of BI x BJ threads owns the (BI-2) x (BJ-2)
// interior cells of its tile; thread (ti,tj) is cell (i0-1+ti, j0-1+tj), i.e.
// the tile grown by one cell (the region on which face states are needed,
// SURVEY.md 7).  Phases, separated by workgroup barriers:
//   0  stage U (tile + 4-cell apron) -> primitives Q in LDS
//   1  per cell: flattening, limited slopes, characteristic tracing ->
//      4 conserved face states in registers; upper states + vertex div(U)
//      -> LDS
//   2  transverse Riemann problems on the cell's lower faces -> LDS
//   3  per cell: transverse correction of its own 4 states; upper states
//      -> LDS
//   4  final Riemann problems + artificial viscosity on the lower faces
//      -> LDS
//   5  conservative update of the interior cells into the second state
//      buffer + CFL minimum of the new state
// LDS: max(Q, fluxes) + upper states + div(U); for 16x32 threads
// 32 + 32 + 4 KiB = 68 KiB, two workgroups (16 waves, 4 per SIMD) per CU.
//
// Compiled twice like compressible.hip (PYRO_FAST = 0 / 1).
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

constexpr int FBI = 16;                 // threads along i (rows)
constexpr int FBJ = 32;                 // threads along j (fast axis)
constexpr int FNT = FBI * FBJ;          // 512
constexpr int FTI = FBI - 2;            // interior cells per tile
constexpr int FTJ = FBJ - 2;
constexpr int FQH = FBI + 6;            // Q tile rows  (tile + 4 apron)
constexpr int FQW = FBJ + 6;
constexpr int FQN = FQH * FQW;          // cells in the Q tile
constexpr int FBUF0 = (4 * FQN > 8 * FNT) ? 4 * FQN : 8 * FNT;   // Q | FT | F
constexpr int FLDS_DOUBLES = FBUF0 + 8 * FNT + FNT;
constexpr size_t FLDS_BYTES = (size_t)FLDS_DOUBLES * sizeof(double);

#include "fused_common.h"

// Cell (gi, gj) of the old state as fill_BC_all leaves it: a ghost cell is read from the
// cell its boundary rule copies from (x fill, then y fill: both maps; itself without
// fuse_fill), with the sign of the variables that reflect oddly on the sides crossed.
// sd: the sides the cell lies beyond (0: interior).
__device__ __forceinline__ Cons load_cons_bc(const double *__restrict__ Uin, const Geom &g,
                                             const FP &P, int gi, int gj, unsigned &sd)
{
    const int si = bc_src(P.mr, gi, g.ilo, g.ihi), sj = bc_src(P.mc, gj, g.jlo, g.jhi);
    const size_t k = (size_t)si * g.pitch + sj, pl = g.plane;
    sd = (gi < g.ilo ? 1u : 0u) | (gi > g.ihi ? 2u : 0u) | (gj < g.jlo ? 4u : 0u) |
         (gj > g.jhi ? 8u : 0u);
    Cons U{Uin[k], Uin[pl + k], Uin[2 * pl + k], Uin[3 * pl + k]};
    U.d = odd_sides(P.odd & sd) ? -U.d : U.d;
    U.E = odd_sides((P.odd >> 4) & sd) ? -U.E : U.E;
    U.mx = odd_sides((P.odd >> 8) & sd) ? -U.mx : U.mx;
    U.my = odd_sides((P.odd >> 12) & sd) ? -U.my : U.my;
    return U;
}

__device__ __forceinline__ Cons lds_get(const double *b, int t)
{
    return Cons{b[t], b[FNT + t], b[2 * FNT + t], b[3 * FNT + t]};
}
__device__ __forceinline__ void lds_put(double *b, int t, const Cons &U)
{
    b[t] = U.d; b[FNT + t] = U.E; b[2 * FNT + t] = U.mx; b[3 * FNT + t] = U.my;
}
// developer timing aid (tools/fused_phases.sh): -DPYRO_FUSED_STOP=k ends the
// kernel after phase k, storing one value that depends on the phase's results
#ifndef PYRO_FUSED_STOP
#define PYRO_FUSED_STOP 99
#endif
#define PYRO_PHASE_END(k, val)                                                   \
    if (PYRO_FUSED_STOP == (k)) {                                                \
        if (i < g.qx && j < g.qy) Uout[(size_t)i * p + j] = (val);               \
        return;                                                                  \
    }

#ifndef PYRO_FUSED_MINW
// waves per SIMD the register allocation must allow: 4 = two 512-thread
// workgroups per CU (128 VGPRs, 48 B/lane scratch).  Measured at 8192^2:
// 6.75 ms vs 9.48 ms with one workgroup per CU (146 VGPRs, no scratch) -- the
// second workgroup fills the VALU while the first sits in a barrier.
#define PYRO_FUSED_MINW 4
#endif

// STD: the default reconstruction (limiter 2 = 4th-order MC, flattening on) as
// compile-time constants: straight-line code the compiler schedules across the
// eight limited slopes (3.84 -> 3.61 ms at 8192^2; also making "no sources" a
// constant gave 3.59 and was not kept: problems with gravity use this instance
// too); STD = false reads both from the parameters
template <int SOLVER, bool STD = false>   // compressible.riemann: 0 HLLC, 1 CGF, 2 HLLC_lm
__global__ __launch_bounds__(FNT, PYRO_FUSED_MINW) void k_ctu_fused(const double *__restrict__ Uin,
                                                   double *__restrict__ Uout, Geom g, FP P_in,
                                                   int *__restrict__ flag,
                                                   double *__restrict__ partial,
                                                   const StepScalars *__restrict__ SC)
{
    HIP_DYNAMIC_SHARED(double, lds)
    FP P = P_in;
    if (SC) {   // device-side run: this step's dt lives in device memory
        if (!SC->active) {
            // past tmax / after an invalid state: nothing happens (the host picks the
            // buffer that holds the last state that did advance, comp_evolve)
            if (threadIdx.x == 0 && threadIdx.y == 0)
                partial[xcd_tile(blockIdx.x, P.ntiles)] = INFINITY;
            return;
        }
        P.dt = SC->dt; P.dtdx = SC->dtdx; P.dtdy = SC->dtdy; P.hdtV = SC->hdtV; P.dtdV = SC->dtdV;
    }
    double *B0 = lds;                 // Q (phase 0-1) | FT (2-3) | F (4-5)
    double *S = lds + FBUF0;          // upper face states XP(0..3), YP(4..7)
    double *D = S + 8 * FNT;          // vertex div(U)

    // (a walk that keeps every XCD on a contiguous band of tiles for ANY tile
    // count -- xcd_tile falls back to the identity when it is not divisible by
    // 8, as at 8192^2 and 16384^2 -- and goes through super-columns of 8..64
    // tiles was measured: 3.85 vs 3.82 ms at 8192^2.  The apron re-reads are
    // served by the MALL; phase 0 is bound by latency, not by fabric traffic.)
    const int tile = xcd_tile(blockIdx.x, P.ntiles);
    const int i0 = g.ilo + (tile / P.ntj) * FTI;
    const int j0 = g.jlo + (tile % P.ntj) * FTJ;
    const int tj = threadIdx.x, ti = threadIdx.y;
    const int t = ti * FBJ + tj;
    const int i = i0 - 1 + ti, j = j0 - 1 + tj;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const double gamma = P.gamma;

    // ---- phase 0: U -> Q (rho,u,v,p) for the tile + 4-cell apron --------
    // All global loads of the thread (2 cells x 4 planes) are issued before the
    // first use and the conversion is branch-free: with the `if (U.d != 0)` of
    // cons_to_prim the compiler sank the loads of E, mx, my into the branch, so
    // a workgroup went through four dependent HBM round trips here (density,
    // rest, density, rest) instead of one.
    bool bad = false;
    {
        constexpr int NIT = (FQN + FNT - 1) / FNT;
        Cons Ul[NIT];
        bool act[NIT], interior[NIT];
#pragma unroll
        for (int n = 0; n < NIT; n++) {
            const int idx = t + n * FNT;
            act[n] = idx < FQN;
            const int ii = act[n] ? idx : t;       // idle lanes re-read their first cell
            const int r = ii / FQW, c = ii - r * FQW;
            int gi = i0 - 4 + r, gj = j0 - 4 + c;
            const bool inarr = act[n] && gi < g.qx && gj < g.qy;
            gi = (gi < g.qx) ? gi : g.qx - 1;   // ragged last tiles: clamp, unused
            gj = (gj < g.qy) ? gj : g.qy - 1;
            // ghost cells: the cell the boundary rule copies from (itself without
            // fuse_fill), with the sign of reflect-odd variables -- the value
            // fill_BC_all would have stored (x fill, then y fill: both maps)
            unsigned sd;
            const Cons U = load_cons_bc(Uin, g, P, gi, gj, sd);
            Ul[n] = U;
            interior[n] = (sd == 0);
            // the ghost frame of the new state: what the old state's ghost cells hold
            // (after the fill, when it is folded in) -- as in the reference, where the
            // update leaves the ghost cells of the array alone.  Tiles whose aprons
            // overlap write the same values.
            if (inarr && sd != 0) {
                const size_t ko = (size_t)gi * p + gj;
                Uout[ko] = U.d; Uout[pl + ko] = U.E; Uout[2 * pl + ko] = U.mx; Uout[3 * pl + ko] = U.my;
            }
        }
#pragma unroll
        for (int n = 0; n < NIT; n++) {
            const int idx = t + n * FNT;
            Cons U = Ul[n];
            if (interior[n]) U.d = fmax(U.d, P.small_dens);      // clean_state
            bool ok;
            const Prim q = cons_to_prim_nb(U, gamma, ok);
            if (act[n] && interior[n] && !ok) bad = true;
            if (act[n]) {
                B0[idx] = q.r; B0[FQN + idx] = q.u; B0[2 * FQN + idx] = q.v; B0[3 * FQN + idx] = q.p;
            }
        }
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    PYRO_PHASE_END(0, B0[(ti + 3) * FQW + (tj + 3)] + B0[3 * FQN + (ti + 3) * FQW + (tj + 3)])

    // ---- phase 1: xi, slopes, tracing for the thread's own cell --------
    Cons XM, XP, YM, YP;
    {
        const int qc = (ti + 3) * FQW + (tj + 3);   // own cell in the Q tile
        const double *Qr = B0, *Qu = B0 + FQN, *Qv = B0 + 2 * FQN, *Qp = B0 + 3 * FQN;
        double xi = 1.0;
        if (STD || P.use_flattening) {
            // flatten_multid (reconstruction.py:167-183): own coefficient and
            // the one of the UPWIND neighbour (w.r.t. the pressure gradient)
            // in each direction -- the downwind one is never selected
            const int sx = (Qp[qc + FQW] - Qp[qc - FQW] > 0) ? -FQW : FQW;
            const int sy = (Qp[qc + 1] - Qp[qc - 1] > 0) ? -1 : 1;
            const int cx = qc + sx, cy = qc + sy;
            const double xix = flatten_1d(Qp[qc - 2 * FQW], Qp[qc - FQW], Qp[qc + FQW],
                                          Qp[qc + 2 * FQW], Qu[qc - FQW], Qu[qc + FQW], P.z0, P.z1,
                                          P.delta);
            const double px = flatten_1d(Qp[cx - 2 * FQW], Qp[cx - FQW], Qp[cx + FQW],
                                         Qp[cx + 2 * FQW], Qu[cx - FQW], Qu[cx + FQW], P.z0, P.z1,
                                         P.delta);
            const double xiy = flatten_1d(Qp[qc - 2], Qp[qc - 1], Qp[qc + 1], Qp[qc + 2],
                                          Qv[qc - 1], Qv[qc + 1], P.z0, P.z1, P.delta);
            const double py = flatten_1d(Qp[cy - 2], Qp[cy - 1], Qp[cy + 1], Qp[cy + 2],
                                         Qv[cy - 1], Qv[cy + 1], P.z0, P.z1, P.delta);
            xi = fmin(fmin(xix, px), fmin(xiy, py));
        }
        double q0[4], dqx[4], dqy[4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const double *a = B0 + n * FQN;
            q0[n] = a[qc];
            dqx[n] = xi * limited_slope(a[qc - 2 * FQW], a[qc - FQW], a[qc], a[qc + FQW],
                                        a[qc + 2 * FQW], STD ? 2 : P.limiter);
            dqy[n] = xi * limited_slope(a[qc - 2], a[qc - 1], a[qc], a[qc + 1], a[qc + 2],
                                        STD ? 2 : P.limiter);
        }
        Trace lo, hi;
        trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], gamma,
                     P.dtdx, lo, hi);
        XM = prim_to_cons(Prim{lo.r, lo.un, lo.ut, lo.p}, gamma);
        XP = prim_to_cons(Prim{hi.r, hi.un, hi.ut, hi.p}, gamma);
        trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], gamma,
                     P.dtdy, lo, hi);
        YM = prim_to_cons(Prim{lo.r, lo.ut, lo.un, lo.p}, gamma);
        YP = prim_to_cons(Prim{hi.r, hi.ut, hi.un, hi.p}, gamma);
        // vertex divergence at (i-1/2, j-1/2), interface.py:312-330
        D[t] = div_u_vertex(Qu[qc], Qu[qc - 1], Qu[qc - FQW], Qu[qc - FQW - 1], Qv[qc],
                            Qv[qc - FQW], Qv[qc - 1], Qv[qc - FQW - 1], P.dx, P.dy);
        if (P.have_src) {   // apply_source_terms, unsplit_fluxes.py:247-330
            const bool ina = (i < g.qx && j < g.qy);
            // "ambient" upper boundary: the source ghosts are copies of row jhi
            // (BC.py:159-160), not the sources of the ambient ghost state
            const int js = (P.amb_yhi && j > g.jhi) ? g.jhi : j;
            const size_t kc = (size_t)(ina ? i : g.qx - 1) * p + (ina ? js : g.qy - 1);
            Cons Ug{Uin[kc], 0.0, 0.0, Uin[3 * pl + kc]};
            if (i >= g.ilo && i <= g.ihi && js >= g.jlo && js <= g.jhi)
                Ug.d = fmax(Ug.d, P.small_dens);
            const double sgn =
                ((j < g.jlo && P.refl_ylo) || (j > g.jhi && P.refl_yhi)) ? -1.0 : 1.0;
            const double hp = P.heat ? P.heat[(size_t)(ina ? i : g.qx - 1) * p + (ina ? j : g.qy - 1)]
                                     : 0.0;
            add_grav_to_state(XM, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
            add_grav_to_state(XP, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
            add_grav_to_state(YM, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
            add_grav_to_state(YP, Ug, P.grav, P.dt, sgn, P.heat_rate, hp);
        }
        lds_put(S, t, XP);
        lds_put(S + 4 * FNT, t, YP);
    }
    __syncthreads();   // Q is dead from here on; B0 becomes the flux buffer
    PYRO_PHASE_END(1, XM.d + XM.E + XM.mx + XM.my + YM.d + YM.E + YM.mx + YM.my + S[t] + S[7 * FNT + t] + D[t])

    // ---- phase 2: transverse Riemann problems on the lower faces --------
    Cons FxT{0, 0, 0, 0}, FyT{0, 0, 0, 0};
    if (ti >= 1)
        FxT = from_nf(riemann_face<SOLVER>(to_nf(lds_get(S, t - FBJ), true), to_nf(XM, true), gamma,
                                           true, P.solid_xl && i == g.ilo), true);
    if (tj >= 1)
        FyT = from_nf(riemann_face<SOLVER>(to_nf(lds_get(S + 4 * FNT, t - 1), false),
                                           to_nf(YM, false), gamma, false,
                                           P.solid_yl && j == g.jlo), false);
    lds_put(B0, t, FxT);
    lds_put(B0 + 4 * FNT, t, FyT);
    __syncthreads();
    PYRO_PHASE_END(2, FxT.d + FxT.E + FxT.mx + FxT.my + FyT.d + FyT.E + FyT.mx + FyT.my + XM.d + YM.E)

    // ---- phase 3: transverse correction of the cell's own states --------
    const double hdtV = P.hdtV;                          // hdt / V
    const double Ax = P.dy, Ay = P.dx;
    if (tj >= 1 && tj <= FBJ - 2) {
        const Cons Fhi = lds_get(B0 + 4 * FNT, t + 1);   // F_yT at (i, j+1)
        XM = corr(XM, Fhi, FyT, hdtV, Ay);
        XP = corr(XP, Fhi, FyT, hdtV, Ay);
    }
    if (ti >= 1 && ti <= FBI - 2) {
        const Cons Fhi = lds_get(B0, t + FBJ);           // F_xT at (i+1, j)
        YM = corr(YM, Fhi, FxT, hdtV, Ax);
        YP = corr(YP, Fhi, FxT, hdtV, Ax);
    }
    // (all reads of the uncorrected upper states happened before the barrier
    // that closed phase 2; phase 3 itself only reads the flux buffer)
    lds_put(S, t, XP);
    lds_put(S + 4 * FNT, t, YP);
    __syncthreads();
    PYRO_PHASE_END(3, XM.d + XM.E + XM.mx + XM.my + YM.d + YM.E + YM.mx + YM.my + S[t] + S[7 * FNT + t])

    // ---- phase 4: final Riemann problems + artificial viscosity ---------
    const bool in_arr = (i < g.qx && j < g.qy);
    const int ic = in_arr ? i : g.qx - 1, jc = in_arr ? j : g.qy - 1;
    const size_t k = (size_t)ic * p + jc;
    unsigned sdc;
    Cons Uc = load_cons_bc(Uin, g, P, ic, jc, sdc);
    const bool cell_interior = (i >= g.ilo && i <= g.ihi && j >= g.jlo && j <= g.jhi);
    if (cell_interior) Uc.d = fmax(Uc.d, P.small_dens);
    Cons Fx{0, 0, 0, 0}, Fy{0, 0, 0, 0};
    const double d00 = D[t];
    // the lower neighbours' old states for the artificial-viscosity terms:
    // loaded (L2 hits) before the Riemann problems so that the latency is
    // covered by them.  k - p / k - 1 are inside the array for every thread.
    Cons Umx = load_cons_bc(Uin, g, P, ic - 1, jc, sdc);
    Cons Umy = load_cons_bc(Uin, g, P, ic, jc - 1, sdc);
    double avx = 0.0, avy = 0.0;
    // interface.py:366-376: only faces i in [ilo, ihi], j in [jlo, jhi]
    if (ti >= 1 && tj >= 1 && tj <= FBJ - 2 && i >= g.ilo &&
        (i <= g.ihi || (P.avx_hi && i == g.ihi + 1)) && j >= g.jlo && j <= g.jhi) {
        const double divU_x = 0.5 * (d00 + D[t + 1]);
        avx = P.cvisc * fmax(-divU_x * P.dx, 0.0);
    }
    if (tj >= 1 && ti >= 1 && ti <= FBI - 2 && j >= g.jlo &&
        (j <= g.jhi || (P.avy_hi && j == g.jhi + 1)) && i >= g.ilo && i <= g.ihi) {
        const double divU_y = 0.5 * (d00 + D[t + FBJ]);
        avy = P.cvisc * fmax(-divU_y * P.dy, 0.0);
    }
    if (i - 1 >= g.ilo && i - 1 <= g.ihi && j >= g.jlo && j <= g.jhi)
        Umx.d = fmax(Umx.d, P.small_dens);
    if (i >= g.ilo && i <= g.ihi && j - 1 >= g.jlo && j - 1 <= g.jhi)
        Umy.d = fmax(Umy.d, P.small_dens);
    if (ti >= 1 && tj >= 1 && tj <= FBJ - 2) {           // x face (i, j)
        Fx = from_nf(riemann_face<SOLVER>(to_nf(lds_get(S, t - FBJ), true), to_nf(XM, true), gamma,
                                          true, P.solid_xl && i == g.ilo), true);
        Fx.d += avx * (Umx.d - Uc.d);
        Fx.E += avx * (Umx.E - Uc.E);
        Fx.mx += avx * (Umx.mx - Uc.mx);
        Fx.my += avx * (Umx.my - Uc.my);
    }
    if (tj >= 1 && ti >= 1 && ti <= FBI - 2) {           // y face (i, j)
        Fy = from_nf(riemann_face<SOLVER>(to_nf(lds_get(S + 4 * FNT, t - 1), false),
                                          to_nf(YM, false), gamma, false,
                                          P.solid_yl && j == g.jlo), false);
        Fy.d += avy * (Umy.d - Uc.d);
        Fy.E += avy * (Umy.E - Uc.E);
        Fy.mx += avy * (Umy.mx - Uc.mx);
        Fy.my += avy * (Umy.my - Uc.my);
    }
    lds_put(B0, t, Fx);            // FT was last read before the barrier above
    lds_put(B0 + 4 * FNT, t, Fy);
    __syncthreads();
    PYRO_PHASE_END(4, Fx.d + Fx.E + Fx.mx + Fx.my + Fy.d + Fy.E + Fy.mx + Fy.my)

    // ---- phase 5: conservative update + CFL of the new state -----------
    double cfl = INFINITY;
    if (ti >= 1 && ti <= FBI - 2 && tj >= 1 && tj <= FBJ - 2 && cell_interior) {
        const double dtdV = P.dtdV;
        const Cons Fxh = lds_get(B0, t + FBJ);
        const Cons Fyh = lds_get(B0 + 4 * FNT, t + 1);
        Cons Un;   // simulation.py:377-384
        Un.d = Uc.d + dtdV * (Fx.d * Ax - Fxh.d * Ax + Fy.d * Ay - Fyh.d * Ay);
        Un.E = Uc.E + dtdV * (Fx.E * Ax - Fxh.E * Ax + Fy.E * Ay - Fyh.E * Ay);
        Un.mx = Uc.mx + dtdV * (Fx.mx * Ax - Fxh.mx * Ax + Fy.mx * Ay - Fyh.mx * Ay);
        Un.my = Uc.my + dtdV * (Fx.my * Ax - Fxh.my * Ax + Fy.my * Ay - Fyh.my * Ay);
        if (P.have_src)   // simulation.py:406-423
            grav_update(Un, Uc, P.grav, P.dt, P.heat_rate, P.heat ? P.heat[k] : 0.0);
        Uout[k] = Un.d; Uout[pl + k] = Un.E; Uout[2 * pl + k] = Un.mx; Uout[3 * pl + k] = Un.my;
        cfl = cfl_cell(Un, gamma, P.dx, P.dy);
    }
    cfl = block_reduce_min(cfl);
    if (t == 0) partial[tile] = cfl;
}

// ===========================================================================
// SphericalPolar grids (x = r, y = theta; mesh/patch.py:242-312): the whole step of
// compressible.hip's staged spherical set (k_prim, k_sph_src + ghost fill of the sources, k_xi,
// k_sph_states, k_sph_riemann_t, k_sph_final, k_sph_update: nine launches through 45 work
// planes) as ONE launch of the tile kernel above with the geometry terms -- the same phases,
// the same expressions on the same operands (bit-identical in the bit-faithful build):
//  - tracing with per-cell dt / Lx, dt / Ly and the geometric source (interface.py:106, 215-234);
//  - external sources (radial gravity + the geometric terms, simulation.py:117-124): evaluated
//    for the thread's own cell -- a ghost cell takes the value of the cell its boundary rule
//    copies from, with the variable's sign, like the reference's ghost-filled source arrays;
//  - CGF interface states whose pressure stays out of the area-weighted flux difference and
//    enters as a gradient (riemann.py:1092-1096, 1156-1171; unsplit_fluxes.py:411-488): the
//    face pressures travel in two more LDS planes;
//  - transverse correction, conservative update and CFL with the area / volume / length
//    arrays, the vertex divergence of interface.py:331-364, the source predictor-corrector
//    (simulation.py:330-423).
// Boundaries: outflow / reflect / periodic sides (index maps of the tile kernel); everything
// else steps through the staged set.
// ===========================================================================
#include "sph_common.h"
constexpr int FLDS_DOUBLES_SPH = FLDS_DOUBLES + 2 * FNT;       // + the face pressures
constexpr size_t FLDS_BYTES_SPH = (size_t)FLDS_DOUBLES_SPH * sizeof(double);

template <bool STD, bool FAC>
__global__ __launch_bounds__(FNT, PYRO_FUSED_MINW) void k_ctu_fused_sph(const double *__restrict__ Uin,
                                                       double *__restrict__ Uout, Geom g, FP P_in,
                                                       SphG G, int *__restrict__ flag,
                                                       double *__restrict__ partial,
                                                       const StepScalars *__restrict__ SC)
{
    HIP_DYNAMIC_SHARED(double, lds)
    FP P = P_in;
    if (SC) {   // device-side run (pyrohip_comp_evolve): this step's dt lives in device memory
        if (!SC->active) {
            if (threadIdx.x == 0 && threadIdx.y == 0)
                partial[xcd_tile(blockIdx.x, P.ntiles)] = INFINITY;
            return;
        }
        P.dt = SC->dt;
    }
    double *B0 = lds;                 // Q (phase 0-1) | FT (2-3) | F (4-5)
    double *S = lds + FBUF0;          // upper face states XP(0..3), YP(4..7)
    double *D = S + 8 * FNT;          // vertex div(U)
    double *PT = D + FNT;             // face pressures: x faces, y faces
    const int tile = xcd_tile(blockIdx.x, P.ntiles);
    const int i0 = g.ilo + (tile / P.ntj) * FTI;
    const int j0 = g.jlo + (tile % P.ntj) * FTJ;
    const int tj = threadIdx.x, ti = threadIdx.y;
    const int t = ti * FBJ + tj;
    const int i = i0 - 1 + ti, j = j0 - 1 + tj;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const double gamma = P.gamma;

    // ---- phase 0: U -> Q for the tile + 4-cell apron (as k_ctu_fused) --------
    bool bad = false;
    {
        constexpr int NIT = (FQN + FNT - 1) / FNT;
        Cons Ul[NIT];
        bool act[NIT], interior[NIT];
#pragma unroll
        for (int n = 0; n < NIT; n++) {
            const int idx = t + n * FNT;
            act[n] = idx < FQN;
            const int ii = act[n] ? idx : t;
            const int r = ii / FQW, c = ii - r * FQW;
            int gi = i0 - 4 + r, gj = j0 - 4 + c;
            const bool inarr = act[n] && gi < g.qx && gj < g.qy;
            gi = (gi < g.qx) ? gi : g.qx - 1;
            gj = (gj < g.qy) ? gj : g.qy - 1;
            unsigned sd;
            const Cons U = load_cons_bc(Uin, g, P, gi, gj, sd);
            Ul[n] = U;
            interior[n] = (sd == 0);
            if (inarr && sd != 0) {       // the ghost frame of the new state: the old ghost cells
                const size_t ko = (size_t)gi * p + gj;
                Uout[ko] = U.d; Uout[pl + ko] = U.E; Uout[2 * pl + ko] = U.mx; Uout[3 * pl + ko] = U.my;
            }
        }
#pragma unroll
        for (int n = 0; n < NIT; n++) {
            const int idx = t + n * FNT;
            Cons U = Ul[n];
            if (interior[n]) U.d = fmax(U.d, P.small_dens);      // clean_state
            bool ok;
            const Prim q = cons_to_prim_nb(U, gamma, ok);
            if (act[n] && interior[n] && !ok) bad = true;
            if (act[n]) {
                B0[idx] = q.r; B0[FQN + idx] = q.u; B0[2 * FQN + idx] = q.v; B0[3 * FQN + idx] = q.p;
            }
        }
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();

    // the thread's own cell in the arrays (ragged last tiles: clamped, unused)
    const bool in_arr = (i < g.qx && j < g.qy);
    const int ic = in_arr ? i : g.qx - 1, jc = in_arr ? j : g.qy - 1;
    const size_t k = (size_t)ic * p + jc;
    // (cells one row / column up: inside the array for every cell the tile uses)
    const int ipc = (ic + 1 < g.qx) ? ic + 1 : ic, jpc = (jc + 1 < g.qy) ? jc + 1 : jc;
    const SphAt<FAC> GA{G, p, P.dx};
    const double hdt = 0.5 * P.dt;

    // ---- phase 1: xi, slopes, tracing, sources for the thread's own cell --------
    Cons XM, XP, YM, YP;
    {
        const int qc = (ti + 3) * FQW + (tj + 3);
        const double *Qr = B0, *Qu = B0 + FQN, *Qv = B0 + 2 * FQN, *Qp = B0 + 3 * FQN;
        (void)Qr;
        double xi = 1.0;
        if (STD || P.use_flattening) {
            const int sx = (Qp[qc + FQW] - Qp[qc - FQW] > 0) ? -FQW : FQW;
            const int sy = (Qp[qc + 1] - Qp[qc - 1] > 0) ? -1 : 1;
            const int cx = qc + sx, cy = qc + sy;
            const double xix = flatten_1d(Qp[qc - 2 * FQW], Qp[qc - FQW], Qp[qc + FQW],
                                          Qp[qc + 2 * FQW], Qu[qc - FQW], Qu[qc + FQW], P.z0, P.z1,
                                          P.delta);
            const double px = flatten_1d(Qp[cx - 2 * FQW], Qp[cx - FQW], Qp[cx + FQW],
                                         Qp[cx + 2 * FQW], Qu[cx - FQW], Qu[cx + FQW], P.z0, P.z1,
                                         P.delta);
            const double xiy = flatten_1d(Qp[qc - 2], Qp[qc - 1], Qp[qc + 1], Qp[qc + 2],
                                          Qv[qc - 1], Qv[qc + 1], P.z0, P.z1, P.delta);
            const double py = flatten_1d(Qp[cy - 2], Qp[cy - 1], Qp[cy + 1], Qp[cy + 2],
                                         Qv[cy - 1], Qv[cy + 1], P.z0, P.z1, P.delta);
            xi = fmin(fmin(xix, px), fmin(xiy, py));
        }
        double q0[4], dqx[4], dqy[4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const double *a = B0 + n * FQN;
            q0[n] = a[qc];
            dqx[n] = xi * limited_slope(a[qc - 2 * FQW], a[qc - FQW], a[qc], a[qc + FQW],
                                        a[qc + 2 * FQW], STD ? 2 : P.limiter);
            dqy[n] = xi * limited_slope(a[qc - 2], a[qc - 1], a[qc], a[qc + 1], a[qc + 2],
                                        STD ? 2 : P.limiter);
        }
        const double cs = psqrt(pdiv(gamma * q0[3], q0[0]));   // interface.py:122
        Trace lo, hi;
        trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], gamma,
                     pdiv(P.dt, GA.Lx(ic, jc)), lo, hi);
        {   // :216-224
            const double rs = -0.5 * P.dt * GA.dlAx(ic, jc) * q0[0] * q0[1];
            hi.r += rs; lo.r += rs;
            hi.p += rs * cs * cs; lo.p += rs * cs * cs;
        }
        XM = prim_to_cons(Prim{lo.r, lo.un, lo.ut, lo.p}, gamma);
        XP = prim_to_cons(Prim{hi.r, hi.un, hi.ut, hi.p}, gamma);
        trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], gamma,
                     pdiv(P.dt, GA.Ly(ic, jc)), lo, hi);
        {   // :226-234
            const double rs = -0.5 * P.dt * GA.dlAy(ic, jc) * q0[0] * q0[2];
            hi.r += rs; lo.r += rs;
            hi.p += rs * cs * cs; lo.p += rs * cs * cs;
        }
        YM = prim_to_cons(Prim{lo.r, lo.ut, lo.un, lo.p}, gamma);
        YP = prim_to_cons(Prim{hi.r, hi.ut, hi.un, hi.p}, gamma);
        // vertex divergence at (i-1/2, j-1/2), interface.py:331-364
        {
            const double rr = (i + 0.5 - g.ng) * P.dx + G.xmin;
            const double rl = (i - 0.5 - g.ng) * P.dx + G.xmin;
            const double rc = (i - g.ng) * P.dx + G.xmin;
            const double ur = 0.5 * (Qu[qc] + Qu[qc - 1]);
            const double ul = 0.5 * (Qu[qc - FQW] + Qu[qc - FQW - 1]);
            const double ux = pdiv(ur * rr * rr - ul * rl * rl, rc * rc * P.dx);
            const double sint = G.sint[jc], sinb = G.sinb[jc], sinc = G.sinc[jc];
            double vy = 0.0;
            if (sinc != 0.0) {
                const double vt = 0.5 * (Qv[qc] + Qv[qc - FQW]);
                const double vb = 0.5 * (Qv[qc - 1] + Qv[qc - FQW - 1]);
                vy = pdiv(sint * vt - sinb * vb, rc * sinc * P.dy);
            }
            D[t] = ux + vy;
        }
        // get_external_sources on the interior (simulation.py:117-124), ghost cells by the
        // boundary rules of the state's variables; apply_source_terms (unsplit_fluxes.py:308-326)
        {
            const int si = bc_src(P.mr, ic, g.ilo, g.ihi), sj = bc_src(P.mc, jc, g.jlo, g.jhi);
            const size_t ks = (size_t)si * p + sj;
            const unsigned sd = (ic < g.ilo ? 1u : 0u) | (ic > g.ihi ? 2u : 0u) | (jc < g.jlo ? 4u : 0u) |
                                (jc > g.jhi ? 8u : 0u);
            Cons Us{Uin[ks], Uin[pl + ks], Uin[2 * pl + ks], Uin[3 * pl + ks]};
            Us.d = fmax(Us.d, P.small_dens);
            double Sx = Us.d * P.grav;
            double SE = Us.mx * P.grav;
            Sx += pdiv(Us.my * Us.my, Us.d * GA.x(si, sj));
            double Sy = pdiv(-Us.mx * Us.my, Us.d);
            SE = odd_sides((P.odd >> 4) & sd) ? -SE : SE;
            Sx = odd_sides((P.odd >> 8) & sd) ? -Sx : Sx;
            Sy = odd_sides((P.odd >> 12) & sd) ? -Sy : Sy;
            const double sE = hdt * SE, sx = hdt * Sx, sy = hdt * Sy;
            XM.mx += sx; XM.my += sy; XM.E += sE;
            XP.mx += sx; XP.my += sy; XP.E += sE;
            YM.mx += sx; YM.my += sy; YM.E += sE;
            YP.mx += sx; YP.my += sy; YP.E += sE;
        }
        lds_put(S, t, XP);
        lds_put(S + 4 * FNT, t, YP);
    }
    __syncthreads();   // Q is dead from here on; B0 becomes the flux buffer

    // ---- phase 2: transverse Riemann problems on the lower faces --------
    Cons FxT{0, 0, 0, 0}, FyT{0, 0, 0, 0};
    double pxt = 0.0, pyt = 0.0;
    if (ti >= 1)
        FxT = sphf_face(lds_get(S, t - FBJ), XM, gamma, true, P.solid_xl && i == g.ilo, pxt);
    if (tj >= 1)
        FyT = sphf_face(lds_get(S + 4 * FNT, t - 1), YM, gamma, false, P.solid_yl && j == g.jlo, pyt);
    lds_put(B0, t, FxT);
    lds_put(B0 + 4 * FNT, t, FyT);
    PT[t] = pxt; PT[FNT + t] = pyt;
    __syncthreads();

    // ---- phase 3: transverse correction of the cell's own states.  A face's two states are
    // corrected with the volume / length of the cell ABOVE the face (unsplit_fluxes.py:444-481:
    // hdtV and Lx / Ly at (i, j) for the states of face (i, j)): the upper states of this cell
    // take the next cell's
    if (tj >= 1 && tj <= FBJ - 2) {
        const Cons Fhi = lds_get(B0 + 4 * FNT, t + 1);   // F_yT at (i, j+1)
        const double Ahi = GA.Ay(ic, jpc), Alo = GA.Ay(ic, jc);
        const double dpy = PT[FNT + t + 1] - pyt;
        XM = sphf_corrected(XM, Fhi, Ahi, FyT, Alo, pdiv(hdt, GA.V(ic, jc)));
        XM.my += pdiv(-hdt * dpy, GA.Ly(ic, jc));
        XP = sphf_corrected(XP, Fhi, Ahi, FyT, Alo, pdiv(hdt, GA.V(ipc, jc)));
        XP.my += pdiv(-hdt * dpy, GA.Ly(ipc, jc));
    }
    if (ti >= 1 && ti <= FBI - 2) {
        const Cons Fhi = lds_get(B0, t + FBJ);           // F_xT at (i+1, j)
        const double Ahi = GA.Ax(ipc, jc), Alo = GA.Ax(ic, jc);
        const double dpx = PT[t + FBJ] - pxt;
        YM = sphf_corrected(YM, Fhi, Ahi, FxT, Alo, pdiv(hdt, GA.V(ic, jc)));
        YM.mx += pdiv(-hdt * dpx, GA.Lx(ic, jc));
        YP = sphf_corrected(YP, Fhi, Ahi, FxT, Alo, pdiv(hdt, GA.V(ic, jpc)));
        YP.mx += pdiv(-hdt * dpx, GA.Lx(ic, jpc));
    }
    lds_put(S, t, XP);
    lds_put(S + 4 * FNT, t, YP);
    __syncthreads();

    // ---- phase 4: final Riemann problems + artificial viscosity ---------
    unsigned sdc;
    Cons Uc = load_cons_bc(Uin, g, P, ic, jc, sdc);
    const bool cell_interior = (i >= g.ilo && i <= g.ihi && j >= g.jlo && j <= g.jhi);
    if (cell_interior) Uc.d = fmax(Uc.d, P.small_dens);
    Cons Fx{0, 0, 0, 0}, Fy{0, 0, 0, 0};
    double px = 0.0, py = 0.0;
    const double d00 = D[t];
    Cons Umx = load_cons_bc(Uin, g, P, ic - 1, jc, sdc);
    Cons Umy = load_cons_bc(Uin, g, P, ic, jc - 1, sdc);
    double avx = 0.0, avy = 0.0;
    // interface.py:366-376: only faces i in [ilo, ihi], j in [jlo, jhi]
    if (ti >= 1 && tj >= 1 && tj <= FBJ - 2 && i >= g.ilo && i <= g.ihi && j >= g.jlo && j <= g.jhi) {
        const double divU_x = 0.5 * (d00 + D[t + 1]);
        avx = P.cvisc * fmax(-divU_x * GA.Lx(ic, jc), 0.0);
    }
    if (tj >= 1 && ti >= 1 && ti <= FBI - 2 && j >= g.jlo && j <= g.jhi && i >= g.ilo && i <= g.ihi) {
        const double divU_y = 0.5 * (d00 + D[t + FBJ]);
        avy = P.cvisc * fmax(-divU_y * GA.Ly(ic, jc), 0.0);
    }
    if (i - 1 >= g.ilo && i - 1 <= g.ihi && j >= g.jlo && j <= g.jhi)
        Umx.d = fmax(Umx.d, P.small_dens);
    if (i >= g.ilo && i <= g.ihi && j - 1 >= g.jlo && j - 1 <= g.jhi)
        Umy.d = fmax(Umy.d, P.small_dens);
    if (ti >= 1 && tj >= 1 && tj <= FBJ - 2) {           // x face (i, j)
        Fx = sphf_face(lds_get(S, t - FBJ), XM, gamma, true, P.solid_xl && i == g.ilo, px);
        Fx.d += avx * (Umx.d - Uc.d);
        Fx.E += avx * (Umx.E - Uc.E);
        Fx.mx += avx * (Umx.mx - Uc.mx);
        Fx.my += avx * (Umx.my - Uc.my);
    }
    if (tj >= 1 && ti >= 1 && ti <= FBI - 2) {           // y face (i, j)
        Fy = sphf_face(lds_get(S + 4 * FNT, t - 1), YM, gamma, false, P.solid_yl && j == g.jlo, py);
        Fy.d += avy * (Umy.d - Uc.d);
        Fy.E += avy * (Umy.E - Uc.E);
        Fy.mx += avy * (Umy.mx - Uc.mx);
        Fy.my += avy * (Umy.my - Uc.my);
    }
    lds_put(B0, t, Fx);
    lds_put(B0 + 4 * FNT, t, Fy);
    PT[t] = px; PT[FNT + t] = py;      // (the transverse pressures were last read before the barrier above)
    __syncthreads();

    // ---- phase 5: conservative update with the area / volume arrays, the pressure
    // gradients and the source predictor-corrector (simulation.py:375-423) + CFL -----------
    double cfl = INFINITY;
    if (ti >= 1 && ti <= FBI - 2 && tj >= 1 && tj <= FBJ - 2 && cell_interior) {
        const double dtdV = pdiv(P.dt, GA.V(ic, jc));
        const Cons Fxh = lds_get(B0, t + FBJ);
        const Cons Fyh = lds_get(B0 + 4 * FNT, t + 1);
        const double Ax0 = GA.Ax(ic, jc), Ax1 = GA.Ax(ipc, jc), Ay0 = GA.Ay(ic, jc), Ay1 = GA.Ay(ic, jpc);
        double Un[4];
        const double Uo[4] = {Uc.d, Uc.E, Uc.mx, Uc.my};
        Un[0] = Uo[0] + dtdV * (Fx.d * Ax0 - Fxh.d * Ax1 + Fy.d * Ay0 - Fyh.d * Ay1);
        Un[1] = Uo[1] + dtdV * (Fx.E * Ax0 - Fxh.E * Ax1 + Fy.E * Ay0 - Fyh.E * Ay1);
        Un[2] = Uo[2] + dtdV * (Fx.mx * Ax0 - Fxh.mx * Ax1 + Fy.mx * Ay0 - Fyh.mx * Ay1);
        Un[3] = Uo[3] + dtdV * (Fx.my * Ax0 - Fxh.my * Ax1 + Fy.my * Ay0 - Fyh.my * Ay1);
        Un[2] -= pdiv(P.dt * (PT[t + FBJ] - px), GA.Lx(ic, jc));
        Un[3] -= pdiv(P.dt * (PT[FNT + t + 1] - py), GA.Ly(ic, jc));
        // S_old = S(U_old); U += dt S_old; S_new (time-centred x-momentum); U += dt/2 (S_new - S_old)
        const double r = GA.x(ic, jc), grav = P.grav, dt = P.dt;
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
        Cons Uw;
        Uw.d = Un[0];   // the density source is zero
        Uw.E = Un[1] + 0.5 * dt * (SE_new - SE_old);
        Uw.mx = Un[2] + 0.5 * dt * (Sx_new - Sx_old);
        Uw.my = Un[3] + 0.5 * dt * (Sy_new - Sy_old);
        Uout[k] = Uw.d; Uout[pl + k] = Uw.E; Uout[2 * pl + k] = Uw.mx; Uout[3 * pl + k] = Uw.my;
        cfl = cfl_cell(Uw, gamma, GA.Lx(ic, jc), GA.Ly(ic, jc));
        cfl = sphf_ghost_cfl<FAC>(Uw, gamma, g, P, GA, i, j, cfl);
    }
    cfl = block_reduce_min(cfl);
    if (t == 0) partial[tile] = cfl;
}

// ghost frame of all 4 planes old -> new (the reference updates in place, so
// ghost cells keep their pre-step values)
// O(perimeter): blockIdx.y enumerates the 2*ng ghost rows (all j) followed by
// the nx interior rows (only their 2*ng ghost columns)
__global__ void k_copy_frame4(const double *__restrict__ src, double *__restrict__ dst, Geom g)
{
    // 1-d grid: the 2 ng full ghost rows in pieces of 256 columns, then the ghost columns of
    // the interior rows (a 2-d grid whose column part ran in block column 0 only launched
    // 33 thousand workgroups at 16384^2 for 1024 that had work)
    const int ng = g.ng;
    const int nxb = (g.qy + 255) / 256, nrowblk = 2 * ng * nxb;
    const int b = blockIdx.x;
    int i, j;
    if (b < nrowblk) {                              // a piece of a full ghost row
        const int rr = b / nxb;
        i = (rr < ng) ? rr : g.ihi + 1 + (rr - ng);
        j = (b - rr * nxb) * 256 + (int)threadIdx.x;
        if (j >= g.qy) return;
    } else {                                        // ghost columns of interior rows
        const int t = threadIdx.x;                  // 256 threads: 2*ng columns x rows
        const int rows_per_block = 256 / (2 * ng);
        const int r = (b - nrowblk) * rows_per_block + t / (2 * ng);
        const int kx = t % (2 * ng);
        if (r >= g.nx || t >= rows_per_block * 2 * ng) return;
        i = g.ilo + r;
        j = (kx < ng) ? kx : g.jhi + 1 + (kx - ng);
    }
    const size_t k = (size_t)i * g.pitch + j;
#pragma unroll
    for (int n = 0; n < 4; n++) dst[n * g.plane + k] = src[n * g.plane + k];
}

// second state buffer + the kernel parameters both single-launch kernels share
int fused_prepare(pyrohip_state *s, const pyrohip_comp_params *p, double dt, FP &P, double *&Uin,
                  double *&Uout, bool reset_flag, bool second_buffer)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    // (second_buffer = false: a launch that writes somewhere else -- the method-of-lines
    // right-hand side into the k state: no 4 planes allocated and zeroed on every stage state)
    if (!s->alt_base && second_buffer) {
        size_t n = g.plane * 4 + 16;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->alt_base, n * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(s->alt_base, 0, n * sizeof(double), c->stream));
    }
    Uin = s->d;
    Uout = s->alt_base ? s->alt_base + geom_lead(g) : nullptr;
    P.gamma = p->gamma; P.dx = p->dx; P.dy = p->dy; P.dt = dt;
    P.z0 = p->z0; P.z1 = p->z1; P.delta = p->delta; P.cvisc = p->cvisc;
    P.small_dens = p->small_dens;
    P.limiter = p->limiter; P.use_flattening = p->use_flattening;
    P.avx_hi = p->avisc_xhi_interior; P.avy_hi = p->avisc_yhi_interior;
    P.dtdx = dt / p->dx; P.dtdy = dt / p->dy;
    P.hdtV = (0.5 * dt) / (p->dx * p->dy);
    P.dtdV = dt / (p->dx * p->dy);
    P.gm1 = p->gamma - 1.0; P.rgm1 = 1.0 / (p->gamma - 1.0);
    P.rdx = 1.0 / p->dx; P.rdy = 1.0 / p->dy;
    P.grav = p->grav;
    P.refl_ylo = (s->bc[3 * 4 + 2] == PYROHIP_BC_REFLECT_ODD);
    P.refl_yhi = (s->bc[3 * 4 + 3] == PYROHIP_BC_REFLECT_ODD);
    P.amb_yhi = (s->bc[3 * 4 + 3] == PYROHIP_BC_AMBIENT);
    P.heat = s->heat; P.heat_rate = s->heat ? p->heat_rate : 0.0;
    P.have_src = (p->grav != 0.0 || s->heat != nullptr);
    P.solid_xl = p->solid_xl; P.solid_yl = p->solid_yl;
    P.ntj = P.ntiles = 0;
    P.L = P.ncb = P.nsb = P.nunits = 0;
    P.prio_duty = 0;
    P.sb_first = 0; P.sb_step = 1; P.n_extra = 0; P.n_short = 0; P.Ls = 0; P.n_tail = 0; P.units_short = 0; P.prio_board = nullptr; P.prio_tag = 0;
    P.mr = bc_map(g.ilo, g.ihi, g.ng, 0, 0, false);      // identity (tile kernel: fused_fill_maps)
    P.mc = P.mr;
    P.odd = 0;
    P.pol = nullptr; P.pol_m = 0; P.pol_pre = 0;
    P.rk_k = nullptr; P.rk_n = 0; P.rk_a[0] = P.rk_a[1] = P.rk_a[2] = 0.0;
    P.rk_final = 0; P.rk_nb = 0; P.rk_b[0] = P.rk_b[1] = P.rk_b[2] = P.rk_b[3] = 0.0; P.rk_out = nullptr;
    if (reset_flag) PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    return 0;
}

// after the step kernel: ghost frame old -> new, minimum of the per-workgroup
// CFL partials (all-reduced over the slabs when decomposed), positivity flag,
// swap of the two state buffers
void fused_copy_frame(pyrohip_state *s)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const int rows_per_block = 256 / (2 * g.ng);
    const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(k_copy_frame4, dim3(nblk), dim3(256), 0, c->stream,
                       (const double *)s->d, s->alt_base + geom_lead(g), g);
}

int fused_tail(pyrohip_state *s, double *part, int nparts, bool frame_copied, const double **dmin_out,
               bool defer)
{
    pyrohip_ctx *c = s->ctx;
    if (!frame_copied) fused_copy_frame(s);
    if (defer && !c->global_cfl && nparts <= 128 * kMinStageBlocks) {
        // device-side stepping: the next policy kernel (one workgroup anyway) takes the
        // minimum of the partials itself (up to 128 per thread: the row-marching kernel's
        // wavefronts up to 16384^2; two launches less per step, 14 us of a 0.72 ms step at
        // 4096^2) and leaves it where this function would have
        s->pend_part = part;
        s->pend_n = nparts;
        *dmin_out = part + nparts + kMinStageBlocks;
        return 0;
    }
    const double *dmin = launch_min_reduce(c->stream, part, nparts);
    s->cfl_is_global = false;
    if (c->global_cfl) {   // multi-GPU: the next dt needs the minimum over all slabs
        PYRO_TRY(comm_allreduce_min_device(c, const_cast<double *>(dmin)));
        s->cfl_is_global = true;
    }
    PYRO_CHECK_HIP(hipGetLastError());
    *dmin_out = dmin;
    return 0;
}

void fused_swap(pyrohip_state *s)
{
    double *old_base = s->base;
    s->base = s->alt_base;
    s->alt_base = old_base;
    s->d = s->base + geom_lead(s->g);
}

int fused_sync(pyrohip_state *s, const double *dmin)
{
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync((char *)c->reduce_host + 8, s->d_flag, sizeof(int),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    const int flagv = *(int *)((char *)c->reduce_host + 8);
    if (flagv & 1) {   // like the reference's assert: the state is left untouched
        s->next_cfl_min = -1.0;
        set_error("invalid state: min(rho) <= 0 or min(e) <= 0 on the interior "
                  "(compressible/simulation.py:68-71)");
        return PYROHIP_ERR_STATE;
    }
    fused_swap(s);
    s->next_cfl_min = ((double *)c->reduce_host)[0];
    s->cfl_kind = 0;
    return 0;
}

// S == nullptr: one step with the host's dt, minimum and flag read back, buffers
// swapped if the state was valid.  S != nullptr (pyrohip_comp_evolve): dt comes
// from *S on the device, nothing is read back, *dmin_out is the device address of
// the new CFL minimum
int comp_step_fused_ex(pyrohip_state *s, const pyrohip_comp_params *p, double dt,
                       const StepScalars *S, const double **dmin_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Uout;
    PYRO_TRY(fused_prepare(s, p, dt, P, Uin, Uout, S == nullptr));
    if (p->fuse_fill) {
        // the caller checked comp_can_fuse_fill(): outflow / reflect / periodic / halo sides
        // (the same kinds for all four variables; halo rows are data), no source terms
        P.mr = bc_map(g.ilo, g.ihi, g.ng, s->bc[0], s->bc[1], true);
        P.mc = bc_map(g.jlo, g.jhi, g.ng, s->bc[2], s->bc[3], true);
        for (int n = 0; n < 4; n++)
            for (int sd = 0; sd < 4; sd++)
                if (s->bc[n * 4 + sd] == PYROHIP_BC_REFLECT_ODD) P.odd |= 1u << (4 * n + sd);
    }
    const int nti = (g.nx + FTI - 1) / FTI;
    P.ntj = (g.ny + FTJ - 1) / FTJ;
    P.ntiles = nti * P.ntj;
    PYRO_TRY(c->reduce.ensure((P.ntiles + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    // instances: Riemann solver x (default reconstruction as constants | generic)
    using KernelT = void (*)(const double *, double *, Geom, FP, int *, double *,
                             const StepScalars *);
    static const KernelT kernels[3][2] = {
        {k_ctu_fused<0, false>, k_ctu_fused<0, true>},
        {k_ctu_fused<1, false>, k_ctu_fused<1, true>},
        {k_ctu_fused<2, false>, k_ctu_fused<2, true>}};
#ifndef PYRO_EMU
    static bool attr_set = false;
    if (!attr_set) {
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 2; b++)
                PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)kernels[a][b],
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)FLDS_BYTES));
        attr_set = true;
    }
#endif
    {
        const int solver = (p->riemann == 1 || p->riemann == 2) ? p->riemann : 0;
        const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
        const KernelT kern = kernels[solver][std_rec];
        PYRO_LAUNCH(c, "k_ctu_fused", kern, dim3(P.ntiles), dim3(FBJ, FBI), FLDS_BYTES,
                    (const double *)Uin, Uout, g, P, s->d_flag, part, S);
    }
    // slab of a decomposed run with the halo communicator: the new boundary rows go out
    // on the halo stream right after the kernel (which wrote the ghost frame itself) --
    // the same protocol as the row-marching kernel's, whatever kernel a rank picked
    const bool post = s->nb_set && comm_can_overlap(s);
    if (post) PYRO_TRY(comm_post_halo(s, Uout));
    const double *dmin;
    PYRO_TRY(fused_tail(s, part, P.ntiles, true, &dmin, S != nullptr));   // the kernel wrote the ghost frame
    if (S) { fused_swap(s); *dmin_out = dmin; s->halo_pending = post; return 0; }
    const int rc = fused_sync(s, dmin);
    s->halo_pending = post && rc == 0;
    return rc;
}

int comp_step_fused(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    return comp_step_fused_ex(s, p, dt, nullptr, nullptr);
}

// SphericalPolar grid, one step with the host's dt in ONE launch (k_ctu_fused_sph); the caller
// (pyrohip_comp_step) checked comp_can_fuse_sph(): CGF, outflow / reflect / periodic sides
// S == nullptr: one step with the host's dt; S != nullptr (pyrohip_comp_evolve): dt from *S on the
// device, nothing read back, *dmin_out = device address of the new CFL minimum (comp_step_fused_ex)
int comp_step_fused_sph_ex(pyrohip_state *s, const pyrohip_comp_params *p, double dt,
                           const StepScalars *S, const double **dmin_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Uout;
    PYRO_TRY(fused_prepare(s, p, dt, P, Uin, Uout, S == nullptr));
    // ghost cells (of the state and of the source terms) through the boundary rules
    P.mr = bc_map(g.ilo, g.ihi, g.ng, s->bc[0], s->bc[1], true);
    P.mc = bc_map(g.jlo, g.jhi, g.ng, s->bc[2], s->bc[3], true);
    for (int n = 0; n < 4; n++)
        for (int sd = 0; sd < 4; sd++)
            if (s->bc[n * 4 + sd] == PYROHIP_BC_REFLECT_ODD) P.odd |= 1u << (4 * n + sd);
    const SphGeom &h = *s->sph;
    const SphG G{h.Lx, h.Ly, h.Ax, h.Ay, h.V, h.dlAx, h.dlAy, h.x2d, h.sint, h.sinb, h.sinc, h.xmin,
                 h.rowf, h.colf, (int)h.qxp, (int)h.qyp};
    const int fac = (h.rowf && h.colf) ? 1 : 0;       // the caller handed the 1-d factors over
    const int nti = (g.nx + FTI - 1) / FTI;
    P.ntj = (g.ny + FTJ - 1) / FTJ;
    P.ntiles = nti * P.ntj;
    PYRO_TRY(c->reduce.ensure((P.ntiles + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    using KernelT = void (*)(const double *, double *, Geom, FP, SphG, int *, double *,
                             const StepScalars *);
    static const KernelT kernels[2][2] = {{k_ctu_fused_sph<false, false>, k_ctu_fused_sph<true, false>},
                                          {k_ctu_fused_sph<false, true>, k_ctu_fused_sph<true, true>}};
#ifndef PYRO_EMU
    static bool attr_set = false;
    if (!attr_set) {
        for (int b = 0; b < 4; b++)
            PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)kernels[b >> 1][b & 1],
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)FLDS_BYTES_SPH));
        attr_set = true;
    }
#endif
    const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
    PYRO_LAUNCH(c, "k_ctu_fused_sph", kernels[fac][std_rec], dim3(P.ntiles), dim3(FBJ, FBI), FLDS_BYTES_SPH,
                (const double *)Uin, Uout, g, P, G, s->d_flag, part, S);
    const double *dmin;
    PYRO_TRY(fused_tail(s, part, P.ntiles, true, &dmin, S != nullptr));   // the kernel wrote the ghost frame
    if (S) { fused_swap(s); *dmin_out = dmin; s->halo_pending = false; return 0; }
    // (the minimum is method_compute_timestep's: whole array, ghost cells of the NEW state as the
    // boundary rules will fill them included -- sphf_ghost_cfl; pyrohip_comp_dt takes it from here)
    const int rc = fused_sync(s, dmin);
    s->cfl_is_global = false;
    return rc;
}

int comp_step_fused_sph(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    return comp_step_fused_sph_ex(s, p, dt, nullptr, nullptr);
}

}  // namespace PYRO_NS
}  // namespace pyro
