// Definitions shared by the two single-launch CTU kernels: the 2-d LDS tile
// kernel (comp_fused.hip, kernel_set 1) and the row-marching kernel
// (comp_wave.hip, kernel_set 2).  Included inside namespace pyro::PYRO_NS.
#pragma once

struct FP {   // kernel parameters
    double gamma, dx, dy, dt;
    double z0, z1, delta, cvisc, small_dens;
    int limiter, use_flattening;
    int avx_hi, avy_hi;
    int ntj, ntiles;
    // uniform quotients, evaluated once on the host with the reference's
    // expressions (IEEE double on both sides: same bits)
    double dtdx, dtdy;    // dt/dx, dt/dy          interface.py:106
    double hdtV;          // (0.5*dt)/(dx*dy)      unsplit_fluxes.py:444-445
    double dtdV;          // dt/(dx*dy)            simulation.py:375
    double gm1, rgm1;     // gamma - 1 and its reciprocal (the fast build multiplies by it)
    double rdx, rdy;      // 1 / dx, 1 / dy (fast build)
    double grav;          // compressible.grav (0: no source terms)
    int refl_ylo, refl_yhi;   // y-momentum reflects oddly at the lower / upper y wall
    int amb_yhi;              // "ambient" boundary on the upper y side
    int have_src;             // gravity and / or a heating source
    double heat_rate;         // S[E] += rho * heat_rate * heat[i,j] (ghost-filled plane)
    const double *heat;
    int solid_xl, solid_yl;   // CGF wall rule (riemann.py:274-286)
    int L, ncb, nsb;          // row-marching kernel: rows per strip, column blocks, strips
    int nunits;               // ... (column block, strip) pairs of this launch
    int prio_duty;            // ... eighths of the time the second wavefront of a SIMD has priority (0: age decides)
    int sb_first, sb_step;    // ... strip of workgroup b: sb_first + (b / ncb) * sb_step
    int n_extra;              // ... column strips [0, n_extra) are cut into nsb + 1 row strips (one-round launches: every slot filled)
    int n_short, Ls;          // ... many-round launches: the LAST n_short of the nsb row strips are short ones (Ls rows), dealt to the
                              //     ends of the eight XCD queues: the slots drain over a short strip's life (comp_wave.hip)
    int n_tail, units_short;  // ... long strips BEHIND the short ones (a slab's last boundary strip: 1); short units of THIS launch (0: plain dealing)
    int *prio_board;          // ... rows-left board of the SIMD pairs (nullptr: priority turns by prio_duty) and this launch's tag
    int prio_tag;
    // tile kernel: the ghost fill folded into the loads (pyrohip_comp_params.fuse_fill):
    // row / column maps of the boundary rules (identity without) and, per variable and
    // side, whether the ghost value changes sign (bit 4 n + side)
    BcMap mr, mc;
    unsigned odd;
    // row-marching kernel as the only launch of a step (pyrohip_comp_evolve, k_ctu_wave<.., ONE>):
    // ghost cells are READ through mr / mc / odd (no filled frame), and every wavefront derives
    // this step's dt from the previous launch's CFL minima itself (common.h: StepPolicy)
    StepPolicy *pol;          // (device memory)
    int pol_m;                // step of the call this launch is
    int pol_pre;              // S[pol_m & 1] is this step's already (k_dt_policy ran: step 0)
    // method-of-lines instance with the Runge-Kutta stage folded in (k_ctu_wave<.., MOL, .., RKF>):
    // the stage state y_s = y_0 + dt sum_j a_sj k_j (mesh/integration.py:105-118) is built per row
    // as it enters the window -- interior cells; ghost cells are the images the boundary rules
    // give (mr / mc / odd), as the fill of the stage state would have left them -- and the last
    // stage stores y_0 + dt sum_s b_s k_s (:120-129) into the other state buffer instead of its k
    const double *rk_k;       // the k state's first plane (slot j: rk_k + 4 j plane)
    int rk_n;                 // increments that enter this stage's start (columns j < rk_n of a_s)
    double rk_a[3];           // a[s][j] (terms with a zero weight are skipped)
    int rk_final;             // this launch is the last stage: the final update
    int rk_nb;                // ... stages of the method
    double rk_b[4];           // ... b[s]
    double *rk_out;           // ... the buffer the new state goes to
};

// parity of a 4-bit set of sides
__device__ __forceinline__ bool odd_sides(unsigned m)
{
    m &= 15u;
    m ^= m >> 2;
    m ^= m >> 1;
    return (m & 1u) != 0;
}

// host side, defined in comp_fused.hip
__global__ void k_copy_frame4(const double *__restrict__ src, double *__restrict__ dst, Geom g);
int fused_prepare(pyrohip_state *s, const pyrohip_comp_params *p, double dt, FP &P, double *&Uin,
                  double *&Uout, bool reset_flag = true, bool second_buffer = true);
// after the step kernel(s): ghost frame, minimum of the CFL partials (all-reduced over
// the slabs when decomposed) -> device address of the minimum
int fused_tail(pyrohip_state *s, double *part, int nparts, bool frame_copied, const double **dmin,
               bool defer = false);
// single-step API: read the minimum and the positivity flag back, swap the buffers
int fused_sync(pyrohip_state *s, const double *dmin);
// device-side run: swap the buffers without looking (the kernels freeze the state
// themselves once the flag is up)
void fused_swap(pyrohip_state *s);
void fused_copy_frame(pyrohip_state *s);

__device__ __forceinline__ ConsN to_nf(const Cons &U, bool x)
{
    return x ? ConsN{U.d, U.E, U.mx, U.my} : ConsN{U.d, U.E, U.my, U.mx};
}
__device__ __forceinline__ Cons from_nf(const ConsN &F, bool x)
{
    return x ? Cons{F.d, F.E, F.mn, F.mt} : Cons{F.d, F.E, F.mt, F.mn};
}
__device__ __forceinline__ Cons corr(const Cons &U, const Cons &Fhi, const Cons &Flo, double hdtV,
                                     double A)
{
    Cons r;   // U += -hdtV*(F_hi*A - F_lo*A), unsplit_fluxes.py:447-471
#if PYRO_FAST
    const double k = -hdtV * A;     // fast build: one product, one fma per component
    r.d = fma(k, Fhi.d - Flo.d, U.d);
    r.E = fma(k, Fhi.E - Flo.E, U.E);
    r.mx = fma(k, Fhi.mx - Flo.mx, U.mx);
    r.my = fma(k, Fhi.my - Flo.my, U.my);
#else
    r.d = U.d + (-hdtV * (Fhi.d * A - Flo.d * A));
    r.E = U.E + (-hdtV * (Fhi.E * A - Flo.E * A));
    r.mx = U.mx + (-hdtV * (Fhi.mx * A - Flo.mx * A));
    r.my = U.my + (-hdtV * (Fhi.my * A - Flo.my * A));
#endif
    return r;
}

// the same with k = -hdtV * A given (fast build of the row-marching kernel)
__device__ __forceinline__ Cons corr_k(const Cons &U, const Cons &Fhi, const Cons &Flo, double k)
{
    return Cons{fma(k, Fhi.d - Flo.d, U.d), fma(k, Fhi.E - Flo.E, U.E),
                fma(k, Fhi.mx - Flo.mx, U.mx), fma(k, Fhi.my - Flo.my, U.my)};
}

// limited slope from the limit2 values of the two neighbours (limiter 2), the
// cell's own limit2 (limiter 1) or none: the expressions of limited_slope()
// (stencil.h, reconstruction.py:9-120) with the shared limit2 passed in
__device__ __forceinline__ double slope_shared(double l2m, double l20, double l2p, double am1,
                                               double a0, double ap1, int limiter)
{
    if (limiter == 0) return 0.5 * (ap1 - am1);
    if (limiter == 1) return l20;
    const double dc = (2. / 3.) * (ap1 - am1 - 0.25 * (l2p + l2m));
    const double dl = ap1 - a0;
    const double dr = a0 - am1;
    return mc_select_l4(dc, dl, dr);
}

#if PYRO_FAST
// Contracted build, row-marching kernel (round 6): the MC-limited slopes as HALF slopes in signed
// min / max form -- no sign copy, no product of the one-sided differences, no compare + selects.
// With lo = min(dl, dr), hi = max(dl, dr), a = max(lo, 0), b = min(hi, 0):
//     limit2 / 2 = max(min((ap - am) / 4, a), b)
// (both differences positive: b = 0 and min(dc / 2, lo) > 0; both negative: a = 0, min(dc / 2, 0) =
// dc / 2 and max(dc / 2, hi); of opposite signs or one of them zero: a = b = 0 and the result is 0),
// and the same with the fourth-order centred slope of limit4 in place of (ap - am) / 4 -- stencil.h
// (mc_select_l4) shows that where dl dr > 0 it shares their sign.  Scaling by the power of two is
// exact: 2 x the result equals limit2 / limit4 of stencil.h up to the rounding of the centred
// difference's factors ((2/3) x vs 2 (1/3) x).  12 + 7 -> 10 + 5 instructions per variable and direction.
__device__ __forceinline__ void half_clip(double dl, double dr, double &a, double &b)
{
    a = fmax(fmin(dl, dr), 0.0);
    b = fmin(fmax(dl, dr), 0.0);
}
__device__ __forceinline__ double half_limit2(double am, double a0, double ap)
{
    double a, b;
    half_clip(ap - a0, a0 - am, a, b);
    return fmax(fmin(0.25 * (ap - am), a), b);
}
// half of slope_shared(): h2m / h20 / h2p = half_limit2 of the lower neighbour, the cell, the upper one
__device__ __forceinline__ double half_slope_shared(double h2m, double h20, double h2p, double am1,
                                                    double a0, double ap1, int limiter)
{
    if (limiter == 0) return 0.25 * (ap1 - am1);
    if (limiter == 1) return h20;
    double a, b;
    half_clip(ap1 - a0, a0 - am1, a, b);
    const double hc = (1. / 3.) * (ap1 - am1 - 0.5 * (h2p + h2m));
    return fmax(fmin(hc, a), b);
}
#endif

// cons_to_prim (hydro.h) without the branch on rho != 0: same operations on
// the same operands when rho != 0 (bit-identical), zeros otherwise
__device__ __forceinline__ Prim cons_to_prim_nb(const Cons &U, double gamma, bool &ok)
{
#if PYRO_FAST
    // fast build: no guard for rho == 0 (a state the reference's assert rejects anyway:
    // the quotients become NaN and `ok` comes out false), pressure without the detour
    // through the specific internal energy
    {
        const double rd = prcp(U.d);
        Prim q;
        q.r = U.d;
        q.u = U.mx * rd;
        q.v = U.my * rd;
        q.p = fma(-0.5, fma(U.my, q.v, U.mx * q.u), U.E) * (gamma - 1.0);
        ok = (q.p > 0.0) && (U.d > 0.0);
        return q;
    }
#else
    const bool nz = (U.d != 0.0);
    const double ds = nz ? U.d : 1.0;
    const double rd = PYRO_FAST ? prcp(ds) : 0.0;
    const double u = pvel(U.mx, ds, rd);
    const double v = pvel(U.my, ds, rd);
    const double e = pdivr(U.E - 0.5 * U.d * (u * u + v * v), ds, rd);
    Prim q;
    q.r = U.d;
    q.u = nz ? u : 0.0;
    q.v = nz ? v : 0.0;
    const double es = nz ? e : 0.0;
    q.p = U.d * es * (gamma - 1.0);
    ok = (es > 0.0) && (U.d > 0.0);
    return q;
#endif
}

