// The compressible CTU step on a SphericalPolar grid (x = r, y = theta; pyro/mesh/patch.py:242-312)
// as ONE launch of autonomous row-marching wavefronts -- the design of comp_wave.hip (kernel_set 2)
// with the geometry terms of pyro/compressible (unsplit_fluxes.py:411-440, 444-494; interface.py:
// 106, 215-234, 331-376; simulation.py:117-124, 330-423; riemann.py:1092-1096, 1156-1171):
//
//   * one wavefront = 64 columns (lane = column j = theta index), the inner 56 updated; it walks
//     down a strip of rows (row i = r index); the x direction lives in a 5-row register window,
//     what a row hands to the next (face states, F_xT, F_x, the face pressures) in a per-lane LDS
//     stash (own-lane slots, no barrier); the y direction comes from the neighbouring lanes by
//     whole-wave DPP rotations;
//   * the geometry is rebuilt in registers from its 1-d factors (sph_common.h: SphAt<true>; the
//     row factors are wavefront-uniform: scalar loads) or read from the planes (SphAt<false>);
//   * the arithmetic is the tile kernel's (comp_fused.hip: k_ctu_fused_sph), expression by
//     expression -- tracing with dt / Lx, dt / Ly and the geometric source, radial gravity + the
//     geometric source terms on the four face states (a ghost cell takes the value of the cell its
//     boundary rule copies from, with the variable's sign), CGF interface states whose pressure
//     stays out of the area-weighted flux difference and enters as a gradient, the spherical
//     vertex divergence, transverse corrections / update / CFL with areas, volumes and lengths,
//     the source predictor-corrector -- so the bit-faithful build is BIT-IDENTICAL to it and to
//     the staged spherical set (tests/test_device_compressible.py, tests/test_fullsize_legs.py).
//
// The tile kernel recomputes a 4-cell apron around 14 x 30 interior cells (2369 lane-instructions
// per cell update, 81 k wavefronts at 2048^2: profiles/r05_sph2048_pmc.json); here the apron is 8
// of 64 columns and 8 warm-up rows per strip.  Ghost cells of the state are read from a FILLED
// frame (the caller's fill / k_fill_frame2 of a device-side run); the source terms and the CFL
// minimum of the new state's ghost cells go through the boundary rules' index maps like the tile
// kernel's.  Boundaries: outflow / reflect / periodic sides.
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
#include "sph_common.h"

namespace {

constexpr int SWOUT = 56;          // columns a wavefront updates (reach of a cell update: 4 columns)
// the limited slopes: stencil.h's (bit-faithful build) / the half slopes of fused_common.h
#if PYRO_FAST
#define SPHW_LIMIT2 half_limit2
#define SPHW_SLOPE half_slope_shared
#else
#define SPHW_LIMIT2 limit2
#define SPHW_SLOPE slope_shared
#endif
#if defined(PYRO_EMU)
#define SPHW_FENCE() do {} while (0)
#else
#define SPHW_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#if !defined(PYRO_EMU)
template <int CTRL> __device__ __forceinline__ double sphw_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lm1(double v) { return sphw_dpp<0x13C>(v); }   // wave_ror:1: lane l-1
__device__ __forceinline__ double lp1(double v) { return sphw_dpp<0x134>(v); }   // wave_rol:1: lane l+1
#else
__device__ __forceinline__ double lm1(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double lp1(double v) { return __shfl_down(v, 1, 64); }
#endif
__device__ __forceinline__ Cons lm1(const Cons &U) { return Cons{lm1(U.d), lm1(U.E), lm1(U.mx), lm1(U.my)}; }
__device__ __forceinline__ Cons lp1(const Cons &U) { return Cons{lp1(U.d), lp1(U.E), lp1(U.mx), lp1(U.my)}; }

// per-lane LDS stash (slot s of lane l at doubles s * 64 + l)
constexpr int SS_YM = 0, SS_YP = 4, SS_FXT = 8, SS_XP = 12, SS_XPC = 16, SS_FX = 20, SS_L2 = 24,
              SS_PXT = 32, SS_PX = 33, SS_CFL = 34,
              SS_CB = 35, SS_CC = 36, SS_CT = 37,     // FAC: the lane's column factors B, C, T (sph_common.h)
              // FAC, contracted build: |B|, |C|, 1 / T in SS_CB / SS_CC / SS_CT, 1 / |E| in SS_CRE; the
              // running maximum of (|u| + c) / Lx, (|v| + c) / Ly over the lane's new cells in SS_AMAX
              SS_CRE = 38, SS_AMAX = 39,
              SS_SLOTS = 40;     // (40 x 512 B x 8 wavefronts = 160 KB: the LDS of a CU exactly)
constexpr size_t SPHW_LDS_BYTES = (size_t)SS_SLOTS * 64 * sizeof(double);

__device__ __forceinline__ Cons sget(const double *st, int s)
{
    return Cons{st[s * 64], st[(s + 1) * 64], st[(s + 2) * 64], st[(s + 3) * 64]};
}
__device__ __forceinline__ void sput(double *st, int s, const Cons &U)
{
    st[s * 64] = U.d; st[(s + 1) * 64] = U.E; st[(s + 2) * 64] = U.mx; st[(s + 3) * 64] = U.my;
}

template <bool STD, bool FAC>
__global__ __launch_bounds__(64, 2) void k_sph_wave(const double *__restrict__ Uin, double *__restrict__ Uout,
                                                    Geom g, FP P, SphG G, int *__restrict__ flag,
                                                    double *__restrict__ partial,
                                                    const StepScalars *__restrict__ S)
{
    HIP_DYNAMIC_SHARED(double, lds)
    const int l = threadIdx.x;
    double *st = lds + l;
    // workgroup -> (column strip, row strip): XCD x takes the units [x per, (x + 1) per) (comp_wave.hip)
    const int per = (P.nunits + 7) / 8;
    const int unit = pyro_uniform(((int)blockIdx.x % 8) * per + (int)blockIdx.x / 8);
    if (unit >= P.nunits) return;
    const int nreg = P.ncb * P.nsb;      // (units behind: the extra strip of the column strips [0, n_extra): comp_wave.hip)
    const int cb = pyro_uniform(unit < nreg ? unit % P.ncb : unit - nreg);
    const int sb = pyro_uniform(unit < nreg ? unit / P.ncb : P.nsb);
    double dt = P.dt;
    if (S) {
        if (!S->active) {      // past tmax / after an invalid state: nothing happens
            if (l == 0) partial[sb * P.ncb + cb] = INFINITY;
            return;
        }
        dt = S->dt;
    }
    int i0 = g.ilo + sb * P.L;
    int i1 = (sb == P.nsb - 1) ? g.ihi + 1 : i0 + P.L;
    if (cb < P.n_extra) {      // nsb + 1 strips of equal length (to a row)
        i0 = g.ilo + (int)((long)sb * g.nx / (P.nsb + 1));
        i1 = g.ilo + (int)((long)(sb + 1) * g.nx / (P.nsb + 1));
    }
    const int j = g.jlo + cb * SWOUT - 4 + l;
    const int jc = (j < g.qy) ? j : g.qy - 1;              // ragged last strip: clamped, unused
    const int jpc = (jc + 1 < g.qy) ? jc + 1 : jc;
    const bool jin = (j >= g.jlo && j <= g.jhi);
    const bool jout = jin && l >= 4 && l <= 59;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const int limiter = STD ? 2 : P.limiter;
    const bool flat = STD || P.use_flattening;
    const double gamma = P.gamma;
    const SphAt<FAC> GA{G, p, P.dx};
    // FAC: the lane's column factors sit in the stash (read where used: an LDS read instead of a
    // load through the vector memory path in the middle of a row's work), those of column j + 1
    // come from the neighbouring lane; the row factors of rows i-1, i, i+1 are wavefront-uniform:
    // scalar loads where they are used
    // FF: contracted build on the factors -- every quotient by a geometry value is a product with
    // tabulated reciprocals (rows: 1 / Ly, 1 / (F G), 1 / x = dlogAx / 2; columns: 1 / |E|, 1 / T:
    // pyrohip_state_set_geometry), 1 / V = (1 / |E|) (1 / (F G)); the bit-faithful build divides
    constexpr bool FF = (PYRO_FAST != 0) && FAC;
    if (FAC) {
        if (FF) {
            st[SS_CB * 64] = fabs(GA.cf(0, jc)); st[SS_CC * 64] = fabs(GA.cf(1, jc));
            st[SS_CT * 64] = GA.cf(5, jc); st[SS_CRE * 64] = GA.cf(4, jc);
            st[SS_AMAX * 64] = 0.0;
        } else {
            st[SS_CB * 64] = GA.cf(0, jc); st[SS_CC * 64] = GA.cf(1, jc); st[SS_CT * 64] = GA.cf(3, jc);
        }
    }
    // row factor k (A D F G Ly dlogAx x) of row r: the row number is wavefront-uniform, and told so
    // the compiler reads the table with a scalar load (no vector memory traffic, no vector register)
    // (through the constant address space: the tables are never written by a kernel, but only a
    // pointer that says so lets the compiler use the scalar cache)
#if defined(PYRO_EMU)
    const double *const rowf_c = G.rowf;
#else
    typedef const __attribute__((address_space(4))) double *ConstTab;
    const ConstTab rowf_c = (ConstTab)(unsigned long long)G.rowf;
#endif
    // (measured and dropped: a sliding window of the three rows' factors in scalar registers, the
    // next row's requested an iteration ahead -- the kernel is out of scalar registers, the window
    // went to vector lanes (v_writelane / v_readlane 116 -> 180 per row) and the step took the same time)
    auto RF = [&](int kf, int r) { return rowf_c[pyro_uniform(r) * kSphRowStride + kf]; };
    const double cE = -2.0 * 3.14159265358979323846 / 3.0;     // E = (-2 pi / 3) B (mesh/patch.py: device_geometry)
    const double hdt = 0.5 * dt;
    const double dtdx = pdiv(dt, P.dx);                    // dt / Lx: Lx = dr everywhere (patch.py:262)
    const int sj = bc_src(P.mc, jc, g.jlo, g.jhi);         // column the source terms of a ghost column come from
    const unsigned sdj = (jc < g.jlo ? 4u : 0u) | (jc > g.jhi ? 8u : 0u);
    const double sint = G.sint[jc], sinb = G.sinb[jc], sinc = G.sinc[jc];

    // the rows of a strip as [scalar base of the strip's first row, per plane] + [32-bit byte offset]
    // (saddr form of the loads / stores: comp_wave.hip)
    const int rbase = (i0 - 8 > 0) ? i0 - 8 : 0;
    const char *const sbase_in = (const char *)(Uin + (size_t)rbase * p);
    char *const sbase_out = (char *)(Uout + (size_t)rbase * p);
    const unsigned pitch8 = (unsigned)p * 8u, lane8 = (unsigned)jc * 8u;
    const size_t plb = pl * sizeof(double);
    (void)sbase_in; (void)sbase_out; (void)plb; (void)pitch8; (void)lane8;
    auto loadU = [&](int row) {
        row = row < 0 ? 0 : (row > g.qx - 1 ? g.qx - 1 : row);
#if defined(PYRO_EMU)
        const size_t kk = (size_t)row * p + jc;
        return Cons{Uin[kk], Uin[pl + kk], Uin[2 * pl + kk], Uin[3 * pl + kk]};
#else
        const unsigned off = (unsigned)(row - rbase) * pitch8 + lane8;
        return Cons{*(const double *)(sbase_in + off), *(const double *)(sbase_in + plb + off),
                    *(const double *)(sbase_in + 2 * plb + off), *(const double *)(sbase_in + 3 * plb + off)};
#endif
    };
    auto row_in = [&](int r) { return r >= g.ilo && r <= g.ihi; };

    double wr[5] = {1, 1, 1, 1, 1}, wu[5] = {0, 0, 0, 0, 0};   // primitive window, rows k-4 .. k
    double wv[5] = {0, 0, 0, 0, 0}, wp[5] = {1, 1, 1, 1, 1};
    double fxa = 1.0, fxb = 1.0;                               // flatten_x of rows k-4, k-3
    Cons Ue{1.0, 1.0, 0.0, 0.0}, Uem = Ue;                     // old state, rows k-3 / k-4
    double Dp = 0.0, up = 0.0, vp = 0.0;                       // vertex div(U) of row k-4; u, v at (k-4, j-1)
    // SPHW_DELAY (contracted build on the GPU; comp_wave.hip has the measurement): the stores of a row's
    // update at the top of the next iteration -- behind the consumption of the arrived row, in front of
    // the next request.  The old state of row k-4 (update, viscosity of its row) is still read a second
    // time -- requested behind the delayed stores --: rebuilt from the primitives every step a conserved
    // state drifts (comp_wave.hip); only row k-3's (source terms on the face states, the other operand of
    // the x viscosity flux) is rebuilt
#if PYRO_FAST && !defined(PYRO_EMU) && !defined(PYRO_SPHW_NO_DELAY)
    constexpr bool SPHW_DELAY = true;
#else
    constexpr bool SPHW_DELAY = false;
#endif
    Cons Upre = loadU(i0 - 4), Urep = SPHW_DELAY ? Cons{1.0, 1.0, 0.0, 0.0} : loadU(i0 - 7);
    Cons Upend{0.0, 0.0, 0.0, 0.0};
    Cons Ucx{1.0, 1.0, 0.0, 0.0};          // SPHW_DELAY: the old state of row k-4, read a second time
    auto store_row = [&](const Cons &V, int row) {
#if defined(PYRO_EMU)
        const size_t ko = (size_t)row * p + j;
        Uout[ko] = V.d; Uout[pl + ko] = V.E; Uout[2 * pl + ko] = V.mx; Uout[3 * pl + ko] = V.my;
#else
        const unsigned offo = (unsigned)(row - rbase) * pitch8 + (unsigned)j * 8u;
        *(double *)(sbase_out + offo) = V.d; *(double *)(sbase_out + plb + offo) = V.E;
        *(double *)(sbase_out + 2 * plb + offo) = V.mx; *(double *)(sbase_out + 3 * plb + offo) = V.my;
#endif
    };
    bool bad = false;
    {
        const Cons one{1.0, 1.0, 0.0, 0.0};
        sput(st, SS_YM, one); sput(st, SS_YP, one); sput(st, SS_XP, one); sput(st, SS_XPC, one);
        sput(st, SS_FXT, one); sput(st, SS_FX, one); sput(st, SS_L2, one); sput(st, SS_L2 + 4, one);
        st[SS_PXT * 64] = 1.0; st[SS_PX * 64] = 1.0; st[SS_CFL * 64] = INFINITY;
    }
#if !defined(PYRO_EMU)
    // launches of up to two rounds of resident wavefronts: the two wavefronts of a SIMD take turns
    // at the priority, two rows each, so that the pair ends together (comp_wave.hip; here the
    // second one holds it five eighths of the time: 2048^2 0.325 / 0.314 / 0.307 / 0.315 / 0.326 ms
    // per step with 0 / 4 / 5 / 6 / 7 eighths)
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
        __hip_atomic_store(prio_mine, (P.prio_tag << 16) | (i1 + 3 - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prio_seen = __hip_atomic_load(prio_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
#else
    auto prio_publish = [](int) {};
#endif
    for (int k = i0 - 4; k <= i1 + 3; k++) {
#if !defined(PYRO_EMU)
        if (prio_fb) {
            const int seen = __builtin_amdgcn_readfirstlane(prio_seen);
            const int left = i1 + 3 - k;
            const int other_left = ((seen >> 16) == P.prio_tag) ? (seen & 0xffff) : left;
            if (left > other_left) __builtin_amdgcn_s_setprio(1);
            else if (left < other_left) __builtin_amdgcn_s_setprio(0);
        } else
        if (P.prio_duty > 0) {
            const int phase = ((k - i0) >> 1) & 7;
            if (wslot ? (phase < P.prio_duty) : (phase >= P.prio_duty)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
#pragma unroll
        for (int n = 0; n < 4; n++) { wr[n] = wr[n + 1]; wu[n] = wu[n + 1]; wv[n] = wv[n + 1]; wp[n] = wp[n + 1]; }
        if (!SPHW_DELAY) Uem = Ue;
        if (SPHW_DELAY) {
            // (window index 1 = row k-3 after the shift above; the floor is in the primitives)
            Ue = prim_to_cons(Prim{wr[1], wu[1], wv[1], wp[1]}, gamma);
        } else {
        Ue = Urep;
        if (row_in(k - 3) && jin) Ue.d = fmax(Ue.d, P.small_dens);      // clean_state
        Urep = loadU(k - 2);
        }
        // ---- row k arrives: primitives
        {
            Cons U = Upre;
            if (!SPHW_DELAY) { Upre = loadU(k + 1); prio_publish(k); }
            const bool interior = row_in(k) && jin;
            if (interior) U.d = fmax(U.d, P.small_dens);
            bool ok;
            const Prim q = cons_to_prim_nb(U, gamma, ok);
            if (interior && !ok) bad = true;
            wr[4] = q.r; wu[4] = q.u; wv[4] = q.v; wp[4] = q.p;
            if (SPHW_DELAY) {
#if !defined(PYRO_EMU)
                asm volatile("" : "+v"(wr[4]), "+v"(wu[4]), "+v"(wv[4]), "+v"(wp[4]));   // (the arrived row is consumed HERE)
#endif
                SPHW_FENCE();
                if (k - 1 >= i0 + 4 && jout) store_row(Upend, k - 5);      // the update iteration k-1 made
                Ucx = loadU(k - 4);                 // (in front of the next row's request: its wait leaves that one out)
                Upre = loadU(k + 1);
                prio_publish(k);
                SPHW_FENCE();
            }
        }
        // ---- flatten_x and limit2_x of row k-2 (window index 2)
        double fxn = 1.0, l2n[4] = {0, 0, 0, 0};
        if (k >= i0) {
            if (flat) fxn = flatten_1d(wp[0], wp[1], wp[3], wp[4], wu[1], wu[3], P.z0, P.z1, P.delta);
            if (limiter != 0) {
                l2n[0] = SPHW_LIMIT2(wr[1], wr[2], wr[3]);
                l2n[1] = SPHW_LIMIT2(wu[1], wu[2], wu[3]);
                l2n[2] = SPHW_LIMIT2(wv[1], wv[2], wv[3]);
                l2n[3] = SPHW_LIMIT2(wp[1], wp[2], wp[3]);
            }
        }
        double l2a[4], l2b[4];     // limit2_x of rows k-4 (slot k & 1) and k-3; row k-2 takes the older one's place
        {
            double *sa = st + (SS_L2 + 4 * (k & 1)) * 64, *sb2 = st + (SS_L2 + 4 * ((k + 1) & 1)) * 64;
#pragma unroll
            for (int n = 0; n < 4; n++) { l2a[n] = sa[n * 64]; l2b[n] = sb2[n * 64]; sa[n * 64] = l2n[n]; }
        }
        double Dn = 0.0, um = 0.0, vm = 0.0;
        if (k >= i0 + 2) {
            const int i = k - 3;                       // row c: slopes, states, x flux; row f = i - 1: y flux, update
            // SPHW_DELAY: the old state of row f as read a second time, with the density floor of clean_state
            auto old_f = [&]() {
                Cons U = Ucx;
                if (row_in(i - 1) && jin) U.d = fmax(U.d, P.small_dens);
                return U;
            };
            const int ipc = (i + 1 < g.qx) ? i + 1 : i;
            const bool xface = (k >= i0 + 3);          // row c has a lower x face in the strip
            const bool frow = (k >= i0 + 4);           // row f is updated by this strip
            // geometry of rows i-1 (w = -1), i (0), i+1 (+1) at column j (jp: j + 1)
            const double cB = FAC ? st[SS_CB * 64] : 0.0, cC = FAC ? st[SS_CC * 64] : 0.0;
            const double cBp = FAC ? lp1(cB) : 0.0, cCp = FAC ? lp1(cC) : 0.0;
            auto rowi = [&](int w) { return w < 0 ? i - 1 : (w == 0 ? i : ipc); };
            auto gAx = [&](int w) { return FAC ? fabs(RF(0, rowi(w)) * cB) : GA.Ax(rowi(w), jc); };
            auto gAy = [&](int w, bool jp) {
                return FAC ? fabs((jp ? cCp : cC) * RF(1, rowi(w))) : GA.Ay(rowi(w), jp ? jpc : jc);
            };
            auto gV = [&](int w, bool jp) {
                return FAC ? fabs(((cE * (jp ? cBp : cB)) * RF(2, rowi(w))) * RF(3, rowi(w))) : GA.V(rowi(w), jp ? jpc : jc);
            };
            auto gLy = [&](int w) { return FAC ? RF(4, rowi(w)) : GA.Ly(rowi(w), jc); };
            const double gLx = P.dx;                   // Lx = dr everywhere (patch.py:262; checked by the caller for FAC)
            // FF: a / V, a / Ly, a / Lx as products (|A B| = |A| |B| exactly: the stash holds |B|, |C|)
            const double cRE = FF ? st[SS_CRE * 64] : 0.0, cREp = FF ? lp1(cRE) : 0.0;
            auto overV = [&](double a, int w, bool jp) {
                return FF ? a * ((jp ? cREp : cRE) * RF(8, rowi(w))) : pdiv(a, gV(w, jp));
            };
            auto overLy = [&](double a, int w) { return FF ? a * RF(7, rowi(w)) : pdiv(a, gLy(w)); };
            auto overLx = [&](double a) { return FF ? a * P.rdx : pdiv(a, gLx); };
            const double q0[4] = {wr[1], wu[1], wv[1], wp[1]};
            const double qm[4] = {wr[0], wu[0], wv[0], wp[0]};
            const double qp[4] = {wr[2], wu[2], wv[2], wp[2]};
            double ym[4], yp[4], l2y[4] = {0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < 4; n++) {
                ym[n] = lm1(q0[n]);
                yp[n] = lp1(q0[n]);
                if (limiter != 0) l2y[n] = SPHW_LIMIT2(ym[n], q0[n], yp[n]);
            }
            um = ym[1]; vm = ym[2];
            double xi = 1.0;
            if (flat) {     // flatten_multid (reconstruction.py:167-183)
                const double fy = flatten_1d(lm1(ym[3]), ym[3], yp[3], lp1(yp[3]), ym[2], yp[2], P.z0, P.z1, P.delta);
                const double fym = lm1(fy), fyp = lp1(fy);
                const double px_ = (qp[3] - qm[3] > 0) ? fxa : fxn;
                const double py_ = (yp[3] - ym[3] > 0) ? fym : fyp;
                xi = fmin(fmin(fxb, px_), fmin(fy, py_));
            }
#if PYRO_FAST
            xi = xi + xi;      // (the contracted build's slopes are half slopes: fused_common.h)
#endif
            double dqx[4], dqy[4];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                dqx[n] = xi * SPHW_SLOPE(l2a[n], l2b[n], l2n[n], qm[n], q0[n], qp[n], limiter);
                const double l2m = (limiter == 2) ? lm1(l2y[n]) : 0.0;
                const double l2p = (limiter == 2) ? lp1(l2y[n]) : 0.0;
                dqy[n] = xi * SPHW_SLOPE(l2m, l2y[n], l2p, ym[n], q0[n], yp[n], limiter);
            }
            SPHW_FENCE();
            // ---- external sources on the face states (simulation.py:117-124, unsplit_fluxes.py:
            // 308-326): a ghost cell takes the value of the cell its boundary rule copies from,
            // with the variable's sign
            double sE, sx, sy;
            {
                const unsigned sd = sdj | (i < g.ilo ? 1u : 0u) | (i > g.ihi ? 2u : 0u);
                // (an interior cell is its own source: the old state of row i is in registers)
                double Ud = Ue.d, Umx = Ue.mx, Umy = Ue.my, xs = FAC ? RF(6, i) : GA.x(i, jc);
                double rxs = FF ? 0.5 * RF(5, i) : 0.0;            // 1 / x = dlogAx / 2 (exact)
                if (sd != 0) {
                    const int si = bc_src(P.mr, i, g.ilo, g.ihi);
                    const size_t ks = (size_t)si * p + sj;
                    Ud = Uin[ks]; Umx = Uin[2 * pl + ks]; Umy = Uin[3 * pl + ks];
                    xs = GA.x(si, sj);
                    if (FF) rxs = 0.5 * RF(5, si);
                }
                Ud = fmax(Ud, P.small_dens);
                double Sx = Ud * P.grav;
                double SE = Umx * P.grav;
                double Sy;
                if (FF) {
                    const double rUd = prcp(Ud);
                    Sx = fma(Umy * Umy, rUd * rxs, Sx);
                    Sy = -Umx * Umy * rUd;
                } else {
                Sx += pdiv(Umy * Umy, Ud * xs);
                Sy = pdiv(-Umx * Umy, Ud);
                }
                SE = odd_sides((P.odd >> 4) & sd) ? -SE : SE;
                Sx = odd_sides((P.odd >> 8) & sd) ? -Sx : Sx;
                Sy = odd_sides((P.odd >> 12) & sd) ? -Sy : Sy;
                sE = hdt * SE; sx = hdt * Sx; sy = hdt * Sy;
            }
            // ---- vertex divergence at (i-1/2, j-1/2), interface.py:331-364, and the artificial
            // viscosity coefficients of the faces (i, j) in x and (i-1, j) in y (:366-376)
            {
                const double rr = (i + 0.5 - g.ng) * P.dx + G.xmin;
                const double rl = (i - 0.5 - g.ng) * P.dx + G.xmin;
                const double rc = (i - g.ng) * P.dx + G.xmin;
                const double ur = 0.5 * (q0[1] + um);
                const double ul = 0.5 * (qm[1] + up);
                const double ux = pdiv(ur * rr * rr - ul * rl * rl, rc * rc * P.dx);
                double vy = 0.0;
                if (sinc != 0.0) {
                    const double vt = 0.5 * (q0[2] + qm[2]);
                    const double vb = 0.5 * (vm + vp);
                    vy = pdiv(sint * vt - sinb * vb, rc * sinc * P.dy);
                }
                Dn = ux + vy;
            }
            const double Dn_p = lp1(Dn);
            double avx = 0.0, avy = 0.0;
            if (row_in(i) && jin) {
                const double divU_x = 0.5 * (Dn + Dn_p);
                avx = P.cvisc * fmax(-divU_x * gLx, 0.0);
            }
            if (jin && row_in(i - 1)) {
                const double divU_y = 0.5 * (Dp + Dn);
                avy = P.cvisc * fmax(-divU_y * gLy(-1), 0.0);
            }
            SPHW_FENCE();
            // ---- x states of row c (interface.py:106, 216-224), transverse x flux on its lower face
            // c^2 (interface.py:122; the contracted build: without the root)
            const double cs2 = PYRO_FAST ? gamma * q0[3] * prcp(q0[0]) : 0.0;
            const double cs = PYRO_FAST ? 0.0 : psqrt(pdiv(gamma * q0[3], q0[0]));
            Trace lo, hi;
            trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], gamma, dtdx, lo, hi);
            {
                const double rs = -0.5 * dt * (FAC ? RF(5, i) : GA.dlAx(i, jc)) * q0[0] * q0[1];
                hi.r += rs; lo.r += rs;
                if (PYRO_FAST) { hi.p = fma(rs, cs2, hi.p); lo.p = fma(rs, cs2, lo.p); }
                else { hi.p += rs * cs * cs; lo.p += rs * cs * cs; }
            }
            Cons XMn = prim_to_cons(Prim{lo.r, lo.un, lo.ut, lo.p}, gamma);
            Cons XPn = prim_to_cons(Prim{hi.r, hi.un, hi.ut, hi.p}, gamma);
            XMn.mx += sx; XMn.my += sy; XMn.E += sE;
            XPn.mx += sx; XPn.my += sy; XPn.E += sE;
            Cons FxTn{0, 0, 0, 0};
            double pxtn = 0.0;
            if (xface) FxTn = sphf_face(sget(st, SS_XP), XMn, gamma, true, P.solid_xl && i == g.ilo, pxtn);
            SPHW_FENCE();
            // ---- row f: y states corrected with F_xT of rows f, f+1 (unsplit_fluxes.py:444-481: the
            // upper state takes the volume of the cell above its face), final y flux
            Cons Fy{0, 0, 0, 0}, Fyh{0, 0, 0, 0};
            double py = 0.0, pyh = 0.0;
            if (frow) {
                const Cons FxTp = sget(st, SS_FXT);
                const double Ahi = gAx(0), Alo = gAx(-1);
                const double dpx = pxtn - st[SS_PXT * 64];
                Cons YMc = sphf_corrected(sget(st, SS_YM), FxTn, Ahi, FxTp, Alo, overV(hdt, -1, false));
                YMc.mx += overLx(-hdt * dpx);
                Cons YPc = sphf_corrected(sget(st, SS_YP), FxTn, Ahi, FxTp, Alo, overV(hdt, -1, true));
                YPc.mx += overLx(-hdt * dpx);
                Fy = sphf_face(lm1(YPc), YMc, gamma, false, P.solid_yl && j == g.jlo, py);
                if (SPHW_DELAY) Uem = old_f();
                const Cons Umy = lm1(Uem);
                Fy.d += avy * (Umy.d - Uem.d);
                Fy.E += avy * (Umy.E - Uem.E);
                Fy.mx += avy * (Umy.mx - Uem.mx);
                Fy.my += avy * (Umy.my - Uem.my);
                Fyh = lp1(Fy);
                pyh = lp1(py);
            }
            if (xface) { sput(st, SS_FXT, FxTn); st[SS_PXT * 64] = pxtn; }
            SPHW_FENCE();
            // ---- y states of row c (interface.py:106, 226-234), transverse y flux on its lower face
            trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], gamma,
                         overLy(dt, 0), lo, hi);
            {
                // dlogAy = 1 / (tan(theta) r)   (FF: (1 / T) (1 / x))
                const double dla = FF ? st[SS_CT * 64] * (0.5 * RF(5, i))
                                      : (FAC ? pdiv(1.0, st[SS_CT * 64] * RF(6, i)) : GA.dlAy(i, jc));
                const double rs = -0.5 * dt * dla * q0[0] * q0[2];
                hi.r += rs; lo.r += rs;
                if (PYRO_FAST) { hi.p = fma(rs, cs2, hi.p); lo.p = fma(rs, cs2, lo.p); }
                else { hi.p += rs * cs * cs; lo.p += rs * cs * cs; }
            }
            Cons YMn = prim_to_cons(Prim{lo.r, lo.ut, lo.un, lo.p}, gamma);
            Cons YPn = prim_to_cons(Prim{hi.r, hi.ut, hi.un, hi.p}, gamma);
            YMn.mx += sx; YMn.my += sy; YMn.E += sE;
            YPn.mx += sx; YPn.my += sy; YPn.E += sE;
            sput(st, SS_YM, YMn);
            sput(st, SS_YP, YPn);
            double pyt = 0.0;
            const Cons FyT = sphf_face(lm1(YPn), YMn, gamma, false, P.solid_yl && j == g.jlo, pyt);
            SPHW_FENCE();
            // ---- transverse correction of the x states of row c, final x flux
            Cons XMc, XPc;
            {
                const Cons FyTh = lp1(FyT);            // F_yT at (i, j+1)
                const double Ahi = gAy(0, true), Alo = gAy(0, false);
                const double dpy = lp1(pyt) - pyt;
                XMc = sphf_corrected(XMn, FyTh, Ahi, FyT, Alo, overV(hdt, 0, false));
                XMc.my += overLy(-hdt * dpy, 0);
                XPc = sphf_corrected(XPn, FyTh, Ahi, FyT, Alo, overV(hdt, 1, false));
                XPc.my += overLy(-hdt * dpy, 1);
            }
            sput(st, SS_XP, XPn);
            Cons Fxn{0, 0, 0, 0};
            double pxn = 0.0;
            if (xface) {
                Fxn = sphf_face(sget(st, SS_XPC), XMc, gamma, true, P.solid_xl && i == g.ilo, pxn);
                if (SPHW_DELAY) Uem = old_f();
                Fxn.d += avx * (Uem.d - Ue.d);
                Fxn.E += avx * (Uem.E - Ue.E);
                Fxn.mx += avx * (Uem.mx - Ue.mx);
                Fxn.my += avx * (Uem.my - Ue.my);
            }
            sput(st, SS_XPC, XPc);
            SPHW_FENCE();
            // ---- conservative update of row f with the area / volume arrays, the pressure gradients
            // and the source predictor-corrector (simulation.py:375-423) + CFL of the new state
            if (frow && jout) {
                const int f = i - 1;
                const Cons Fxp = sget(st, SS_FX);
                const double pxp = st[SS_PX * 64];
                const double dtdV = overV(dt, -1, false);
                const double Ax0 = gAx(-1), Ax1 = gAx(0), Ay0 = gAy(-1, false), Ay1 = gAy(-1, true);
                double Un[4];
                if (SPHW_DELAY) Uem = old_f();
                const double Uo[4] = {Uem.d, Uem.E, Uem.mx, Uem.my};
                Un[0] = Uo[0] + dtdV * (Fxp.d * Ax0 - Fxn.d * Ax1 + Fy.d * Ay0 - Fyh.d * Ay1);
                Un[1] = Uo[1] + dtdV * (Fxp.E * Ax0 - Fxn.E * Ax1 + Fy.E * Ay0 - Fyh.E * Ay1);
                Un[2] = Uo[2] + dtdV * (Fxp.mx * Ax0 - Fxn.mx * Ax1 + Fy.mx * Ay0 - Fyh.mx * Ay1);
                Un[3] = Uo[3] + dtdV * (Fxp.my * Ax0 - Fxn.my * Ax1 + Fy.my * Ay0 - Fyh.my * Ay1);
                Un[2] -= overLx(dt * (pxn - pxp));
                Un[3] -= overLy(dt * (pyh - py), -1);
                // S_old = S(U_old); U += dt S_old; S_new (time-centred x-momentum); U += dt/2 (S_new - S_old)
                const double r = FAC ? RF(6, f) : GA.x(f, jc), grav = P.grav;
                const double rr_ = FF ? 0.5 * RF(5, f) : 0.0;                // 1 / r
                const double rUo = FF ? prcp(Uo[0]) : 0.0;
                const double Sx_g_old = Uo[0] * grav;
                const double SE_old = Uo[2] * grav;
                const double Sx_old = FF ? fma(Uo[3] * Uo[3], rUo * rr_, Sx_g_old)
                                         : Sx_g_old + pdiv(Uo[3] * Uo[3], Uo[0] * r);
                const double Sy_old = FF ? -Uo[2] * Uo[3] * rUo : pdiv(-Uo[2] * Uo[3], Uo[0]);
                Un[1] = Un[1] + dt * SE_old;
                Un[2] = Un[2] + dt * Sx_old;
                Un[3] = Un[3] + dt * Sy_old;
                const double Sx_g_new = Un[0] * grav;
                const double xmom_new = Un[2] + 0.5 * dt * (Sx_g_new - Sx_g_old);
                const double SE_new = xmom_new * grav;
                const double rUn = FF ? prcp(Un[0]) : 0.0;
                const double Sx_new = FF ? fma(Un[3] * Un[3], rUn * rr_, Sx_g_new)
                                         : Sx_g_new + pdiv(Un[3] * Un[3], Un[0] * r);
                const double Sy_new = FF ? -Un[2] * Un[3] * rUn : pdiv(-Un[2] * Un[3], Un[0]);
                Cons Uw;
                Uw.d = Un[0];   // the density source is zero
                Uw.E = Un[1] + 0.5 * dt * (SE_new - SE_old);
                Uw.mx = Un[2] + 0.5 * dt * (Sx_new - Sx_old);
                Uw.my = Un[3] + 0.5 * dt * (Sy_new - Sy_old);
                if (SPHW_DELAY) Upend = Uw;
                else store_row(Uw, f);
                if (FF) {
                    // running maximum of (|u| + c) / Lx, (|v| + c) / Ly: one reciprocal at the end
                    // (a ghost cell's lengths are its own: sphf_ghost_cfl keeps the quotient form)
                    double ax, ay;
                    cfl_speeds(Uw, gamma, ax, ay);
                    st[SS_AMAX * 64] = fmax(st[SS_AMAX * 64], fmax(ax * P.rdx, ay * RF(7, f)));
                    st[SS_CFL * 64] = sphf_ghost_cfl<FAC>(Uw, gamma, g, P, GA, f, j, st[SS_CFL * 64]);
                } else {
                double cfl = cfl_cell(Uw, gamma, gLx, gLy(-1));
                cfl = sphf_ghost_cfl<FAC>(Uw, gamma, g, P, GA, f, j, cfl);
                st[SS_CFL * 64] = fmin(st[SS_CFL * 64], cfl);
                }
            }
            if (xface) { sput(st, SS_FX, Fxn); st[SS_PX * 64] = pxn; }
        }
        // hand the rows on
        fxa = fxb; fxb = fxn;
        Dp = Dn; up = um; vp = vm;
    }
    if (SPHW_DELAY && i1 + 3 >= i0 + 4 && jout) store_row(Upend, i1 - 1);      // the update the last iteration made
    if (bad) atomicOr(flag, 1);
    double cflw = st[SS_CFL * 64];
    if (FF) { const double am = st[SS_AMAX * 64]; if (am > 0.0) cflw = fmin(cflw, prcp(am)); }
    const double wm = wave_reduce_min(cflw);
    if (l == 0) partial[sb * P.ncb + cb] = wm;
}

// rows per strip: whole rounds of the resident wavefronts (two per SIMD) + one strip time for the
// stragglers of the last round (comp_wave.hip: wave_rows); a strip costs L + 8 iterations
int sphw_rows(int nx, int ncb, int slots)
{
    if (nx <= 32) return nx;
    long best_cost = -1;
    int best = 32;
    for (int L = 32; L <= 160 && L <= nx; L++) {
        int nsb = (nx + L - 1) / L;
        if (nsb > 1 && nx - (nsb - 1) * L < 4) nsb--;
        const int Leff = (nx + nsb - 1) / nsb;
        const long waves = (long)ncb * nsb;
        const long rounds = (waves + slots - 1) / slots;
        const long cost = (waves <= slots ? 1 : rounds + 1) * (Leff + 8);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = L; }
    }
    return best;
}

}  // namespace

// SphericalPolar grid, one step in ONE launch of the row-marching kernel.  The caller checked what
// the tile kernel needs too (comp_api.hip: comp_can_fuse_sph: CGF, outflow / reflect / periodic
// sides) and has FILLED the state's ghost cells.  S == nullptr: one step with the host's dt;
// S != nullptr (pyrohip_comp_evolve): dt from *S, *dmin_out = device address of the CFL minimum.
int comp_step_wave_sph_ex(pyrohip_state *s, const pyrohip_comp_params *p, double dt,
                          const StepScalars *S, const double **dmin_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Uout;
    PYRO_TRY(fused_prepare(s, p, dt, P, Uin, Uout, S == nullptr));
    // (the source terms of ghost cells and the CFL minimum over the new state's ghost cells go
    // through the boundary rules)
    P.mr = bc_map(g.ilo, g.ihi, g.ng, s->bc[0], s->bc[1], true);
    P.mc = bc_map(g.jlo, g.jhi, g.ng, s->bc[2], s->bc[3], true);
    for (int n = 0; n < 4; n++)
        for (int sd = 0; sd < 4; sd++)
            if (s->bc[n * 4 + sd] == PYROHIP_BC_REFLECT_ODD) P.odd |= 1u << (4 * n + sd);
    const SphGeom &h = *s->sph;
    const SphG G{h.Lx, h.Ly, h.Ax, h.Ay, h.V, h.dlAx, h.dlAy, h.x2d, h.sint, h.sinb, h.sinc, h.xmin,
                 h.rowf, h.colf, (int)h.qxp, (int)h.qyp};
    const int fac = (h.rowf && h.colf) ? 1 : 0;       // the caller handed the 1-d factors over
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    P.ncb = (g.ny + SWOUT - 1) / SWOUT;
    P.L = sphw_rows(g.nx, P.ncb, 8 * cus);
    if (p->march_rows > 0) P.L = p->march_rows < g.nx ? p->march_rows : g.nx;
    P.nsb = (g.nx + P.L - 1) / P.L;
    if (P.nsb > 1 && g.nx - (P.nsb - 1) * P.L < g.ng) P.nsb--;
    // (one round: as many strips as wavefront slots; the pairs of the SIMDs told their rows left: comp_wave.hip)
#if !defined(PYRO_SPHW_NO_EXTRA)
    P.n_extra = p->march_rows > 0 ? 0 : wave_fill_extra(P.ncb, P.nsb, g.nx, 8 * cus);
#endif
    P.nunits = P.ncb * P.nsb + P.n_extra;
    P.prio_duty = (P.nunits <= 2 * 8 * cus) ? 5 : 0;
#if !defined(PYRO_EMU) && !defined(PYRO_SPHW_NO_FEEDBACK)
    if (P.prio_duty > 0) PYRO_TRY(prio_board_acquire(c, &P.prio_board, &P.prio_tag));
#endif
    PYRO_TRY(c->reduce.ensure((P.nunits + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    using KernelT = void (*)(const double *, double *, Geom, FP, SphG, int *, double *, const StepScalars *);
    static const KernelT kernels[2][2] = {{k_sph_wave<false, false>, k_sph_wave<true, false>},
                                          {k_sph_wave<false, true>, k_sph_wave<true, true>}};
    const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
    PYRO_LAUNCH(c, "k_sph_wave", kernels[fac][std_rec], dim3(8 * ((P.nunits + 7) / 8)), dim3(64), SPHW_LDS_BYTES,
                (const double *)Uin, Uout, g, P, G, s->d_flag, part, S);
    PYRO_CHECK_HIP(hipGetLastError());
    // the ghost frame of the new buffer: the old (filled) ghost cells, unless the fill before this
    // step of a device-side run has written both frames (comp_api.hip: k_fill_frame2)
    const bool frame_done = s->frame_prefilled;
    s->frame_prefilled = false;
    const double *dmin;
    PYRO_TRY(fused_tail(s, part, P.nunits, frame_done, &dmin, S != nullptr));
    if (S) { fused_swap(s); *dmin_out = dmin; s->halo_pending = false; return 0; }
    // (the minimum is method_compute_timestep's: whole array, the new state's ghost cells as the
    // boundary rules will fill them included -- sphf_ghost_cfl)
    const int rc = fused_sync(s, dmin);
    s->cfl_is_global = false;
    return rc;
}

int comp_step_wave_sph(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    return comp_step_wave_sph_ex(s, p, dt, nullptr, nullptr);
}

}  // namespace PYRO_NS
}  // namespace pyro
