// Compressible CTU + Riemann step as ONE kernel per time step: autonomous
// wavefronts marching along the rows (kernel_set 2).
//
// Same arithmetic as the other kernel sets (per-cell functions of hydro.h /
// stencil.h in the reference's operation order).  One wavefront = 64 columns
// (lane = column j, 512-B row loads per plane), of which the inner 56 are
// updated; it walks down a strip of L rows on its own:
//
//   * x direction (rows, index i): rolling windows in REGISTERS.  limit2_x,
//     flatten_x, the x face states, FxT and Fx of a row are computed once and
//     handed from one iteration to the next.
//   * y direction (neighbouring lanes): ds_bpermute lane shuffles.  No LDS
//     memory, no workgroup barrier -- a value is read from the neighbour lane's
//     register at the point of use, so nothing has to be published a pipeline
//     stage ahead and a row goes through ALL stages of the algorithm in two
//     consecutive iterations.
//
// Iteration k of the strip [i0, i1):
//   S0  load row k -> primitives (window rows k-4..k); flatten_x, limit2_x of
//       row k-2
//   S2  row c = k-3: limit2_y / flatten_y (+ neighbours'), xi, limited slopes,
//       characteristic tracing -> XM XP YM YP; FxT(c) (XP of row c-1 carried),
//       FyT(c) (YP of lane j-1); transverse correction of the x states (FyT of
//       lane j+1); final x flux Fx(c) + artificial viscosity
//   S4  row c-1 = k-4: transverse correction of the y states (FxT of rows c-1, c),
//       final y flux Fy (YP of lane j-1) + artificial viscosity, conservative
//       update (Fy of lane j+1), store, CFL
//
// Lanes 0-3 and 60-63 are apron (87.5 % of the lanes produce output; a strip
// costs L + 8 iterations for L rows), which buys: no barriers, no LDS, the
// shortest possible carried state (70 doubles per lane).  HBM traffic: every
// row is read once per column strip (x 64/56) and written once; the second
// read of a row three iterations later (old state for the update and the
// viscosity terms) is an L1/L2 hit.
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


constexpr int WOUT = 56;          // columns a wavefront updates
// the limited slopes: stencil.h's (bit-faithful build) / the half slopes of fused_common.h
#if PYRO_FAST
#define LIMIT2 half_limit2
#define SLOPE_SHARED half_slope_shared
#else
#define LIMIT2 limit2
#define SLOPE_SHARED slope_shared
#endif
// stage boundary: the scheduler may not move instructions across it.  The
// stages are written in the order that keeps the live ranges short (a value is
// produced right before the Riemann problem that consumes it); left alone, the
// scheduler interleaves the stages for ILP and pushes the kernel over 256 VGPRs.
#if defined(PYRO_EMU) || defined(PYRO_WAVE_NO_SCHED_BARRIER)
#define STAGE_FENCE() do {} while (0)
#else
#define STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef PYRO_WAVE_PRIO_SHIFT
#define PYRO_WAVE_PRIO_SHIFT 1    // the priority changes hands in units of 2 rows
#endif
#ifndef PYRO_WAVE_MINW
#define PYRO_WAVE_MINW 2          // waves per SIMD the register allocation must allow
#endif

// value of the same variable in lane l-1 / l+1.  The ends of the wavefront are
// apron lanes whose results are never stored, but what they compute matters
// for speed: a wavefront executes every lazily evaluated branch (flattening,
// the shock branches of the wave-speed estimate) that ANY lane takes.  The DPP
// moves therefore ROTATE (lane 0 reads lane 63: a finite physical state, equal
// to its own in uniform regions) -- with shifts + bound_ctrl the end lanes read
// 0, went to NaN and dragged the wavefront through the slow paths (+8 v_rsq /
// v_rcp per row, PMC); shifts that keep the own value need a register copy in
// front of every move (the `old` operand is tied to the destination).  The
// shuffle variant (emulator, -DPYRO_WAVE_BPERMUTE) keeps the own value.  gfx950 keeps the GFX9 whole-wave DPP shifts (wave_shr:1 /
// wave_shl:1 move data across all 64 lanes, tools/dpp_probe.hip): two
// v_mov_b32_dpp per double, a register-to-register VALU move without the
// LDS-crossbar round trip of ds_bpermute (__shfl_up / __shfl_down).
#if !defined(PYRO_EMU) && !defined(PYRO_WAVE_BPERMUTE)
template <int CTRL> __device__ __forceinline__ double lane_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_m1(double v) { return lane_dpp<0x13C>(v); }   // wave_ror:1
__device__ __forceinline__ double lane_p1(double v) { return lane_dpp<0x134>(v); }   // wave_rol:1
__device__ __forceinline__ double lane_m2(double v) { return lane_m1(lane_m1(v)); }
__device__ __forceinline__ double lane_p2(double v) { return lane_p1(lane_p1(v)); }
#else
__device__ __forceinline__ double lane_m1(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double lane_p1(double v) { return __shfl_down(v, 1, 64); }
__device__ __forceinline__ double lane_m2(double v) { return __shfl_up(v, 2, 64); }
__device__ __forceinline__ double lane_p2(double v) { return __shfl_down(v, 2, 64); }
#endif
__device__ __forceinline__ Cons lane_m1(const Cons &U)
{
    return Cons{lane_m1(U.d), lane_m1(U.E), lane_m1(U.mx), lane_m1(U.my)};
}
__device__ __forceinline__ Cons lane_p1(const Cons &U)
{
    return Cons{lane_p1(U.d), lane_p1(U.E), lane_p1(U.mx), lane_p1(U.my)};
}

// Per-lane stash in LDS for the values a row hands to the next iteration
// (slot s of lane l at doubles s*64 + l: conflict-free 512-B rows).  Own-lane
// data only, so no barrier: LDS operations of one wavefront complete in order.
// Every slot is read (old row) before it is written (new row) within an
// iteration, so one slot per value is enough.
constexpr int ST_YM = 0, ST_YP = 4, ST_FXT = 8, ST_XP = 12, ST_XPC = 16, ST_FX = 20;
constexpr int ST_L2 = 24;         // limit2_x of two rows: 2 x 4 slots, the older row at (k & 1)
constexpr int ST_AX = 32, ST_AY = 33;   // running maxima of |u| + c, |v| + c of the lane's new cells
constexpr int ST_XPQ = 34;        // (un, ut, p) of the uncorrected XP (fast build)
constexpr int ST_SLOTS = 37;
// Uniform doubles (kernel parameters) live in a table behind the stash and are
// read with a broadcast ds_read where a stage needs them.  Left in the argument
// registers they do not fit: ~50 uniform doubles (parameters, x / y variants,
// VALU-derived quotients) exceed the 102 SGPRs, the overflow is parked in VGPRs
// and scratch as loop invariants and reloaded every iteration through vmcnt,
// i.e. behind the HBM prefetch of the next row.  `volatile` keeps the compiler
// from hoisting the reads (and what is derived from them) out of the row loop.
enum { C_GAMMA, C_DX, C_DY, C_DT, C_Z0, C_Z1, C_DELTA, C_CVISC, C_SMALLD, C_DTDX, C_DTDY, C_HDTV,
       C_DTDV, C_GRAV, C_HEATR, C_GM1, C_RGM1, C_RDX, C_RDY, C_KSL, C_KSR, C_RGP1,
       // fast build: products of the above that the stages use as one factor
       C_KX, C_KY,          // -hdtV dy, -hdtV dx   (transverse corrections)
       C_CX, C_CY,          // dtdV dy, dtdV dx     (conservative update)
       C_CVDX, C_CVDY,      // cvisc dx, cvisc dy   (artificial viscosity)
       C_N };
constexpr size_t WLDS_BYTES = (size_t)(ST_SLOTS * 64 + C_N) * sizeof(double);
#define UC(name) (ct[C_##name])
// Where a uniform comes from.  Bit-faithful build: the table (its stages keep more values
// alive; scalar registers spilled into vector lanes cost it registers it does not have:
// 16.4 vs 15.7 ms).  Fast build: scalar registers -- kernel arguments, this step's dt
// quotients (scalar loads from the device-side scalars) and the products the stages use,
// computed once per wavefront and moved to scalar registers -- measured 9.4 vs 9.9 ms at
// 16384^2: a scalar operand needs no LDS round trip, and what the allocator spills goes to
// vector LANES (v_readlane, no memory).
#if PYRO_FAST && !defined(PYRO_EMU)
#define US(name, expr) (expr)
#else
#define US(name, expr) UC(name)
#endif
// ... a uniform that a row uses once (artificial viscosity, the two correction factors, the update's
// factors, 1 / dx): experiment PYRO_WAVE_FEW_SGPR reads those from the table in the fast build too
#if defined(PYRO_WAVE_FEW_SGPR)
#define US1(name, expr) UC(name)
#else
#define US1(name, expr) US(name, expr)
#endif
#if defined(PYRO_WAVE_FEW_SGPR) && PYRO_WAVE_FEW_SGPR >= 2
#define US2(name, expr) UC(name)
#else
#define US2(name, expr) US(name, expr)
#endif
#if defined(PYRO_WAVE_FEW_SGPR) && PYRO_WAVE_FEW_SGPR >= 3
#define US3(name, expr) UC(name)
#else
#define US3(name, expr) US(name, expr)
#endif
// an entry only one of the two builds uses (the table reads are volatile: an unused one
// would still be issued)
#if defined(PYRO_EMU)     // (the emulated fast build divides by the operand, not by its reciprocal)
#define UC_FAST(name) UC(name)
#define UC_EXACT(name) UC(name)
#elif PYRO_FAST
#define UC_FAST(name) UC(name)
#define UC_EXACT(name) 0.0
#else
#define UC_FAST(name) 0.0
#define UC_EXACT(name) UC(name)
#endif
#define UC_GASK() GasKTab{ct, US3(GAMMA, P.gamma)}
// (an explicit LDS pointer type: a plain `volatile double *` is a generic pointer
// that the address-space inference leaves alone, i.e. flat loads through vmcnt)
#if defined(PYRO_EMU)
typedef volatile double *UniformTab;
#else
typedef volatile __attribute__((address_space(3))) double *UniformTab;
#endif

// gamma and the uniform quotients of it the HLLC solver uses (hydro.h GasK), read from the
// table where a branch needs them: ksl / ksr only on compressed faces, rgp1 only in the
// two-shock estimate (as a GasK value all four were read for every Riemann problem)
struct GasKTab {
    UniformTab ct;
    double gam;
    __device__ __forceinline__ double g() const { return gam; }
    __device__ __forceinline__ double sl() const { return ct[C_KSL]; }
    __device__ __forceinline__ double sr() const { return ct[C_KSR]; }
    __device__ __forceinline__ double gp1() const { return ct[C_RGP1]; }
};

struct FlatKTab {     // flattening parameters, read past the early exits only
    UniformTab ct;
    __device__ __forceinline__ double z0() const { return ct[C_Z0]; }
    __device__ __forceinline__ double z1() const { return ct[C_Z1]; }
    __device__ __forceinline__ double delta() const { return ct[C_DELTA]; }
};

__device__ __forceinline__ Cons st_get(const double *st, int s)
{
    return Cons{st[s * 64], st[(s + 1) * 64], st[(s + 2) * 64], st[(s + 3) * 64]};
}
__device__ __forceinline__ void st_put(double *st, int s, const Cons &U)
{
    st[s * 64] = U.d; st[(s + 1) * 64] = U.E; st[(s + 2) * 64] = U.mx; st[(s + 3) * 64] = U.my;
}


// MOL: the method-of-lines right-hand side of compressible_rk (compressible_rk/fluxes.py:28-180,
// simulation.py:10-44) instead of the CTU step: the same march -- density floor, primitives,
// flattening, limited slopes, artificial viscosity, one Riemann problem per face -- with
// piecewise linear face states (no tracing), no transverse problems / corrections, and
// k = -div F + S stored in place of the new state (Uout = four planes of the k state; the
// arithmetic of the staged k_rk_states / k_rk_flux / k_rk_rhs of compressible.hip, expression by
// expression).  No sponge (the staged set carries it).
// ONE: this launch is the whole step of a device-side run (pyrohip_comp_evolve, comp_api.hip):
// ghost cells are read through the boundary rules (no filled frame) and the last wavefront to
// finish runs the driver's dt policy for the next step.
// RKF (with MOL): the Runge-Kutta stage folded into the launch -- the stage state built at load from
// y_0 and the earlier increments, ghost cells through the boundary rules, and in the last stage
// the final update + the CFL minimum of compressible_rk instead of the k store (fused_common.h: FP::rk_*)
// SRC = false: an instance without the source-term blocks (gravity, heating) for runs that have none --
// skipped at run time they still cost registers whose pending loads force a full vmcnt wait where their
// paths join the row's work: 31.7 -> 32.3 Gcell/s at 16384^2 (round 6)
#if defined(PYRO_WAVE_TIMELINE) && PYRO_FAST && !defined(PYRO_EMU)
// developer build (tools/wave_timeline.py): every wavefront of k_ctu_wave leaves (start, end) on the
// 100 MHz constant clock, its hardware id and its unit in a device array read back by
// pyrohip_debug_wave_timeline -- where does a one-round launch lose the time between the average
// wavefront's life and the kernel's?
__device__ unsigned long long g_wave_timeline[4 * 65536];
#endif
template <int SOLVER, bool STD, bool MOL = false, bool ONE = false, bool RKF = false, bool SRC = true,
          int FINT = -1, bool FB = false>   // SOLVER, STD as k_ctu_fused
__global__ __launch_bounds__(64, PYRO_WAVE_MINW) void k_ctu_wave(const double *__restrict__ Uin,
                                                                 double *__restrict__ Uout, Geom g,
                                                                 FP P, int *__restrict__ flag,
                                                                 double *__restrict__ partial,
                                                                 const StepScalars *__restrict__ S)
{
    HIP_DYNAMIC_SHARED(double, lds)
    static_assert(!RKF || (MOL && !ONE), "the folded Runge-Kutta stage is a method-of-lines launch");
    constexpr bool MAPS = ONE || RKF;      // ghost cells are read through the boundary rules
    // FINT: the last stage (final update instead of the k store) known to the compiler: 1 / 0; -1 = P.rk_final.
    // One instance for both carried the last stage's operands through the others' rows (255 registers, 44 B of
    // scratch per lane in the contracted build)
    const bool rk_final = RKF && (FINT < 0 ? P.rk_final != 0 : FINT == 1);
    const int l = threadIdx.x;
    double *st = lds + l;
    // workgroup -> (column strip, row strip): workgroups are dealt round-robin to the 8 XCDs
    // (each with its own L2); XCD x takes the units [x per, (x + 1) per) in order, so the strips
    // that share apron rows and the cache lines at a column cut meet in ONE L2 at about the
    // same time (the launch pads the grid to a multiple of 8)
    const int per = (P.nunits + 7) / 8;
    int unit = ((int)blockIdx.x % 8) * per + (int)blockIdx.x / 8;
    if (P.units_short > 0) {
        // (many rounds: every XCD's queue ends with its share of the short strips -- wave_short_tail below)
        const int NL = P.nunits - P.units_short, NS = P.units_short;
        const int perL = (NL + 7) / 8, perS = (NS + 7) / 8;
        const int x = (int)blockIdx.x % 8, r = (int)blockIdx.x / 8;
        if (r < perL) unit = (x * perL + r < NL) ? x * perL + r : P.nunits;
        else unit = (x * perS + r - perL < NS) ? NL + x * perS + r - perL : P.nunits;
    }
    if (unit >= P.nunits) return;
    // (units behind the ncb x nsb regular ones: the extra strip of the column strips [0, n_extra) -- a launch
    // that fits the resident slots in one round is cut into exactly as many strips as there are slots,
    // wave_extra_units below)
    const int nreg = P.ncb * P.nsb;
    const int cb = unit < nreg ? unit % P.ncb : unit - nreg;
    const int sb = unit < nreg ? P.sb_first + (unit / P.ncb) * P.sb_step : P.nsb;
    int i0 = g.ilo + sb * P.L;                             // strip rows [i0, i1)
    // (the last strip runs to the end of the grid: it may be up to ng - 1 rows longer
    // than L, so that no strip is shorter than the ghost width -- comp_step_wave_ex)
    int i1 = (sb == P.nsb - 1) ? g.ihi + 1 : i0 + P.L;
    if (cb < P.n_extra) {      // nsb + 1 strips of equal length (to a row)
        i0 = g.ilo + (int)((long)sb * g.nx / (P.nsb + 1));
        i1 = g.ilo + (int)((long)(sb + 1) * g.nx / (P.nsb + 1));
    }
    if (P.n_short > 0 && sb >= P.nsb - P.n_short - P.n_tail) {
        // long strips, then n_short short ones, then n_tail long ones (a slab's last boundary strip)
        const int nl = P.nsb - P.n_short - P.n_tail;
        if (sb < nl + P.n_short) { i0 = g.ilo + nl * P.L + (sb - nl) * P.Ls; i1 = i0 + P.Ls; }
        else { i0 = g.ilo + nl * P.L + P.n_short * P.Ls + (sb - nl - P.n_short) * P.L; i1 = i0 + P.L; }
        if (sb == P.nsb - 1) i1 = g.ihi + 1;
    }
    const int j = g.jlo + cb * WOUT - 4 + l;               // this lane's column
    const int jc = (j < g.qy) ? j : g.qy - 1;              // ragged last strip: clamp, unused
    const bool jin = (j >= g.jlo && j <= g.jhi);
    const bool jout = jin && l >= 4 && l <= 59;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const int limiter = STD ? 2 : P.limiter;
    const bool flat = STD || P.use_flattening;
    const bool HAVE_SRC = SRC && P.have_src;
    // fast build, HLLC: the transverse Riemann problems take the traced primitive
    // face states as they are (hllc_flux_impl<true>)
    constexpr bool TQ = (PYRO_FAST != 0) && (SOLVER == 0) && !MOL;
    UniformTab ct = (UniformTab)(lds + ST_SLOTS * 64);
    // this step's dt and its quotients: by value (single steps), from the step scalars in device
    // memory (device-side run, k_dt_policy between the launches), or -- ONE -- derived here from
    // the previous launch's CFL minima (common.h: StepPolicy)
    double v_dt = P.dt, v_dtdx = P.dtdx, v_dtdy = P.dtdy, v_hdtV = P.hdtV, v_dtdV = P.dtdV;
    bool active = true;
    if (ONE) {
        StepPolicy *const pol = P.pol;
        const int m = P.pol_m;
        if (P.pol_pre) {
            const StepScalars *Sm = &pol->S[m & 1];
            v_dt = Sm->dt; v_dtdx = Sm->dtdx; v_dtdy = Sm->dtdy; v_hdtV = Sm->hdtV; v_dtdV = Sm->dtdV;
            active = Sm->active != 0;
        } else {
            StepScalars L = pol->S[(m - 1) & 1];
            const unsigned long long *prev = pol->slots + (size_t)((m - 1) % 3) * kPolSetWords;
            // (plain loads: what the previous launch left is visible across a kernel boundary like
            // the state itself, and the 20 000 wavefronts of a launch then share 64 L2 lines -- loads
            // past the L2 made this prologue ~5 us per wavefront)
            const double cmin = __shfl(wave_reduce_min(__longlong_as_double(
                                           (long long)prev[(size_t)l * kPolStride])), 0, 64);
            double dtm;
            dt_policy_apply(&L, cmin, (*(volatile int *)flag & (2 << ((m - 1) & 1))) != 0, &dtm, 0, 0);
            if (unit == 0) {     // the books, and the slots of the launch after this one
                atomicExch(&pol->slots[(size_t)((m + 1) % 3) * kPolSetWords + (size_t)l * kPolStride],
                           (unsigned long long)__double_as_longlong((double)INFINITY));
                if (l == 0) { pol->S[m & 1] = L; pol->dts[m] = dtm; }
            }
            v_dt = L.dt; v_dtdx = L.dtdx; v_dtdy = L.dtdy; v_hdtV = L.hdtV; v_dtdV = L.dtdV;
            active = L.active != 0;
        }
    } else if (S) {
        v_dt = S->dt; v_dtdx = S->dtdx; v_dtdy = S->dtdy; v_hdtV = S->hdtV; v_dtdV = S->dtdV;
        active = S->active != 0;
    }
    // the wavefront's CFL minimum (in every lane) and its positivity flag at the end
#if defined(PYRO_WAVE_TIMELINE) && PYRO_FAST && !defined(PYRO_EMU)
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto finish = [&](double cfl, bool bad) {
        if (bad) atomicOr(flag, ONE ? (2 << (P.pol_m & 1)) : 1);
        if (l != 0) return;
#if defined(PYRO_WAVE_TIMELINE) && PYRO_FAST && !defined(PYRO_EMU)
        if (unit < 65536) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_wave_timeline[4 * unit] = tl_t0;
            g_wave_timeline[4 * unit + 1] = __builtin_amdgcn_s_memrealtime();
            g_wave_timeline[4 * unit + 2] = ((unsigned long long)xcc << 32) | hwid;
            g_wave_timeline[4 * unit + 3] = ((unsigned long long)sb << 32) | (unsigned)cb;
        }
#endif
        if (ONE)
            atomicMin(P.pol->slots + (size_t)(P.pol_m % 3) * kPolSetWords + (size_t)(unit % kPolSlots) * kPolStride,
                      (unsigned long long)__double_as_longlong(cfl));
        else
            partial[sb * P.ncb + cb] = cfl;
    };
    if (!active) {
        // device-side run, past tmax or after an invalid state: nothing happens (the
        // host picks the buffer that holds the last state that did advance, comp_evolve)
        if (!ONE) finish(INFINITY, false);
        return;
    }
#if PYRO_FAST && !defined(PYRO_EMU)
    // this step's dt quotients (device-side run: scalar loads from the step scalars) and the
    // products the stages use as one factor: computed once, kept in scalar registers
    auto sgpr = [](double v) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                                __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    // (ONE: uniform values out of vector arithmetic)
    const double s_dtdx = ONE ? sgpr(v_dtdx) : v_dtdx, s_dtdy = ONE ? sgpr(v_dtdy) : v_dtdy;
    const double s_hdtV = ONE ? sgpr(v_hdtV) : v_hdtV, s_dtdV = ONE ? sgpr(v_dtdV) : v_dtdV;
    const double s_kx = sgpr(-s_hdtV * P.dy), s_ky = sgpr(-s_hdtV * P.dx);
    const double s_cx = sgpr(s_dtdV * P.dy), s_cy = sgpr(s_dtdV * P.dx);
    const double s_cvdx = sgpr(P.cvisc * P.dx), s_cvdy = sgpr(P.cvisc * P.dy);
#endif
    if (l == 0) {     // one wavefront, LDS operations complete in order: no barrier needed
        // (device-side run: this step's dt and its quotients live in device memory)
        ct[C_GAMMA] = P.gamma; ct[C_DX] = P.dx; ct[C_DY] = P.dy; ct[C_DT] = v_dt;
        ct[C_Z0] = P.z0; ct[C_Z1] = P.z1; ct[C_DELTA] = P.delta; ct[C_CVISC] = P.cvisc;
        ct[C_SMALLD] = P.small_dens;
        ct[C_DTDX] = v_dtdx; ct[C_DTDY] = v_dtdy;
        ct[C_HDTV] = v_hdtV; ct[C_DTDV] = v_dtdV;
        ct[C_GRAV] = P.grav; ct[C_HEATR] = P.heat_rate;
        ct[C_GM1] = P.gamma - 1.0; ct[C_RGM1] = prcp(P.gamma - 1.0);
        ct[C_RDX] = prcp(P.dx); ct[C_RDY] = prcp(P.dy);
        const GasK K0 = make_gask(P.gamma);
        ct[C_KSL] = K0.ksl; ct[C_KSR] = K0.ksr; ct[C_RGP1] = K0.rgp1;
#if PYRO_FAST
        const double hdtV = v_hdtV, dtdV = v_dtdV;
        ct[C_KX] = -hdtV * P.dy; ct[C_KY] = -hdtV * P.dx;
        ct[C_CX] = dtdV * P.dy; ct[C_CY] = dtdV * P.dx;
        ct[C_CVDX] = P.cvisc * P.dx; ct[C_CVDY] = P.cvisc * P.dy;
#endif
    }
#if defined(PYRO_EMU)
    // (the emulator's lanes are fibers that run ahead of each other between two shuffles: after
    // the shuffles of the dt policy above, lane 0 need not be the first to go on)
    if (ONE) hipemu::wave_barrier();
#endif

    // ONE: a ghost cell is read from the cell its boundary rule copies from (x rule, then y rule:
    // array_indexer.py:163-274), with the sign of the variables that reflect oddly -- what
    // fill_BC_all would have left in it, so the frame of Uin need not be filled.  The sign is
    // applied where the row is consumed, an iteration after its load (fix_sign).
    const int jsrc = MAPS ? bc_src(P.mc, jc, g.jlo, g.jhi) : jc;
    // RKF: this stage's weights dt a_sj (integration.py:116: self.dt*a[istage, s], then times k)
    const double rkc0 = RKF ? v_dt * P.rk_a[0] : 0.0, rkc1 = RKF ? v_dt * P.rk_a[1] : 0.0,
                 rkc2 = RKF ? v_dt * P.rk_a[2] : 0.0;
    // RKF: the LAST increment with a non-zero weight is added where the row is consumed, an
    // iteration after its load (loadK / addK below) -- added right behind the loads, inside
    // loadU, the sum waits for the row that was just requested: the stages ran at a VALU busy of
    // 0.45 (first counter pass; 0.65 with this: profiles/r05_rk4096_pmc.json; 0.57-0.60 ms per stage against 0.46 for the plain
    // right-hand side).  Earlier increments (TVD3's second stage has two) are added eagerly;
    // the order of accumulation is the reference's either way.
    // (contracted build only: the bit-faithful one reads every row twice, and a second deferred
    // increment in flight costs it more in spills than the wait -- 3.9 -> 4.8 ms per RK4 step at 4096^2)
    int rk_last = -1;
    if (RKF && PYRO_FAST) {
        for (int jn = 0; jn < 3; jn++)
            if (jn < P.rk_n && P.rk_a[jn] != 0.0) rk_last = jn;
    }
    const int rk_eager = (RKF && !PYRO_FAST) ? 3 : rk_last;      // increments j < rk_eager are added at the load
    const double rkcl = rk_last == 0 ? rkc0 : (rk_last == 1 ? rkc1 : rkc2);
    auto src_of = [&](int row) {
        row = row < 0 ? 0 : (row > g.qx - 1 ? g.qx - 1 : row);
        const int srow = MAPS ? bc_src(P.mr, row, g.ilo, g.ihi) : row;
        return (size_t)srow * p + jsrc;
    };
    // Plain instances (a filled frame, no Runge-Kutta stage): the rows of a strip are addressed as
    // [scalar base of the strip's first row, per plane] + [32-bit byte offset: row offset (scalar) +
    // lane offset] -- the `saddr` form of global_load / global_store, ONE v_add_u32 per row instead of
    // a 64-bit multiply-add + shift + four 64-bit adds per four-plane access (18 of the 870 vector
    // instructions of a row).  A strip spans at most 160 + 15 rows: the offset stays far below 4 GB.
#if !defined(PYRO_EMU)
    constexpr bool SADDR = !MAPS && !RKF;
#else
    constexpr bool SADDR = false;
#endif
    const int rbase = (i0 - 8 > 0) ? i0 - 8 : 0;           // first row the strip touches (k - 4 of its first iteration)
    const char *const sbase_in = (const char *)(Uin + (size_t)rbase * p);
    char *const sbase_out = (char *)(Uout + (size_t)rbase * p);
    const unsigned pitch8 = (unsigned)p * 8u, lane8 = (unsigned)jc * 8u;
    const size_t plb = pl * sizeof(double);
    auto loadU = [&](int row) {
        if (SADDR) {
            row = row < 0 ? 0 : (row > g.qx - 1 ? g.qx - 1 : row);
#if defined(PYRO_WAVE_EXP_L2ROWS)      // (timing experiment, WRONG results: every strip re-reads its first 8 rows -- cache hits)
            row = rbase + ((row - rbase) & 7);
#endif
            const unsigned off = (unsigned)(row - rbase) * pitch8 + lane8;
            return Cons{*(const double *)(sbase_in + off), *(const double *)(sbase_in + plb + off),
                        *(const double *)(sbase_in + 2 * plb + off), *(const double *)(sbase_in + 3 * plb + off)};
        }
        const size_t kk = src_of(row);
        Cons U{Uin[kk], Uin[pl + kk], Uin[2 * pl + kk], Uin[3 * pl + kk]};
        if (RKF) {
            // (the source cell of a mapped ghost cell is an interior cell: y_0 + the increments, in
            // the reference's order of accumulation; a zero weight adds nothing and is not read)
            const double *K = P.rk_k + kk;
            if (rk_eager > 0 && P.rk_n > 0 && P.rk_a[0] != 0.0) {
                U.d += rkc0 * K[0]; U.E += rkc0 * K[pl]; U.mx += rkc0 * K[2 * pl]; U.my += rkc0 * K[3 * pl];
            }
            if (rk_eager > 1 && P.rk_n > 1 && P.rk_a[1] != 0.0) {
                const double *K1 = K + 4 * pl;
                U.d += rkc1 * K1[0]; U.E += rkc1 * K1[pl]; U.mx += rkc1 * K1[2 * pl]; U.my += rkc1 * K1[3 * pl];
            }
            if (rk_eager > 2 && P.rk_n > 2 && P.rk_a[2] != 0.0) {
                const double *K2 = K + 8 * pl;
                U.d += rkc2 * K2[0]; U.E += rkc2 * K2[pl]; U.mx += rkc2 * K2[2 * pl]; U.my += rkc2 * K2[3 * pl];
            }
        }
        return U;
    };
    auto loadK = [&](int row) {      // the deferred increment of that row (RKF; unused elsewhere)
        if (!RKF || rk_last < 0) return Cons{0.0, 0.0, 0.0, 0.0};
        const double *K = P.rk_k + (size_t)(4 * rk_last) * pl + src_of(row);
        return Cons{K[0], K[pl], K[2 * pl], K[3 * pl]};
    };
    auto addK = [&](Cons &U, const Cons &K) {
        if (RKF && rk_last >= 0) {
            U.d += rkcl * K.d; U.E += rkcl * K.E; U.mx += rkcl * K.mx; U.my += rkcl * K.my;
        }
    };
    auto fix_sign = [&](Cons &U, int row) {
        if (!MAPS || !P.odd) return;
        const unsigned sd = (j < g.jlo ? 4u : 0u) | (j > g.jhi ? 8u : 0u) |
                            (row < g.ilo ? 1u : 0u) | (row > g.ihi ? 2u : 0u);
        U.d = odd_sides(P.odd & sd) ? -U.d : U.d;
        U.E = odd_sides((P.odd >> 4) & sd) ? -U.E : U.E;
        U.mx = odd_sides((P.odd >> 8) & sd) ? -U.mx : U.mx;
        U.my = odd_sides((P.odd >> 12) & sd) ? -U.my : U.my;
    };
    auto row_in = [&](int r) { return r >= g.ilo && r <= g.ihi; };
    // (un, ut, p) of a conserved face state in the normal frame, as HLLC derives them
    auto faceq = [&](const ConsN &U, double gamma) {
        const double ri = prcp(U.d);
        const double un = U.mn * ri, ut = U.mt * ri;
        return FaceQ{un, ut, (U.E - 0.5 * U.d * (un * un + ut * ut)) * (gamma - 1.0)};
    };

    // state carried from one iteration to the next in registers (rows relative
    // to the iteration k that is about to start) ...
    double wr[5] = {1, 1, 1, 1, 1}, wu[5] = {0, 0, 0, 0, 0};   // Q window, rows k-4..k
    double wv[5] = {0, 0, 0, 0, 0}, wp[5] = {1, 1, 1, 1, 1};
    double fxa = 1.0, fxb = 1.0;                               // flatten_x of rows k-4, k-3
    Cons Ue{1.0, 1.0, 0.0, 0.0}, Uem = Ue;                     // old state, rows k-3 / k-4
    double Dp = 0.0;                                           // vertex div(U) of row k-4
    double up = 0.0, vp = 0.0;                                 // u, v at (k-4, j-1)
    Cons Upre = loadU(i0 - 4);                                 // row k, in flight
    Cons Kpre = loadK(i0 - 4);                                 // (RKF: its deferred increment)
    // (method of lines, contracted build: the old state of rows k-3 / k-4 -- artificial viscosity,
    // source terms -- is rebuilt from the primitive window instead of read a second time: with the
    // Runge-Kutta stage folded into the load the second read costs the increments' planes too,
    // and three rows of y_0 + k_j per wavefront no longer sit in the L2 between the two reads --
    // an RKF stage took 0.65-1.37 ms at 4096^2 against 0.49 for the plain right-hand side)
    // DELAY (round 6; the plain contracted CTU instance): the four stores of a row's update are issued at
    // the top of the NEXT iteration, between the consumption of the row that arrived and the request of
    // the next one.  Stores and loads share the vmcnt counter and complete in order: with the stores at
    // the end of the iteration, the wait for the prefetched row at the top of the next one also waited
    // for the stores issued just before (they are conditional, the compiler must assume none is younger
    // than the loads: s_waitcnt vmcnt(0)) -- a fifth of a wavefront's cycles (SQ_WAIT_INST_ANY 55 k of 255 k
    // per wavefront, SQ_WAIT_INST_LDS 3 k: profiles/r06_default16384_fm1_pmc.json).  The new state waits in
    // registers.  To make room, the old states of rows k-3 / k-4 that feed the artificial-viscosity fluxes
    // are rebuilt from the primitive window (conservative either way: a flux); the old state the UPDATE
    // starts from is NOT -- the first version rebuilt that one too, and a conserved state that goes through
    // u = m / rho, p = (gamma - 1)(E - ...) and back every step drifts: the developed 1024^2 blast (2333 steps)
    // was 1.1e-9 from the oracle instead of 4e-11.  It is read a second time as before (a cache hit), requested
    // behind the delayed stores at the top of the iteration whose end consumes it (Ucx).
#if !defined(PYRO_WAVE_NO_DELAY)
    constexpr bool REBUILD = (PYRO_FAST != 0) && !MOL && !MAPS && SADDR;   // (viscosity operands only; GPU build)
    // (the bit-faithful build must keep the second read -- rebuilding changes bits -- and has no eight
    // registers for the waiting state: with the delayed stores it spills 76 instead of 28 B per lane,
    // 15.7 -> 17.3 ms per step at 16384^2; the method-of-lines stages are fabric-bound: 2.13 -> 2.15 ms)
    constexpr bool DELAY = REBUILD && SADDR;
#else
    constexpr bool REBUILD = false, DELAY = false;                   // (developer A/B)
#endif
    constexpr bool NOREP = (PYRO_FAST != 0) && (MOL || REBUILD);
    Cons Urep = NOREP ? Cons{1.0, 1.0, 0.0, 0.0} : loadU(i0 - 7);    // row k-3 again, in flight
    Cons Krep = NOREP ? Cons{0.0, 0.0, 0.0, 0.0} : loadK(i0 - 7);
    bool bad = false;
    Cons Upend{0.0, 0.0, 0.0, 0.0};        // DELAY: the new state of row k-5, stored at the top of iteration k
    Cons Ucx{1.0, 1.0, 0.0, 0.0};          // DELAY: the old state of row k-4 (second read), for the update
    // ... and in the stash: uncorrected YM, YP, XP, FxT, corrected XP and Fx of row k-4
    {
        const Cons one{1.0, 1.0, 0.0, 0.0};
        st_put(st, ST_YM, one); st_put(st, ST_YP, one); st_put(st, ST_XP, one);
        st_put(st, ST_XPC, one); st_put(st, ST_FXT, one); st_put(st, ST_FX, one);
        st_put(st, ST_L2, one); st_put(st, ST_L2 + 4, one);
        st[ST_AX * 64] = 0.0; st[ST_AY * 64] = 0.0;
        st[ST_XPQ * 64] = 0.0; st[(ST_XPQ + 1) * 64] = 0.0; st[(ST_XPQ + 2) * 64] = 1.0;
    }

#if !defined(PYRO_EMU)
    // The two wavefronts of a SIMD are arbitrated by age: the older one runs almost as if it
    // were alone, the younger one gets what is left, and when the older one is done the younger
    // finishes alone at half the issue rate -- measured per wavefront at 4096^2 (one round of
    // resident wavefronts; tools/timeline_report.py): 1.11 M cycles for one half of the
    // wavefronts, 1.56 M for the other half, on every SIMD.  With many rounds a new wavefront
    // takes the place of the finished one and nothing is lost; with one or two rounds the
    // tail is a third of the launch.  So the two take turns at the higher priority, a few
    // rows each (by the wavefront's slot number on its SIMD), and finish together.
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int wslot = (int)(hw_id & 1u);
    // Round 6, second half: the turns by phase are blind -- each wavefront counts its OWN rows, the two
    // drift apart and the better duty changes with the strip length (tools/wave_timeline.py, 4096^2: the
    // second wavefront ends 90 us before the first with seven eighths, 80 us after it with four, together with
    // five; 3072^2 wants seven).  With a board (P.prio_board: one word per SIMD and slot in device memory, the
    // two wavefronts of a SIMD sit on one XCD = one L2) each tells the other how many rows it has left, a row
    // late, and the one with more rows left takes the priority: the pair ends together whatever the strips.
    // (FB: instances of their own -- the launches of many rounds, the headline's, keep the code they had; the
    // method-of-lines instances have no register for it)
    const bool prio_fb = FB && !MOL && P.prio_board != nullptr;
    int *prio_mine = nullptr, *prio_other = nullptr;
    int prio_seen = 0;
    if (prio_fb) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        int *const pair = P.prio_board + 2 * (int)(((xcc & 15u) << 12) | ((hw_id >> 4) & 0xfffu));   // (SIMD, CU, SH, SE)
        prio_mine = pair + wslot;
        prio_other = pair + (wslot ^ 1);
    }
    // (the word goes out and the other one is asked for BEHIND the request of the next row: loads return in
    // order, in front of it a miss of this word would hold up the row; workgroup scope -- sc0: through the
    // CU's cache to the XCD's L2, the two wavefronts share both)
    auto prio_publish = [&](int k) {
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
            // (the word asked for an iteration ago: complete with the row that arrived since)
            const int seen = __builtin_amdgcn_readfirstlane(prio_seen);
            const int left = i1 + 3 - k;
            const int other_left = ((seen >> 16) == P.prio_tag) ? (seen & 0xffff) : left;
            if (left > other_left) __builtin_amdgcn_s_setprio(1);
            else if (left < other_left) __builtin_amdgcn_s_setprio(0);
        } else
        if (P.prio_duty > 0) {
            const int phase = ((k - i0) >> PYRO_WAVE_PRIO_SHIFT) & 7;
            if (wslot ? (phase < P.prio_duty) : (phase >= P.prio_duty)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
#pragma unroll
        for (int n = 0; n < 4; n++) {
            wr[n] = wr[n + 1]; wu[n] = wu[n + 1]; wv[n] = wv[n + 1]; wp[n] = wp[n + 1];
        }
        if (DELAY)     // (row k-4 = window index 0 after the shift: rebuilt, not carried -- eight registers)
            Uem = prim_to_cons_g(Prim{wr[0], wu[0], wv[0], wp[0]}, US2(GM1, P.gm1), US2(RGM1, P.rgm1));
        else
        Uem = Ue;
        if (NOREP) {
            // (window index 1 = row k-3 after the shift above; floor and signs are in the primitives)
            Ue = prim_to_cons_g(Prim{wr[1], wu[1], wv[1], wp[1]}, US2(GM1, P.gm1), US2(RGM1, P.rgm1));
        } else {
        Ue = Urep;
        addK(Ue, Krep);
        fix_sign(Ue, k - 3);
        if (row_in(k - 3) && jin) Ue.d = fmax(Ue.d, US2(SMALLD, P.small_dens));      // clean_state
        }
        // (issuing this second read of row k-2 in the middle of the iteration instead -- eight
        // registers less while the slopes and the first Riemann problems are worked on -- was
        // measured: the allocator spills elsewhere, 10.35 vs 10.46 ms fast, 16.37 vs 15.96 exact)
        if (!NOREP) { Urep = loadU(k - 2); Krep = loadK(k - 2); }
        // ---- S0: row k -> primitives
        {
            Cons U = Upre;
            addK(U, Kpre);
            fix_sign(U, k);
            if (!DELAY) {
            Upre = loadU(k + 1);
            Kpre = loadK(k + 1);
            prio_publish(k);
            }
            const bool interior = row_in(k) && jin;
            // (RKF: the stage state is a temporary of the step -- nothing to keep the floor in)
            if (MOL && !RKF && interior && U.d < US2(SMALLD, P.small_dens))     // clean_state works in place
                const_cast<double *>(Uin)[(size_t)k * p + jc] = US2(SMALLD, P.small_dens);
            if (interior) U.d = fmax(U.d, US2(SMALLD, P.small_dens));
            bool ok;
            const Prim q = cons_to_prim_nb(U, US3(GAMMA, P.gamma), ok);
            if (interior && !ok) bad = true;
            wr[4] = q.r; wu[4] = q.u; wv[4] = q.v; wp[4] = q.p;
            if (DELAY) {
                // (the row that arrived is consumed ABOVE this fence: its wait sees only its own loads
                // outstanding; then the delayed stores, then the request of the next row.  The empty
                // statement pins the primitives HERE: pure arithmetic sinks below a scheduling fence)
#if !defined(PYRO_EMU)
                asm volatile("" : "+v"(wr[4]), "+v"(wu[4]), "+v"(wv[4]), "+v"(wp[4]));
#endif
                STAGE_FENCE();
                if (k - 1 >= i0 + 4 && jout) {      // the update of row k-5, made by iteration k-1
                    const unsigned off = (unsigned)(k - 5 - rbase) * pitch8 + (unsigned)j * 8u;
                    *(double *)(sbase_out + off) = Upend.d; *(double *)(sbase_out + plb + off) = Upend.E;
                    *(double *)(sbase_out + 2 * plb + off) = Upend.mx; *(double *)(sbase_out + 3 * plb + off) = Upend.my;
                }
                Ucx = loadU(k - 4);                 // (in front of the next row's request: its wait leaves that one out)
                Upre = loadU(k + 1);
                prio_publish(k);
                STAGE_FENCE();
            }
        }
        // ---- flatten_x and limit2_x of row k-2 (window index 2)
        double fxn = 1.0, l2n[4] = {0, 0, 0, 0};
        if (k >= i0) {
            if (flat)
                fxn = flatten_1d_k(wp[0], wp[1], wp[3], wp[4], wu[1], wu[3], FlatKTab{ct});
            if (limiter != 0) {
                l2n[0] = LIMIT2(wr[1], wr[2], wr[3]);
                l2n[1] = LIMIT2(wu[1], wu[2], wu[3]);
                l2n[2] = LIMIT2(wv[1], wv[2], wv[3]);
                l2n[3] = LIMIT2(wp[1], wp[2], wp[3]);
            }
        }
        // limit2_x of rows k-4 (slot k & 1) and k-3 from the stash; row k-2 takes the
        // older one's place
        double l2a[4], l2b[4];
        {
            double *sa = st + (ST_L2 + 4 * (k & 1)) * 64, *sb = st + (ST_L2 + 4 * ((k + 1) & 1)) * 64;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                l2a[n] = sa[n * 64]; l2b[n] = sb[n * 64];
                sa[n * 64] = l2n[n];
            }
        }
        double Dn = 0.0, um = 0.0, vm = 0.0;
        // ---- rows c = k-3 (window index 1: slopes, states, x flux) and f = k-4
        // (y flux, update)
        if (k >= i0 + 2) {
            const int i = k - 3;
            const bool xface = (k >= i0 + 3);          // row c has a lower x face in the strip
            const bool frow = (k >= i0 + 4);           // row f is updated by this strip
            // RKF, last stage (contracted build): the operands of row f's final update -- y_0 and the
            // earlier increments of the cell -- are requested HERE and summed behind the slopes
            // (below), the sum waits in the stash slots the method of lines leaves free (ST_FXT).
            // Loaded where they are used, at the end of the iteration, the wavefront stood still
            // for them: the last stage took 1.22 ms against 0.50 for the others.
            const bool yfin = RKF && (PYRO_FAST != 0) && rk_final && frow && jout;
            Cons Yf0{0.0, 0.0, 0.0, 0.0}, Yf1 = Yf0, Yf2 = Yf0, Yf3 = Yf0;
            if (yfin) {
                const size_t kr = (size_t)(i - 1) * p + j;
                Yf0 = Cons{Uin[kr], Uin[pl + kr], Uin[2 * pl + kr], Uin[3 * pl + kr]};
                const double *K0 = P.rk_k + kr;
                if (P.rk_nb > 1 && P.rk_b[0] != 0.0) Yf1 = Cons{K0[0], K0[pl], K0[2 * pl], K0[3 * pl]};
                if (P.rk_nb > 2 && P.rk_b[1] != 0.0) {
                    const double *K1 = K0 + 4 * pl;
                    Yf2 = Cons{K1[0], K1[pl], K1[2 * pl], K1[3 * pl]};
                }
                if (P.rk_nb > 3 && P.rk_b[2] != 0.0) {
                    const double *K2 = K0 + 8 * pl;
                    Yf3 = Cons{K2[0], K2[pl], K2[2 * pl], K2[3 * pl]};
                }
            }
            const double q0[4] = {wr[1], wu[1], wv[1], wp[1]};
            const double qm[4] = {wr[0], wu[0], wv[0], wp[0]};
            const double qp[4] = {wr[2], wu[2], wv[2], wp[2]};
            double ym[4], yp[4], l2y[4] = {0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < 4; n++) {
                ym[n] = lane_m1(q0[n]);
                yp[n] = lane_p1(q0[n]);
                if (limiter != 0) l2y[n] = LIMIT2(ym[n], q0[n], yp[n]);
            }
            um = ym[1]; vm = ym[2];
            double xi = 1.0;
            if (flat) {
                // flatten_multid (reconstruction.py:167-183): own coefficient and the
                // one of the UPWIND neighbour (w.r.t. the pressure gradient)
                const double fy = flatten_1d_k(lane_m2(q0[3]), ym[3], yp[3], lane_p2(q0[3]), ym[2],
                                               yp[2], FlatKTab{ct});
                const double fym = lane_m1(fy), fyp = lane_p1(fy);
                const double px = (qp[3] - qm[3] > 0) ? fxa : fxn;
                const double py = (yp[3] - ym[3] > 0) ? fym : fyp;
                xi = fmin(fmin(fxb, px), fmin(fy, py));
            }
#if PYRO_FAST
            xi = xi + xi;      // (the contracted build's slopes are half slopes: fused_common.h)
#endif
            double dqx[4], dqy[4];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                dqx[n] = xi * SLOPE_SHARED(l2a[n], l2b[n], l2n[n], qm[n], q0[n], qp[n], limiter);
                const double l2m = (limiter == 2) ? lane_m1(l2y[n]) : 0.0;
                const double l2p = (limiter == 2) ? lane_p1(l2y[n]) : 0.0;
                dqy[n] = xi * SLOPE_SHARED(l2m, l2y[n], l2p, ym[n], q0[n], yp[n], limiter);
            }
            if (yfin) {      // integration.py:120-129, the terms before this stage's own increment
                if (P.rk_nb > 1 && P.rk_b[0] != 0.0) {
                    const double w = v_dt * P.rk_b[0];
                    Yf0.d += w * Yf1.d; Yf0.E += w * Yf1.E; Yf0.mx += w * Yf1.mx; Yf0.my += w * Yf1.my;
                }
                if (P.rk_nb > 2 && P.rk_b[1] != 0.0) {
                    const double w = v_dt * P.rk_b[1];
                    Yf0.d += w * Yf2.d; Yf0.E += w * Yf2.E; Yf0.mx += w * Yf2.mx; Yf0.my += w * Yf2.my;
                }
                if (P.rk_nb > 3 && P.rk_b[2] != 0.0) {
                    const double w = v_dt * P.rk_b[2];
                    Yf0.d += w * Yf3.d; Yf0.E += w * Yf3.E; Yf0.mx += w * Yf3.mx; Yf0.my += w * Yf3.my;
                }
                st_put(st, ST_FXT, Yf0);
            }
            STAGE_FENCE();
            // source terms of the face states (apply_source_terms, unsplit_fluxes.py:247-330)
            Cons Ug{0.0, 0.0, 0.0, 0.0};
            double sgn = 1.0, hp = 0.0;
            if (!MOL && HAVE_SRC) {
                const bool ina = (j < g.qy);
                // "ambient" upper boundary: the source ghosts are copies of row jhi
                // (BC.py:159-160), not the sources of the ambient ghost state
                const int js = (P.amb_yhi && j > g.jhi) ? g.jhi : j;
                const size_t kc = (size_t)(ina ? i : g.qx - 1) * p + (ina ? js : g.qy - 1);
                Ug.d = Uin[kc]; Ug.my = Uin[3 * pl + kc];
                if (i >= g.ilo && i <= g.ihi && js >= g.jlo && js <= g.jhi)
                    Ug.d = fmax(Ug.d, US2(SMALLD, P.small_dens));
                sgn = ((j < g.jlo && P.refl_ylo) || (j > g.jhi && P.refl_yhi)) ? -1.0 : 1.0;
                hp = P.heat ? P.heat[(size_t)(ina ? i : g.qx - 1) * p + (ina ? j : g.qy - 1)] : 0.0;
            }
            // vertex divergence at (i-1/2, j-1/2), interface.py:312-330, and the
            // artificial viscosity coefficients of the faces (i, j) in x and (i-1, j)
            // in y (interface.py:366-376: only faces i in [ilo, ihi], j in [jlo, jhi])
            Dn = div_u_vertex_r(q0[1], um, qm[1], up, q0[2], qm[2], vm, vp, UC_EXACT(DX), UC_EXACT(DY),
                                US1(RDX, P.rdx), US1(RDY, P.rdy));
            const double Dn_p = lane_p1(Dn);
            double avx = 0.0, avy = 0.0;
            if (i >= g.ilo && (i <= g.ihi || (P.avx_hi && i == g.ihi + 1)) && jin) {
                const double divU_x = 0.5 * (Dn + Dn_p);
#if PYRO_FAST
                avx = US1(CVDX, s_cvdx) * fmax(-divU_x, 0.0);
#else
                avx = UC(CVISC) * fmax(-divU_x * UC(DX), 0.0);
#endif
            }
            if (j >= g.jlo && (j <= g.jhi || (P.avy_hi && j == g.jhi + 1)) && row_in(i - 1)) {
                const double divU_y = 0.5 * (Dp + Dn);
#if PYRO_FAST
                avy = US1(CVDY, s_cvdy) * fmax(-divU_y, 0.0);
#else
                avy = UC(CVISC) * fmax(-divU_y * UC(DY), 0.0);
#endif
            }
            STAGE_FENCE();
            // -- x states of row c, transverse x flux on its lower face
            Trace lo, hi;
            double gamma = US3(GAMMA, P.gamma);
            if (MOL) {     // fluxes.py:107-140: V_r[i] = q - ld/2, V_l[i+1] = q + ld/2
                lo = Trace{q0[0] + -1.0 * 0.5 * dqx[0], q0[1] + -1.0 * 0.5 * dqx[1],
                           q0[2] + -1.0 * 0.5 * dqx[2], q0[3] + -1.0 * 0.5 * dqx[3]};
                hi = Trace{q0[0] + 1.0 * 0.5 * dqx[0], q0[1] + 1.0 * 0.5 * dqx[1],
                           q0[2] + 1.0 * 0.5 * dqx[2], q0[3] + 1.0 * 0.5 * dqx[3]};
            } else
                trace_states(q0[0], q0[1], q0[2], q0[3], dqx[0], dqx[1], dqx[2], dqx[3], gamma,
                             US2(DTDX, s_dtdx), lo, hi);
            double gm1 = UC_EXACT(GM1), rgm1 = US2(RGM1, P.rgm1);
            Cons XMn = prim_to_cons_g(Prim{lo.r, lo.un, lo.ut, lo.p}, gm1, rgm1);
            Cons XPn = prim_to_cons_g(Prim{hi.r, hi.un, hi.ut, hi.p}, gm1, rgm1);
            FaceQ qxm{lo.un, lo.ut, lo.p}, qxp{hi.un, hi.ut, hi.p};
            if (!MOL && HAVE_SRC) {
                add_grav_to_state(XMn, Ug, UC(GRAV), UC(DT), sgn, UC(HEATR), hp);
                add_grav_to_state(XPn, Ug, UC(GRAV), UC(DT), sgn, UC(HEATR), hp);
                if (TQ) { qxm = faceq(to_nf(XMn, true), gamma); qxp = faceq(to_nf(XPn, true), gamma); }
            }
            // (not initialised: FxTn / Fy / Fyh / Fxn are read only under the wave-uniform
            // conditions they are computed under -- xface / frow -- and an initial value
            // costs four register moves each in every iteration)
            Cons FxTn;
            if (!MOL && xface) {
                if (TQ) {
                    const FaceQ ql{st[ST_XPQ * 64], st[(ST_XPQ + 1) * 64], st[(ST_XPQ + 2) * 64]};
                    FxTn = from_nf(hllc_flux_impl<true>(to_nf(st_get(st, ST_XP), true), to_nf(XMn, true),
                                                        ql, qxm, UC_GASK(), true), true);
                } else
                    FxTn = from_nf(riemann_face<SOLVER>(to_nf(st_get(st, ST_XP), true),
                                                        to_nf(XMn, true), UC_GASK(), true,
                                                        P.solid_xl && i == g.ilo), true);
            }
            if (TQ) {
                st[ST_XPQ * 64] = qxp.un; st[(ST_XPQ + 1) * 64] = qxp.ut; st[(ST_XPQ + 2) * 64] = qxp.p;
            }
            STAGE_FENCE();
            // -- row f: y states corrected with FxT of rows f, f+1; final y flux
            Cons Fy, Fyh;
            if (frow) {
                Cons YMc, YPc;
                if (MOL) { YMc = st_get(st, ST_YM); YPc = st_get(st, ST_YP); }
                else {
                const Cons FxTp = st_get(st, ST_FXT);
#if PYRO_FAST
                const double kx = US1(KX, s_kx);
                YMc = corr_k(st_get(st, ST_YM), FxTn, FxTp, kx);
                YPc = corr_k(st_get(st, ST_YP), FxTn, FxTp, kx);
#else
                const double hdtV = UC(HDTV), Ax = UC(DY);
                YMc = corr(st_get(st, ST_YM), FxTn, FxTp, hdtV, Ax);
                YPc = corr(st_get(st, ST_YP), FxTn, FxTp, hdtV, Ax);
#endif
                }
                Fy = from_nf(riemann_face<SOLVER>(to_nf(lane_m1(YPc), false), to_nf(YMc, false),
                                                  UC_GASK(), false, P.solid_yl && j == g.jlo), false);
                // (DELAY: the exact old state of the row, read a second time -- not the rebuilt one)
                Cons Uv = Uem;
                if (DELAY) { Uv = Ucx; if (jin) Uv.d = fmax(Uv.d, US2(SMALLD, P.small_dens)); }
                const Cons Umy = lane_m1(Uv);
                Fy.d += avy * (Umy.d - Uv.d);
                Fy.E += avy * (Umy.E - Uv.E);
                Fy.mx += avy * (Umy.mx - Uv.mx);
                Fy.my += avy * (Umy.my - Uv.my);
                Fyh = lane_p1(Fy);
            }
            if (!MOL && xface) st_put(st, ST_FXT, FxTn);
            STAGE_FENCE();
            // -- y states of row c, transverse y flux on its lower face
            gamma = US3(GAMMA, P.gamma);
            if (MOL) {     // (normal / transverse frame of the y direction: un = v, ut = u)
                lo = Trace{q0[0] + -1.0 * 0.5 * dqy[0], q0[2] + -1.0 * 0.5 * dqy[2],
                           q0[1] + -1.0 * 0.5 * dqy[1], q0[3] + -1.0 * 0.5 * dqy[3]};
                hi = Trace{q0[0] + 1.0 * 0.5 * dqy[0], q0[2] + 1.0 * 0.5 * dqy[2],
                           q0[1] + 1.0 * 0.5 * dqy[1], q0[3] + 1.0 * 0.5 * dqy[3]};
            } else
                trace_states(q0[0], q0[2], q0[1], q0[3], dqy[0], dqy[2], dqy[1], dqy[3], gamma,
                             US2(DTDY, s_dtdy), lo, hi);
            gm1 = UC_EXACT(GM1); rgm1 = US2(RGM1, P.rgm1);
            Cons YMn = prim_to_cons_g(Prim{lo.r, lo.ut, lo.un, lo.p}, gm1, rgm1);
            Cons YPn = prim_to_cons_g(Prim{hi.r, hi.ut, hi.un, hi.p}, gm1, rgm1);
            FaceQ qym{lo.un, lo.ut, lo.p}, qyp{hi.un, hi.ut, hi.p};
            if (!MOL && HAVE_SRC) {
                add_grav_to_state(YMn, Ug, UC(GRAV), UC(DT), sgn, UC(HEATR), hp);
                add_grav_to_state(YPn, Ug, UC(GRAV), UC(DT), sgn, UC(HEATR), hp);
                if (TQ) { qym = faceq(to_nf(YMn, false), gamma); qyp = faceq(to_nf(YPn, false), gamma); }
            }
            st_put(st, ST_YM, YMn);
            st_put(st, ST_YP, YPn);
            Cons FyT;
            if (MOL) {
            } else if (TQ) {
                const FaceQ ql{lane_m1(qyp.un), lane_m1(qyp.ut), lane_m1(qyp.p)};
                FyT = from_nf(hllc_flux_impl<true>(to_nf(lane_m1(YPn), false), to_nf(YMn, false), ql,
                                                   qym, UC_GASK(), false), false);
            } else
                FyT = from_nf(riemann_face<SOLVER>(to_nf(lane_m1(YPn), false), to_nf(YMn, false),
                                                   UC_GASK(), false, P.solid_yl && j == g.jlo), false);
            STAGE_FENCE();
            // -- transverse correction of the x states of row c, final x flux
            Cons XMc = XMn, XPc = XPn;
#if !PYRO_FAST
            const double Ay = UC(DX);
#endif
            if (!MOL) {
            const Cons FyTh = lane_p1(FyT);            // FyT at (i, j+1)
#if PYRO_FAST
            const double ky = US1(KY, s_ky);
            XMc = corr_k(XMn, FyTh, FyT, ky);
            XPc = corr_k(XPn, FyTh, FyT, ky);
#else
            const double hdtV = UC(HDTV);
            XMc = corr(XMn, FyTh, FyT, hdtV, Ay);
            XPc = corr(XPn, FyTh, FyT, hdtV, Ay);
#endif
            }
            if (TQ) {   // the next row's transverse problem reads (rho, E) + ST_XPQ only
                st[ST_XP * 64] = XPn.d; st[(ST_XP + 1) * 64] = XPn.E;
            } else
                st_put(st, ST_XP, XPn);
            Cons Fxn;
            if (xface) {
                Fxn = from_nf(riemann_face<SOLVER>(to_nf(st_get(st, ST_XPC), true), to_nf(XMc, true),
                                                   UC_GASK(), true, P.solid_xl && i == g.ilo), true);
                Cons Uxm = Uem;
                if (DELAY) { Uxm = Ucx; if (jin && row_in(i - 1)) Uxm.d = fmax(Uxm.d, US2(SMALLD, P.small_dens)); }
                Fxn.d += avx * (Uxm.d - Ue.d);
                Fxn.E += avx * (Uxm.E - Ue.E);
                Fxn.mx += avx * (Uxm.mx - Ue.mx);
                Fxn.my += avx * (Uxm.my - Ue.my);
            }
            st_put(st, ST_XPC, XPc);
            STAGE_FENCE();
            // -- conservative update of row f + CFL of the new state
            if (frow && jout) {
                const Cons Fxp = st_get(st, ST_FX);
                Cons Ucd = Ucx;
                Ucd.d = fmax(Ucd.d, US2(SMALLD, P.small_dens));      // clean_state (the rows updated are interior rows)
                const Cons &Uc = DELAY ? Ucd : Uem;
                Cons Un;   // simulation.py:377-384
                if (MOL) {
                    // compressible_rk/simulation.py:16-42 (k_rk_rhs): k = (Fx[i] - Fx[i+1])/dx +
                    // (Fy[j] - Fy[j+1])/dy + S,  S = (0, ymom g + rho e_rate heat, 0, rho g)
                    const size_t kr = (size_t)(i - 1) * p + j;
                    const double dxx = UC(DX), dyy = UC(DY);
                    Un.d = pdiv(Fxp.d - Fxn.d, dxx) + pdiv(Fy.d - Fyh.d, dyy);
                    Un.E = pdiv(Fxp.E - Fxn.E, dxx) + pdiv(Fy.E - Fyh.E, dyy);
                    Un.mx = pdiv(Fxp.mx - Fxn.mx, dxx) + pdiv(Fy.mx - Fyh.mx, dyy);
                    Un.my = pdiv(Fxp.my - Fxn.my, dxx) + pdiv(Fy.my - Fyh.my, dyy);
                    Un.d = Un.d + 0.0;
                    Un.E = Un.E + (Uc.my * UC(GRAV) + Uc.d * UC(HEATR) * (P.heat ? P.heat[kr] : 0.0));
                    Un.mx = Un.mx + 0.0;
                    Un.my = Un.my + Uc.d * UC(GRAV);
                    if (RKF && rk_final) {
                        // integration.py:120-129: y_0 += dt b_s k_s, s = 0 ... -- the earlier
                        // increments from memory, the last one is Un; then the CFL quantity of
                        // compressible_rk/simulation.py:46-56 on the new state
                        Cons Y;
                        if (PYRO_FAST) Y = st_get(st, ST_FXT);      // (summed behind the slopes: yfin above)
                        else {
                        Y = Cons{Uin[kr], Uin[pl + kr], Uin[2 * pl + kr], Uin[3 * pl + kr]};
#pragma unroll
                        for (int sgm = 0; sgm < 3; sgm++) {
                            if (sgm < P.rk_nb - 1 && P.rk_b[sgm] != 0.0) {
                                const double w = v_dt * P.rk_b[sgm];
                                const double *Ks = P.rk_k + (size_t)(4 * sgm) * pl + kr;
                                Y.d += w * Ks[0]; Y.E += w * Ks[pl]; Y.mx += w * Ks[2 * pl]; Y.my += w * Ks[3 * pl];
                            }
                        }
                        }
                        if (P.rk_b[P.rk_nb - 1] != 0.0) {
                            const double w = v_dt * P.rk_b[P.rk_nb - 1];
                            Y.d += w * Un.d; Y.E += w * Un.E; Y.mx += w * Un.mx; Y.my += w * Un.my;
                        }
                        double *O = P.rk_out + kr;
                        O[0] = Y.d; O[pl] = Y.E; O[2 * pl] = Y.mx; O[3 * pl] = Y.my;
                        double ax, ay;
                        cfl_speeds(Y, US3(GAMMA, P.gamma), ax, ay);
                        st[ST_AX * 64] = fmax(st[ST_AX * 64], pdiv(ax, dxx) + pdiv(ay, dyy));
                    } else {
                    Uout[kr] = Un.d; Uout[pl + kr] = Un.E; Uout[2 * pl + kr] = Un.mx;
                    Uout[3 * pl + kr] = Un.my;
                    }
                } else {
#if PYRO_FAST
                const double cx = US1(CX, s_cx), cy = US1(CY, s_cy);
                Un.d = fma(cx, Fxp.d - Fxn.d, fma(cy, Fy.d - Fyh.d, Uc.d));
                Un.E = fma(cx, Fxp.E - Fxn.E, fma(cy, Fy.E - Fyh.E, Uc.E));
                Un.mx = fma(cx, Fxp.mx - Fxn.mx, fma(cy, Fy.mx - Fyh.mx, Uc.mx));
                Un.my = fma(cx, Fxp.my - Fxn.my, fma(cy, Fy.my - Fyh.my, Uc.my));
#else
                const double dtdV = UC(DTDV), Ax = UC(DY);
                Un.d = Uc.d + dtdV * (Fxp.d * Ax - Fxn.d * Ax + Fy.d * Ay - Fyh.d * Ay);
                Un.E = Uc.E + dtdV * (Fxp.E * Ax - Fxn.E * Ax + Fy.E * Ay - Fyh.E * Ay);
                Un.mx = Uc.mx + dtdV * (Fxp.mx * Ax - Fxn.mx * Ax + Fy.mx * Ay - Fyh.mx * Ay);
                Un.my = Uc.my + dtdV * (Fxp.my * Ax - Fxn.my * Ax + Fy.my * Ay - Fyh.my * Ay);
#endif
                const size_t ko = (size_t)(i - 1) * p + j;
                if (HAVE_SRC)   // simulation.py:406-423
                    grav_update(Un, Uc, UC(GRAV), UC(DT), UC(HEATR), P.heat ? P.heat[ko] : 0.0);
#if defined(PYRO_WAVE_NT_STORE) && !defined(PYRO_EMU)
                __builtin_nontemporal_store(Un.d, &Uout[ko]);
                __builtin_nontemporal_store(Un.E, &Uout[pl + ko]);
                __builtin_nontemporal_store(Un.mx, &Uout[2 * pl + ko]);
                __builtin_nontemporal_store(Un.my, &Uout[3 * pl + ko]);
#else
                if (DELAY) {
                    Upend = Un;
                } else if (SADDR) {
                    const unsigned off = (unsigned)(i - 1 - rbase) * pitch8 + (unsigned)j * 8u;
                    *(double *)(sbase_out + off) = Un.d; *(double *)(sbase_out + plb + off) = Un.E;
                    *(double *)(sbase_out + 2 * plb + off) = Un.mx; *(double *)(sbase_out + 3 * plb + off) = Un.my;
                } else {
                Uout[ko] = Un.d; Uout[pl + ko] = Un.E; Uout[2 * pl + ko] = Un.mx;
                Uout[3 * pl + ko] = Un.my;
                }
#endif
                double ax, ay;   // CFL: running maxima of the divisors, one division at the end
                cfl_speeds(Un, US3(GAMMA, P.gamma), ax, ay);
                st[ST_AX * 64] = fmax(st[ST_AX * 64], ax);
                st[ST_AY * 64] = fmax(st[ST_AY * 64], ay);
                }
            }
            if (xface) st_put(st, ST_FX, Fxn);
        }
        // hand the rows on
        fxa = fxb; fxb = fxn;
        Dp = Dn; up = um; vp = vm;
    }
    if (DELAY && i1 + 3 >= i0 + 4 && jout) {      // the update the last iteration made (row i1 - 1)
        const unsigned off = (unsigned)(i1 - 1 - rbase) * pitch8 + (unsigned)j * 8u;
        *(double *)(sbase_out + off) = Upend.d; *(double *)(sbase_out + plb + off) = Upend.E;
        *(double *)(sbase_out + 2 * plb + off) = Upend.mx; *(double *)(sbase_out + 3 * plb + off) = Upend.my;
    }
    if (bad) atomicOr(flag, 1);
    // min over the lane's cells of min(dx / (|u| + c), dy / (|v| + c)), cfl_cell()
    const double ax = st[ST_AX * 64], ay = st[ST_AY * 64];
    double cfl = fmin(ax > 0.0 ? pdiv(P.dx, ax) : INFINITY, ay > 0.0 ? pdiv(P.dy, ay) : INFINITY);
    // (RKF, last stage: min 1 / ((|u| + c)/dx + (|v| + c)/dy) = 1 / max of the divisor, the
    // correctly rounded reciprocal being monotone)
    if (RKF) cfl = ax > 0.0 ? pdiv(1.0, ax) : INFINITY;
    finish(__shfl(wave_reduce_min(cfl), 0, 64), bad);
}

// Rows per strip (a strip costs L + 8 iterations, of which the 8 warm-up ones are cheap: ~3 % of
// the instructions at L = 69).  What matters is the LAST round of resident wavefronts: the
// launch lasts ceil(wavefronts / slots) strip times, and a last round that fills a third to a
// half of the slots is the worst case -- measured at 8192^2 (147 column strips, 2048 slots,
// profiles/r04_march_sweep.txt): 59 rows = 9.98 rounds 2.525 ms, 74 rows = 7.97 rounds 2.525,
// 75 rows = 7.9 rounds 2.549, 64 rows = 9.19 rounds 2.576, 82 rows = 7.18 rounds 2.588, 69 rows =
// 8.54 rounds 2.624 (round 3's choice, by (rounds + 1/2)(L + 8) with fractional rounds), 112 rows
// = 5.31 rounds 2.628, 128 rows 2.688.  So: the strip length in 32 .. 160 rows that minimises
// (ceil(wavefronts / slots) + 1) x (L + 8) -- WHOLE rounds, plus a strip time for the stragglers of
// the last one (few long rounds end worse than many short ones: the two wavefronts of a SIMD are
// served oldest first) --, the shorter strip winning a tie.  One round when everything is resident at once
// (4096^2: 74 x 27 strips of 152 rows).
static int wave_rows(int nx, int ncb, int slots)
{
    // (round 6: strips down to 8 rows -- grids of 1024^2 ... 1792^2 cells fill the slots with ONE round of
    // 10- to 28-row strips and pass the tile kernel that way: comp_api.hip: wave_kernel_pays_ctu)
    if (nx <= 32) return nx;
    long best_cost = -1;
    int best = 32;
    for (int L = 8; L <= 160 && L <= nx; L++) {
        int nsb = (nx + L - 1) / L;
        if (nsb > 1 && nx - (nsb - 1) * L < 4) nsb--;        // short last strip joins its predecessor
        const int Leff = (nx + nsb - 1) / nsb;                // longest strip
        const long waves = (long)ncb * nsb;
        const long rounds = (waves + slots - 1) / slots;
        // strip times: one round when everything is resident at once, otherwise whole rounds
        // + one more for the stragglers of the last one
        const long cost = (waves <= slots ? 1 : rounds + 1) * (Leff + 8);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = L; }
    }
    return best;
}

// Launches of up to two rounds of resident wavefronts: the second wavefront of a SIMD (by
// its slot number) holds the higher priority six eighths of the time, so that the pair ends
// together (see the kernel; measured at 4096^2, one round: 0.804 -> 0.736 ms, duty 4 / 5 / 6 /
// 7 of 8: 0.777 / 0.764 / 0.736 / 0.744).  With many rounds the arbitration by age is the
// better one (8192^2, 8.5 rounds: 2.66 ms against 2.70 with the priorities).
// (round 6, after the waits of a wavefront shrank: strips of 64 rows and more do better with seven
// eighths -- 3072^2 0.389 -> 0.374 ms, 4096^2 0.639 -> 0.623 with 7, 0.651 with 8 --, the 38-row strips of
// 2048^2 with six: 0.193 vs 0.195 / 0.201 with 5 / 7; the method-of-lines launches pass L = 0 and keep six:
// an RK4 step at 4096^2 took 2.20 ms with seven against 2.13)
#if defined(PYRO_WAVE_PRIO_DUTY_FORCE)      // (developer A/B: tools/wave_timeline.py)
static int wave_prio_duty(int nwaves, int slots, int) { return nwaves <= 2 * slots ? PYRO_WAVE_PRIO_DUTY_FORCE : 0; }
#else
static int wave_prio_duty(int nwaves, int slots, int L) { return nwaves <= 2 * slots ? (L >= 64 ? 7 : 6) : 0; }
#endif

// the launch geometry of a step on an nx x ny slab: column strips, rows per strip, row strips
// (a last strip shorter than the ghost width joins its predecessor: the boundary strips of a
// slab must hold the ng rows the neighbour receives as its halo), and whether the first and
// the last strip can go first with the halo exchange beside the interior ones
// the rows-left board of the SIMD pairs for a launch whose wavefronts take turns (prio_duty > 0): 2^16 SIMD
// numbers (XCC, SE, SH, CU, SIMD fields of the hardware id) x 2 slots, zeroed once, tagged per launch
#if !defined(PYRO_WAVE_NO_FEEDBACK)
static int wave_prio_feedback(pyrohip_ctx *c, FP &P)
{
    P.prio_board = nullptr;
    if (P.prio_duty <= 0) return 0;
#if !defined(PYRO_EMU)
    PYRO_TRY(prio_board_acquire(c, &P.prio_board, &P.prio_tag));
#endif
    return 0;
}
#else
static int wave_prio_feedback(pyrohip_ctx *, FP &P) { P.prio_board = nullptr; return 0; }
#endif
struct WaveGeom { int ncb, L, nsb, overlap; };
static WaveGeom wave_geometry(int nx, int ny, int ng, int cus, int march_rows)
{
    WaveGeom w;
    w.ncb = (ny + WOUT - 1) / WOUT;
    w.L = wave_rows(nx, w.ncb, 4 * PYRO_WAVE_MINW * cus);      // slots: wavefronts resident at once
    if (march_rows > 0) w.L = march_rows < nx ? march_rows : nx;
    w.nsb = (nx + w.L - 1) / w.L;
    if (w.nsb > 1 && nx - (w.nsb - 1) * w.L < ng) w.nsb--;
    w.overlap = (w.nsb >= 3 && w.L >= ng) ? 1 : 0;
    return w;
}
// One-round launches: a SIMD that holds ONE wavefront gets little out of it (4096^2, 74 x 27 = 1998 strips on
// 2048 slots: the 50 wavefronts without a partner lived 638 us, the paired ones 535 each -- tools/wave_timeline.py
// -- and the launch lasts as long as its last wavefront).  So the first n_extra column strips are cut into
// nsb + 1 row strips instead of nsb: as many strips as slots.  Single domain only (a slab's strips are
// its protocol), not with a strip length the caller chose.
#if !defined(PYRO_WAVE_NO_EXTRA)
static int wave_extra_units(const WaveGeom &w, int nx, int slots, int march_rows)
{
    return march_rows > 0 ? 0 : wave_fill_extra(w.ncb, w.nsb, nx, slots);
}
#else
static int wave_extra_units(const WaveGeom &, int, int, int) { return 0; }
#endif
// Many-round launches: the slots empty over the last strip's life (16384^2: 1815 / 1319 / 737 / 486 of 2048 wavefronts
// alive 400 / 300 / 200 / 100 us before the end of an 8.4 ms launch, tools/wave_timeline.py: ~2.7 % of the launch, 3.5 % at
// 8192^2).  So the row strips of the last round's worth of units are cut in two, and every XCD's queue of units ends
// with its share of them (the kernel's unit numbering): the drain lasts a short strip's life.  Single domain,
// launches of >= 4 rounds.  -> number of short strips (0: none), *Ls their length,
// *nsb_long the long ones in front
#if !defined(PYRO_WAVE_NO_SHORT_TAIL)
static int wave_short_tail(const WaveGeom &w, int nx, int slots, int march_rows, int *Ls, int *nsb_long)
{
    (void)march_rows;
    if ((long)w.ncb * w.nsb < 4L * slots || w.L < 16) return 0;
    // (measured, Gcell/s with halves / thirds / quarters: 16384^2 (126-row strips) 31.95 / 32.00 / 32.12, 8192^2 (59 rows)
    // 30.10 / 30.07 / 29.99, 6144^2 28.4 / 28.35 / 28.07; without the tail 31.75 / 29.7 / 27.65; a region of half a
    // round of units 31.81 / 29.85 / 27.85, of a round and a half 31.97 / 30.02 / 28.20: short strips of about 30 rows)
#ifndef PYRO_WAVE_TAIL_DIV
#define PYRO_WAVE_TAIL_DIV (w.L >= 100 ? 4 : 2)
#endif
#ifndef PYRO_WAVE_TAIL_ROUNDS_X2
#define PYRO_WAVE_TAIL_ROUNDS_X2 2
#endif
    const int nr1 = (slots + w.ncb - 1) / w.ncb;           // row strips of one round of units
    if (w.nsb < 4 * nr1) return 0;
    const int nr = (nr1 * PYRO_WAVE_TAIL_ROUNDS_X2 + 1) / 2;
    const int nl = w.nsb - nr;
    const int rows = nx - nl * w.L;                        // rows of the short region (>= 1: nsb strips cover nx)
    const int ls = (w.L + PYRO_WAVE_TAIL_DIV - 1) / PYRO_WAVE_TAIL_DIV;
    int ns = (rows + ls - 1) / ls;
    if (ns > 1 && rows - (ns - 1) * ls < 4) ns--;          // (a last strip of < 4 rows joins its predecessor)
    if (rows < 8 || ns < 1) return 0;
    *Ls = ls; *nsb_long = nl;
    return ns;
}
#else
static int wave_short_tail(const WaveGeom &, int, int, int, int *, int *) { return 0; }
#endif
#if !PYRO_FAST
// (for callers that want to know before they launch: bench.py's scaling line, the tests of
// the decomposed runs)  out: ncb, L, nsb, overlap, wavefronts, resident slots
int comp_wave_geometry(int nx, int ny, int ng, int cus, int march_rows, int *out)
{
    const WaveGeom w = wave_geometry(nx, ny, ng, cus > 0 ? cus : 256, march_rows);
    out[0] = w.ncb; out[1] = w.L; out[2] = w.nsb; out[3] = w.overlap;
    out[4] = w.ncb * w.nsb; out[5] = 4 * PYRO_WAVE_MINW * (cus > 0 ? cus : 256);
    return 0;
}
#endif

int comp_step_wave_ex(pyrohip_state *s, const pyrohip_comp_params *p, double dt,
                      const StepScalars *S, const double **dmin_out)   // as comp_step_fused_ex
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Uout;
    PYRO_TRY(fused_prepare(s, p, dt, P, Uin, Uout, S == nullptr));
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    const WaveGeom wg = wave_geometry(g.nx, g.ny, g.ng, cus, p->march_rows);
    P.ncb = wg.ncb;
    P.L = wg.L;
    const int nsb = wg.nsb;
    P.nsb = nsb;
    // slab of a decomposed run with the halo communicator: EVERY step posts the
    // exchange of its new boundary rows (overlapped when the strips allow it), so the
    // protocol does not depend on this rank's geometry
    const bool post = s->nb_set && comm_can_overlap(s);
    P.n_extra = (s->nb_set || post) ? 0 : wave_extra_units(wg, g.nx, 4 * PYRO_WAVE_MINW * cus, p->march_rows);
    int nsb_long = nsb;
    if (!s->nb_set && !post && !(S && s->pol_next))
        P.n_short = wave_short_tail(wg, g.nx, 4 * PYRO_WAVE_MINW * cus, p->march_rows, &P.Ls, &nsb_long);
    if (P.n_short > 0) { P.nsb = nsb_long + P.n_short; P.units_short = P.ncb * P.n_short; }
#if !defined(PYRO_WAVE_NO_SLAB_TAIL)
    // (a slab with the boundary strips on the halo stream: the interior launch ends with short strips too -- the
    // row strips in front of the last boundary strip, which keeps its rows; strip lengths that divide only)
    int slab_short = 0;
    if (post && wg.overlap) {
        int ls = 0, nl1 = 0;
        const int ns1 = wave_short_tail(wg, g.nx, 4 * PYRO_WAVE_MINW * cus, 0, &ls, &nl1);
        const int div = ls > 0 ? (P.L + ls - 1) / ls : 0;
        const int nr = nsb - nl1;                          // (row strips of one round of units: wave_short_tail)
        if (ns1 > 0 && div >= 2 && P.L % div == 0 && nsb - nr - 1 >= 2) {
            P.Ls = P.L / div;
            P.n_short = nr * div;
            P.n_tail = 1;
            P.nsb = nsb + nr * (div - 1);
            slab_short = P.ncb * P.n_short;
        }
    }
#else
    const int slab_short = 0;
#endif
    const int nwg = P.ncb * P.nsb + P.n_extra;
    PYRO_TRY(c->reduce.ensure((nwg + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    using KernelT = void (*)(const double *, double *, Geom, FP, int *, double *,
                             const StepScalars *);
    static const KernelT kernels_src[3][2] = {
        {k_ctu_wave<0, false>, k_ctu_wave<0, true>},
        {k_ctu_wave<1, false>, k_ctu_wave<1, true>},
        {k_ctu_wave<2, false>, k_ctu_wave<2, true>}};
#if PYRO_FAST
    static const KernelT kernels_nosrc[3][2] = {
        {k_ctu_wave<0, false, false, false, false, false>, k_ctu_wave<0, true, false, false, false, false>},
        {k_ctu_wave<1, false, false, false, false, false>, k_ctu_wave<1, true, false, false, false, false>},
        {k_ctu_wave<2, false, false, false, false, false>, k_ctu_wave<2, true, false, false, false, false>}};
    const KernelT (*kernels)[2] = P.have_src ? kernels_src : kernels_nosrc;
#if !defined(PYRO_EMU) && !defined(PYRO_WAVE_NO_FEEDBACK)
    // ... with the rows-left board of the SIMD pairs (launches of one or two rounds)
    static const KernelT kernels_src_fb[3][2] = {
        {k_ctu_wave<0, false, false, false, false, true, -1, true>, k_ctu_wave<0, true, false, false, false, true, -1, true>},
        {k_ctu_wave<1, false, false, false, false, true, -1, true>, k_ctu_wave<1, true, false, false, false, true, -1, true>},
        {k_ctu_wave<2, false, false, false, false, true, -1, true>, k_ctu_wave<2, true, false, false, false, true, -1, true>}};
    static const KernelT kernels_nosrc_fb[3][2] = {
        {k_ctu_wave<0, false, false, false, false, false, -1, true>, k_ctu_wave<0, true, false, false, false, false, -1, true>},
        {k_ctu_wave<1, false, false, false, false, false, -1, true>, k_ctu_wave<1, true, false, false, false, false, -1, true>},
        {k_ctu_wave<2, false, false, false, false, false, -1, true>, k_ctu_wave<2, true, false, false, false, false, -1, true>}};
    const KernelT (*kernels_fb)[2] = P.have_src ? kernels_src_fb : kernels_nosrc_fb;
#else
    const KernelT (*kernels_fb)[2] = kernels;
#endif
#else
    // (the bit-faithful build keeps ONE instance: without the source blocks its register allocation
    // comes out worse -- 15.8 -> 16.4 ms per step at 16384^2)
    const KernelT (*kernels)[2] = kernels_src;
    const KernelT (*kernels_fb)[2] = kernels_src;      // (... and the turns by phase)
#endif
    const int solver = (p->riemann == 1 || p->riemann == 2) ? p->riemann : 0;
    const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
    if (post && wg.overlap) {
        // slab of a decomposed run: the first and the last strip of rows -- the
        // rows the neighbours need as their next halo -- are a launch of their own on the halo
        // stream (highest priority: dispatched first), their exchange is posted behind them
        // there, and the interior strips run BESIDE both on the context's stream.  (Round 5 ran
        // the two launches one after the other: the boundary launch held 586 of 2048 wavefront
        // slots for a whole strip time.)
        // old ghost frame -> new buffer, BEFORE the halos land in it (device-side stepping: the
        // fill before this step has written it already, comp_api.hip: k_fill_frame2)
        if (!s->frame_prefilled) fused_copy_frame(s);
        s->frame_prefilled = false;
        hipStream_t bs = nullptr;
        PYRO_TRY(comm_fork_boundary(s, &bs));
        P.sb_first = 0; P.sb_step = P.nsb - 1;
        P.nunits = 2 * P.ncb;
        P.units_short = 0;
        P.prio_duty = 0;               // (the pair of a SIMD may belong to the other launch)
        PYRO_LAUNCH_ON(c, bs, "k_ctu_wave_boundary", kernels[solver][std_rec], dim3(8 * ((P.nunits + 7) / 8)),
                       dim3(64), WLDS_BYTES, (const double *)Uin, Uout, g, P, s->d_flag, part, S);
        PYRO_TRY(comm_post_halo_here(s, Uout));
        P.sb_first = 1; P.sb_step = 1;
        P.nunits = (P.nsb - 2) * P.ncb;
        P.units_short = slab_short;
        P.prio_duty = wave_prio_duty(nwg, 4 * PYRO_WAVE_MINW * cus, P.L);
        const int nblk = slab_short > 0 ? 8 * ((P.nunits - slab_short + 7) / 8 + (slab_short + 7) / 8)
                                        : 8 * ((P.nunits + 7) / 8);
        PYRO_LAUNCH(c, "k_ctu_wave", kernels[solver][std_rec], dim3(nblk), dim3(64),
                    WLDS_BYTES, (const double *)Uin, Uout, g, P, s->d_flag, part, S);
        PYRO_TRY(comm_join_boundary(s));       // the minimum below reads the boundary strips' partials
        const double *dmin;
        PYRO_TRY(fused_tail(s, part, nwg, true, &dmin, S != nullptr));
        int rc = 0;
        if (S) { fused_swap(s); *dmin_out = dmin; }
        else rc = fused_sync(s, dmin);
        s->halo_pending = (rc == 0);
        return rc;
    }
    P.nunits = nwg;
    P.prio_duty = wave_prio_duty(nwg, 4 * PYRO_WAVE_MINW * cus, P.L);
    PYRO_TRY(wave_prio_feedback(c, P));
    if (S && s->pol_next && !post) {
        // this launch is the whole step (pyrohip_comp_evolve): ghost cells read through the
        // boundary rules, the dt policy of the next step run by the last wavefront to finish
        static const KernelT kernels_one[3][2] = {
            {k_ctu_wave<0, false, false, true>, k_ctu_wave<0, true, false, true>},
            {k_ctu_wave<1, false, false, true>, k_ctu_wave<1, true, false, true>},
            {k_ctu_wave<2, false, false, true>, k_ctu_wave<2, true, false, true>}};
        P.pol = s->pol_next;
        P.pol_m = s->pol_m;
        P.pol_pre = (s->pol_m == 0) ? 1 : 0;
        P.mr = bc_map(g.ilo, g.ihi, g.ng, s->bc[0], s->bc[1], true);
        P.mc = bc_map(g.jlo, g.jhi, g.ng, s->bc[2], s->bc[3], true);
        for (int n = 0; n < 4; n++)
            for (int sd = 0; sd < 4; sd++)
                if (s->bc[n * 4 + sd] == PYROHIP_BC_REFLECT_ODD) P.odd |= 1u << (4 * n + sd);
        PYRO_LAUNCH(c, "k_ctu_wave", kernels_one[solver][std_rec], dim3(8 * ((nwg + 7) / 8)), dim3(64),
                    WLDS_BYTES, (const double *)Uin, Uout, g, P, s->d_flag, part, S);
        PYRO_CHECK_HIP(hipGetLastError());
        s->frame_prefilled = false;
        fused_swap(s);
        *dmin_out = (const double *)(s->d_polmem + 3 * kPolSetWords);
        s->halo_pending = false;
        return 0;
    }
    // (short tail: the grid holds the eight queues of long units, then the eight of short ones)
    const int nblocks = P.n_short > 0 ? 8 * ((P.ncb * (P.nsb - P.n_short) + 7) / 8 + (P.ncb * P.n_short + 7) / 8)
                                      : 8 * ((nwg + 7) / 8);
    PYRO_LAUNCH(c, "k_ctu_wave", (P.prio_board ? kernels_fb : kernels)[solver][std_rec], dim3(nblocks),
                dim3(64), WLDS_BYTES, (const double *)Uin, Uout, g, P, s->d_flag, part, S);
    if (post) {        // too few strips to overlap: the exchange follows the whole update
        if (!s->frame_prefilled) fused_copy_frame(s);
        PYRO_TRY(comm_post_halo(s, Uout));
    }
    const double *dmin;
    // (device-side stepping: the fill before this step may have written the new buffer's ghost
    // frame already, comp_api.hip: k_fill_frame2)
    const bool frame_done = post || s->frame_prefilled;
    s->frame_prefilled = false;
    PYRO_TRY(fused_tail(s, part, nwg, frame_done, &dmin, S != nullptr));
    if (S) { fused_swap(s); *dmin_out = dmin; s->halo_pending = post; return 0; }
    const int rc = fused_sync(s, dmin);
    s->halo_pending = post && rc == 0;
    return rc;
}

int comp_step_wave(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    return comp_step_wave_ex(s, p, dt, nullptr, nullptr);
}

// compressible_rk: k of the stage state y into slot `slot` of the k state, ONE launch of the
// row-marching kernel's method-of-lines instance (no sponge: comp_rk_rhs of compressible.hip)
int comp_rk_rhs_wave(pyrohip_state *s, const pyrohip_comp_params *p, pyrohip_state *kst, int slot)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Uout;
    PYRO_TRY(fused_prepare(s, p, 1.0, P, Uin, Uout, true, false));
    Uout = kst->d + (size_t)(4 * slot) * g.plane;
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    const WaveGeom wg = wave_geometry(g.nx, g.ny, g.ng, cus, p->march_rows);
    P.ncb = wg.ncb; P.L = wg.L; P.nsb = wg.nsb;
    P.n_extra = s->nb_set ? 0 : wave_extra_units(wg, g.nx, 4 * PYRO_WAVE_MINW * cus, p->march_rows);
    const int nwg = P.ncb * wg.nsb + P.n_extra;
    PYRO_TRY(c->reduce.ensure((nwg + kMinStageBlocks + 2) * sizeof(double)));
    using KernelT = void (*)(const double *, double *, Geom, FP, int *, double *,
                             const StepScalars *);
    static const KernelT kernels[3][2] = {
        {k_ctu_wave<0, false, true>, k_ctu_wave<0, true, true>},
        {k_ctu_wave<1, false, true>, k_ctu_wave<1, true, true>},
        {k_ctu_wave<2, false, true>, k_ctu_wave<2, true, true>}};
    const int solver = (p->riemann == 1 || p->riemann == 2) ? p->riemann : 0;
    const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
    P.nunits = nwg;
    P.prio_duty = wave_prio_duty(nwg, 4 * PYRO_WAVE_MINW * cus, 0);
    // (no rows-left board for the method-of-lines launches: measured, an RK4 step at 4096^2 2.20 ms with it, 2.14 without)
    PYRO_LAUNCH(c, "k_ctu_wave_mol", kernels[solver][std_rec], dim3(8 * ((nwg + 7) / 8)), dim3(64), WLDS_BYTES,
                (const double *)Uin, Uout, g, P, s->d_flag, (double *)c->reduce.p, nullptr);
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

// compressible_rk: the WHOLE Runge-Kutta step of compressible_rk/simulation.py:58-104 on the row-
// marching kernel -- stage 0 on the filled state (the driver's fill + the stage's own, one fill:
// idempotent), every later stage with its start y_0 + dt sum_j a_sj k_j built at load and its
// ghost cells read through the boundary rules (k_ctu_wave<.., MOL, false, RKF>), the last one
// storing y_0 + dt sum_s b_s k_s into the second buffer (ghost frame carried over, buffers
// swapped) and leaving the CFL minimum of the new state.  No pyrohip_state_lincomb launches, no
// stage state, no ghost fills of it: an RK4 step at 4096^2 was 4 x 0.49 ms of right-hand sides
// + 4 x 0.44 ms of linear combinations + 0.14 ms of CFL reduction (profiles/r05_rk4096_*).
// a: nstages x nstages, row-major (Butcher tableau, strictly lower triangular); b: nstages.
// S == nullptr: dt by value, flag and minimum read back (single step); S != nullptr: a step of a
// device-side run (pyrohip_comp_rk_evolve), dt from *S, *dmin_out = device address of the minimum.
int comp_rk_step_wave(pyrohip_state *s, const pyrohip_comp_params *p, pyrohip_state *kst, int nstages,
                      const double *a, const double *b, double dt, const StepScalars *S,
                      const double **dmin_out)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    FP P;
    double *Uin, *Unew;
    PYRO_TRY(fused_prepare(s, p, dt, P, Uin, Unew, S == nullptr, true));
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    const WaveGeom wg = wave_geometry(g.nx, g.ny, g.ng, cus, p->march_rows);
    P.ncb = wg.ncb; P.L = wg.L; P.nsb = wg.nsb;
    P.n_extra = s->nb_set ? 0 : wave_extra_units(wg, g.nx, 4 * PYRO_WAVE_MINW * cus, p->march_rows);
    const int nwg = P.ncb * wg.nsb + P.n_extra;
    PYRO_TRY(c->reduce.ensure((nwg + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    using KernelT = void (*)(const double *, double *, Geom, FP, int *, double *, const StepScalars *);
    static const KernelT first[3][2] = {
        {k_ctu_wave<0, false, true>, k_ctu_wave<0, true, true>},
        {k_ctu_wave<1, false, true>, k_ctu_wave<1, true, true>},
        {k_ctu_wave<2, false, true>, k_ctu_wave<2, true, true>}};
#if PYRO_FAST && !defined(PYRO_EMU) && !defined(PYRO_RK_ONE_INSTANCE)
    static const KernelT later_mid[3][2] = {
        {k_ctu_wave<0, false, true, false, true, true, 0>, k_ctu_wave<0, true, true, false, true, true, 0>},
        {k_ctu_wave<1, false, true, false, true, true, 0>, k_ctu_wave<1, true, true, false, true, true, 0>},
        {k_ctu_wave<2, false, true, false, true, true, 0>, k_ctu_wave<2, true, true, false, true, true, 0>}};
    static const KernelT later_fin[3][2] = {
        {k_ctu_wave<0, false, true, false, true, true, 1>, k_ctu_wave<0, true, true, false, true, true, 1>},
        {k_ctu_wave<1, false, true, false, true, true, 1>, k_ctu_wave<1, true, true, false, true, true, 1>},
        {k_ctu_wave<2, false, true, false, true, true, 1>, k_ctu_wave<2, true, true, false, true, true, 1>}};
#else
    static const KernelT later_mid[3][2] = {
        {k_ctu_wave<0, false, true, false, true>, k_ctu_wave<0, true, true, false, true>},
        {k_ctu_wave<1, false, true, false, true>, k_ctu_wave<1, true, true, false, true>},
        {k_ctu_wave<2, false, true, false, true>, k_ctu_wave<2, true, true, false, true>}};
    const KernelT (*later_fin)[2] = later_mid;
#endif
    const int solver = (p->riemann == 1 || p->riemann == 2) ? p->riemann : 0;
    const int std_rec = (p->limiter == 2 && p->use_flattening) ? 1 : 0;
    P.nunits = nwg;
    P.prio_duty = wave_prio_duty(nwg, 4 * PYRO_WAVE_MINW * cus, 0);
    // (no rows-left board for the method-of-lines launches: measured, an RK4 step at 4096^2 2.20 ms with it, 2.14 without)
    const dim3 grid(8 * ((nwg + 7) / 8)), block(64);
    // stage 0: the state itself, ghost cells filled in memory (they stay the state's "stale"
    // ghost cells after the step, like the reference's), density floor in place
    // (a device-side run has filled the frames of both buffers with the launch of its dt policy:
    // comp_api.hip: k_fill_frame2_policy)
    const bool prefilled = s->frame_prefilled;
    s->frame_prefilled = false;
    if (!prefilled) PYRO_TRY(pyrohip_fill_bc(s, -1));
    PYRO_LAUNCH(c, "k_ctu_wave_mol", first[solver][std_rec], grid, block, WLDS_BYTES, (const double *)Uin,
                kst->d, g, P, s->d_flag, part, S);
    P.mr = bc_map(g.ilo, g.ihi, g.ng, s->bc[0], s->bc[1], true);
    P.mc = bc_map(g.jlo, g.jhi, g.ng, s->bc[2], s->bc[3], true);
    for (int n = 0; n < 4; n++)
        for (int sd = 0; sd < 4; sd++)
            if (s->bc[n * 4 + sd] == PYROHIP_BC_REFLECT_ODD) P.odd |= 1u << (4 * n + sd);
    P.rk_k = kst->d;
    for (int st = 1; st < nstages; st++) {
        P.rk_n = st;
        for (int j = 0; j < 3; j++) P.rk_a[j] = (j < st) ? a[st * nstages + j] : 0.0;
        P.rk_final = (st == nstages - 1) ? 1 : 0;
        P.rk_nb = nstages;
        for (int j = 0; j < 4; j++) P.rk_b[j] = (j < nstages) ? b[j] : 0.0;
        P.rk_out = Unew;
        PYRO_LAUNCH(c, "k_ctu_wave_rk", (P.rk_final ? later_fin : later_mid)[solver][std_rec], grid, block, WLDS_BYTES, (const double *)Uin,
                    kst->d + (size_t)(4 * st) * g.plane, g, P, s->d_flag, part, S);
    }
    PYRO_CHECK_HIP(hipGetLastError());
    if (!prefilled) fused_copy_frame(s);  // ghost frame of the state -> the new buffer
    s->cfl_is_global = false;
    if (S && nwg <= 128 * kMinStageBlocks) {
        // device-side run: the next policy launch takes the minimum of the partials itself (fused_tail)
        s->pend_part = part;
        s->pend_n = nwg;
        fused_swap(s);
        *dmin_out = part + nwg + kMinStageBlocks;
        return 0;
    }
    const double *dmin = launch_min_reduce(c->stream, part, nwg);
    if (S) { fused_swap(s); *dmin_out = dmin; return 0; }
    const int rc = fused_sync(s, dmin);   // flag + minimum read back; swap if the state was valid
    if (rc == 0) s->cfl_kind = 1;         // (the minimum is compressible_rk's CFL quantity)
    return rc;
}

}  // namespace PYRO_NS
}  // namespace pyro

#if defined(PYRO_WAVE_TIMELINE) && PYRO_FAST && !defined(PYRO_EMU)
extern "C" int pyrohip_debug_wave_timeline(unsigned long long *out, int nunits)
{
    if (!out || nunits < 1 || nunits > 65536) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pyro::PYRO_NS::g_wave_timeline),
                                    (size_t)nunits * 4 * sizeof(unsigned long long));
}
#endif
