// Row-marching, time-skewed red-black Gauss-Seidel smoother (MG.py:560-600:
// CellCenterMG2d.smooth) for the levels that do not fit the caches.
//
// The tile / band smoothers (multigrid.hip) stage a 64 x 128 region, make 2K colour
// sweeps over it and write the tile back: K = 5 needs two passes over the level for
// the ten iterations of a V-cycle leg, and staging does not overlap the sweeps
// (169 us per launch at 4096^2, 2.4 TB/s; profiles/r02g_mg4096_kernel_stats.csv).
// Here ONE wavefront owns a strip of 128 columns (lane h: columns 2h, 2h + 1) and
// marches up the rows with a rolling window of NP + 2 rows in registers, NP = 2K:
// in step k it loads row k and runs sweep s = 1 .. NP on row k - s, in that order.
// When sweep s reaches row r, row r + 1 has just had sweep s - 1 (same step) and row
// r - 1 had sweep s one step earlier but not yet sweep s + 1 -- and sweep s touched
// only the cells of the class being relaxed, so the cells of the other class it reads
// are exactly those the level-wide sweep s - 1 left: the values are those of 2K whole
// colour sweeps with the ghost cells refilled after each (MG.py:598-599).  Row k - NP
// is final after the step and is stored.  All ten iterations in one pass over v and
// f, loads (PF rows ahead) and stores running under the arithmetic.
//
// Column neighbours come from the neighbouring lane (whole-wave DPP rotation) or are
// the thread's other cell; no LDS, no barrier.  As in the band kernel no ghost cell is
// kept: a cell next to a physical boundary takes ghost_h(its own value), every cell of
// the window is relaxed in every sweep, and what the cells beyond the still-valid part
// of the apron (2K columns / rows around the part a wavefront stores) compute is never
// read by a cell that is stored.  The level is cut into column strips x row chunks;
// wavefronts whose part touches no physical boundary run a copy of the loop without
// the boundary selects (the selects would be a third of its instructions).
#include "mg_march.h"
#include "stencil.h"
#include <type_traits>

#ifndef MGM_SKIP
#define MGM_SKIP 1     // no steps past the last one a stored row depends on (mgm_march)
#endif

namespace pyro {
namespace {

#if !defined(PYRO_EMU)
template <int CTRL> __device__ __forceinline__ double mgm_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_lower(double v) { return mgm_dpp<0x13C>(v); }   // wave_ror:1
__device__ __forceinline__ double from_upper(double v) { return mgm_dpp<0x134>(v); }   // wave_rol:1
#else
__device__ __forceinline__ double from_lower(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double from_upper(double v) { return __shfl_down(v, 1, 64); }
#endif

// a / b with rb = RN(1 / b): the correctly rounded quotient (multigrid.hip: div_by)
__device__ __forceinline__ double quot(double a, double b, double rb)
{
    const double q = a * rb;
    const double e = fma(-b, q, a);
    return fma(e, rb, q);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int N, int U = 0, class F> __device__ __forceinline__ void static_for(F &&fn)
{
    if constexpr (U < N) {
        fn(std::integral_constant<int, U>{});
        static_for<N, U + 1>(fn);
    }
}

// the part of the level a wavefront works on
struct Part {
    int tj0, tj1, ra, rb;   // columns / rows it stores
    int gj0, g0;            // first column / row of its window (both even, unwrapped)
    int nload;              // rows it loads: g0 ... g0 + nload - 1
    bool plo_i, phi_i, plo_j, phi_j;   // its window reaches the physical boundary there
};

__device__ __forceinline__ Part mgm_part(const MGMarch &A, int NP, int W, int tail)
{
    Part P;
    const int n = A.n;
    const bool per_i = (A.code[0] == PYROHIP_BC_PERIODIC), per_j = (A.code[2] == PYROHIP_BC_PERIODIC);
    const int nside = A.nchunks_side > 0 ? 2 : 0, nin = A.ncs - nside;
    const int b = xcd_tile(blockIdx.x, nin * A.nchunks + nside * A.nchunks_side);
    int cs, ch, CR;
    if (b < nin * A.nchunks) { cs = nside / 2 + b % nin; ch = b / nin; CR = A.CR; }
    else {
        const int e = b - nin * A.nchunks;
        cs = (e & 1) ? A.ncs - 1 : 0; ch = e >> 1; CR = A.CR_side;
    }
    P.tj0 = 1 + cs * A.TJ; P.tj1 = min(P.tj0 + A.TJ - 1, n);
    P.ra = A.row0 + ch * CR; P.rb = min(P.ra + CR - 1, A.row1);
    // (a tail needs row rb + 1 final too; row ra - 1 and the columns beside the strip are:
    // ra and tj0 are odd, the even starts below add a row / a column, and the strip's 128
    // columns leave NP + 1 on the right as well)
    // (tail 2 rides on a launch that prolongs: the correction added to the window's first and
    // last column takes a coarse value from beyond the window -- those columns start wrong,
    // one sweep earlier than the rotation alone makes them; two more columns of apron on
    // either side, the strips of such a launch store four columns less: mgm_tj)
    int gj0 = P.tj0 - NP - (tail == 2 ? 2 : 0), g0 = P.ra - NP, gend = P.rb + NP + (tail ? 1 : 0);
    if (!per_j) gj0 = max(gj0, 0);
    if (!per_i) { g0 = max(g0, 0); gend = min(gend, n); }
    // even starts: the class a sweep relaxes in a row / lane is then known at compile time
    gj0 -= (gj0 & 1); g0 -= (g0 & 1);
    P.plo_i = !per_i && g0 == 0; P.phi_i = !per_i && gend == n;
    // a part that ends at the top boundary starts a few rows lower, so that row n is a
    // multiple of W rows above its first row (mgm_march; never a part that starts at the
    // bottom one: mg_march_usable)
    if (P.phi_i && !P.plo_i) g0 -= (W - (n - g0) % W) % W;
    P.gj0 = gj0; P.g0 = g0; P.nload = gend - g0 + 1;
    P.plo_j = !per_j && gj0 == 0; P.phi_j = !per_j && gj0 + MGM_COLS - 1 >= n;
    return P;
}

// sign of a homogeneous ghost cell relative to its mirror cell (PYROHIP_BC_CONST, value 0,
// does not come here: mg_march_usable)
__device__ __forceinline__ double ghost_sign(int code) { return code == PYROHIP_BC_REFLECT_ODD ? -1.0 : 1.0; }

// EDGEI: the level has physical boundaries below / above; EDGEJ: this wavefront's strip
// reaches a physical boundary left / right
template <int NP, int PF, bool POW2, bool PROL, bool EDGEI, bool EDGEJ, int TAIL>
__device__ __forceinline__ void mgm_march(const MGMarch &A, const Part &P, double (*fst)[64])
{
    constexpr int W = NP + 2 + PF;      // window: rows k + PF (in flight) ... k - NP - 1
    static_assert(W % 2 == 0 && PF >= 1, "the unrolled block must keep the row parity");
    const int n = A.n, ln = threadIdx.x & 63;
    const unsigned pitch = (unsigned)A.pitch;
    const bool per_i = (A.code[0] == PYROHIP_BC_PERIODIC), per_j = (A.code[2] == PYROHIP_BC_PERIODIC);
    const int c0 = A.code[0], c1 = A.code[1], c2 = A.code[2], c3 = A.code[3];
    const int g0 = P.g0, gj0 = P.gj0;

    // the thread's columns: array column to load from (wrapped / clamped), may it store
    unsigned gjw[2];
    bool st[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int g = gj0 + 2 * ln + q;
        int w = g + (g < 1 ? n : 0);
        w -= (w > n ? n : 0);
        gjw[q] = (unsigned)(per_j ? w : min(g, n + 1));
        st[q] = g >= P.tj0 && g <= P.tj1;
    }
    // Physical sides.  gj0 is even and so is n: column 1 is some lane's cell 1, column n some
    // lane's cell 0, and the sweeps that relax them are the Q = 1 / Q = 0 steps.  There the
    // cell takes s * (its own value) for the ghost neighbour, s = -1 (Dirichlet) or +1:
    // x + s * me in one fma rounds like the reference's x + ghost.  One select per update
    // in the strips at the sides, none elsewhere.
    const bool isW1 = EDGEJ && P.plo_j && gj0 + 2 * ln + 1 == 1;
    const bool isE0 = EDGEJ && P.phi_j && gj0 + 2 * ln == n;
    const double sW = isW1 ? ghost_sign(c2) : 1.0, sE = isE0 ? ghost_sign(c3) : 1.0;
    // below / above: the row next to the boundary has march index 1 (bottom, g0 = 0) or a
    // multiple of W (top: mgm_part aligns g0), i.e. it meets sweep s in the steps U = s + 1
    // / U = s of one block of the unrolled loop -- 2 NP places known at compile time
    const double sB = ghost_sign(c0), sT = ghost_sign(c1);
    const int kTop = n - g0;

    double v[W][2], f[W][2];
#pragma unroll
    for (int r = 0; r < W; r++) { v[r][0] = v[r][1] = 0.0; f[r][0] = f[r][1] = 0.0; }
    // The diagnostics' tail on top of the prolongation does not fit 256 registers: there the
    // older half of the right-hand side's window (rows k - FA and older: twelve rows) lives
    // in LDS, a slot per thread, written once when a row reaches that age and read once per
    // step and row by the sweep that needs it.
    constexpr bool FST = (TAIL == 2);
    constexpr int FA = NP / 2, FR = W / 2;
    static_assert(!FST || (NP + 1 - FA < FR && W % FR == 0), "the stashed rows must fit the ring");
    // (indices and ages are compile-time constants once the loops are unrolled)
    auto f_at = [&](int S, int age, int q) __attribute__((always_inline)) -> double {
        return (FST && age >= FA) ? fst[(S % FR) * 2 + q][ln] : f[S][q];
    };

    auto row_of = [&](int g) -> unsigned {      // array row of window row g (unwrapped)
        const int w = g + (g < 1 ? n : 0) - (g > n ? n : 0);
        return (unsigned)(per_i ? w : min(max(g, 0), n + 1));
    };
    // no branch around the loads (rows beyond the last one load it again: cache hits): the
    // compiler's s_waitcnt bookkeeping gives up the distance of the prefetch at control flow
    const double *src = A.vin_zero ? A.f : A.vin;   // zero start: a second (cached) read of f, dropped
    const bool vz = A.vin_zero != 0;
    auto load_row = [&](auto sc, int kk) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        const unsigned base = row_of(g0 + min(kk, P.nload - 1)) * pitch;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double x = src[base + gjw[q]];
            f[S][q] = A.f[base + gjw[q]];
            v[S][q] = vz ? 0.0 : x;
        }
    };

    // ---- coarse rows for the prolongation (patch.py:678-736): rows CI - 1, CI, CI + 1 of
    // the coarse solution at the thread's two coarse columns, CI = coarse row under the
    // current fine row, plus the next row in flight ----
    double cw[3][2] = {{0, 0}, {0, 0}, {0, 0}}, cpre[2] = {0, 0};
    unsigned cjw[2] = {0, 0};
    const int nc = n >> 1;
    const unsigned cpitch = (unsigned)A.cpitch;
    auto crow_of = [&](int ci) -> unsigned {
        const int w = ci + (ci < 1 ? nc : 0) - (ci > nc ? nc : 0);
        return (unsigned)(per_i ? w : min(max(ci, 0), nc + 1));
    };
    auto load_crow = [&](double (&dst)[2], int ci) __attribute__((always_inline)) {
        const unsigned base = crow_of(ci) * cpitch;
        dst[0] = A.cv[base + cjw[0]];
        dst[1] = A.cv[base + cjw[1]];
    };
    if (PROL) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int cj = (gj0 >> 1) + ln + q;       // coarse column under fine column gj0 + 2 ln + q
            int w = cj + (cj < 1 ? nc : 0);
            w -= (w > nc ? nc : 0);
            w -= (w > nc ? nc : 0);
            cjw[q] = (unsigned)(per_j ? w : min(max(cj, 0), nc + 1));
        }
        const int ci = g0 >> 1;                       // under fine row g0 (even)
        load_crow(cw[0], ci - 1); load_crow(cw[1], ci); load_crow(cw[2], ci + 1);
        load_crow(cpre, ci + 2);
    }

    // ---- tail state: the row below the one whose residual is taken (its window slot is
    // loaded over by then), the odd row's residuals (restriction), the old solution's row
    // one step ahead and the two sums (diagnostics) ----
    double vA[2] = {0, 0}, rodd[2] = {0, 0}, sres = 0.0;
    const double srel = 0.0;     // (relative_error: once, after the solve's last cycle -- multigrid.hip)

    const bool col_ghosts = P.tj0 == 1 || P.tj1 == n;   // the strip stores column 1 or n
    // rows 0 .. PF - 1 on their way
    static_for<PF>([&](auto sc) __attribute__((always_inline)) { load_row(sc, decltype(sc)::value); });

    // The last step that matters.  Step k sweeps the rows k - 1 ... k - NP; a stored row depends
    // on sweep s of the rows within NP - s of it, so beyond k = (last row that is stored -- or,
    // with a tail, must be final) + NP no sweep touches anything a stored row reads: the NP apron
    // rows above a part are loaded for the sweeps of the rows below them and need no steps of
    // their own (round 3 ran nload + NP steps: 20 of 159 for nothing on a 98-row chunk).
    // (Skipping the apron rows' late sweeps one by one as well -- a second copy of the unrolled
    // block with a scalar compare and branch per sweep for the first and last blocks -- was built
    // and measured: 707 -> 1018 us per 4096^2 V-cycle; two 35 KB copies do not fit the
    // instruction cache.)
    const int sk_hi = P.rb + (TAIL ? 1 : 0) - g0 + NP;
    auto step = [&](auto uc, int k0) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        const int k = k0 + U;
        const bool botBlk = EDGEI && P.plo_i && k0 == 0, topBlk = EDGEI && P.phi_i && k0 == kTop;
        // (1) row k + PF on its way
        load_row(std::integral_constant<int, (U + PF) % W>{}, k + PF);
        // (2) row k enters: scale f (exact, see mg_pow2), add the prolonged correction
        {
            constexpr int S = U % W;
            if (POW2) { f[S][0] *= A.rdenom; f[S][1] *= A.rdenom; }
            if (PROL) {
                if (U & 1) {       // fine row g0 + k is odd: the first of its pair, next coarse row
                    cw[0][0] = cw[1][0]; cw[0][1] = cw[1][1];
                    cw[1][0] = cw[2][0]; cw[1][1] = cw[2][1];
                    cw[2][0] = cpre[0]; cw[2][1] = cpre[1];
                    load_crow(cpre, ((g0 + k + 1) >> 1) + 2);
                }
                // k_mg_prolong_add's expression for fine cell (fi, fj) = (row - 1, column - 1):
                // (q0 -+ 0.25 m_x) -+ 0.25 m_y, + for odd fi / fj; 0.25 (0.5 d) == 0.125 d
                const double west = from_lower(cw[1][0]), east = from_upper(cw[1][1]);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double tx = 0.125 * (cw[2][q] - cw[0][q]);
                    const double ty = 0.125 * (q ? east - cw[1][0] : cw[1][1] - west);
                    const double a = (U & 1) ? cw[1][q] - tx : cw[1][q] + tx;   // fi = g0 + k - 1
                    const double e = q ? a - ty : a + ty;                       // fj = gj0 + 2 ln + q - 1
                    v[S][q] += e;
                }
            }
        }
        if constexpr (FST) {
            constexpr int SF = (U - FA + 2 * W) % W;
            fst[(SF % FR) * 2][ln] = f[SF][0];
            fst[(SF % FR) * 2 + 1][ln] = f[SF][1];
        }
        // (3) sweep s on row k - s; its cells of the class being relaxed are the thread's
        // column (k + 1) & 1 in every one of these rows
        constexpr int Q = (U + 1) & 1;
#pragma unroll
        for (int s = 1; s <= NP; s++) {
            const int SR = (U - s + 2 * W) % W, SU = (SR + 1) % W, SD = (SR + W - 1) % W;
            const double fq = f_at(SR, s, Q);
            const double me = v[SR][Q];
            const double up = v[SU][Q], dn = v[SD][Q];
            const double own = v[SR][Q ^ 1];
            const double oth = Q ? from_upper(v[SR][0]) : from_lower(v[SR][1]);
            const double e = Q ? oth : own, w = Q ? own : oth;
            double sj, si;                       // e + w, up + dn with the ghost cells' values
            if (!EDGEJ) sj = e + w;
            else if (Q) sj = fma(isW1 ? me : w, sW, e);
            else sj = fma(isE0 ? me : e, sE, w);
            if (EDGEI && U - s == 1 && botBlk) si = fma(me, sB, up);
            else if (EDGEI && U == s && topBlk) si = fma(me, sT, dn);
            else si = up + dn;
            if (POW2)
                v[SR][Q] = fma(A.ky, sj, fma(A.kx, si, fq));
            else
                v[SR][Q] = quot(fq + A.xc * si + A.yc * sj, A.denom, A.rdenom);
        }
        // (4) row k - NP is final
        {
            constexpr int S = (U - NP + 2 * W) % W;
            const int gi = g0 + k - NP;
            if (gi >= P.ra && gi <= P.rb) {
                const unsigned row = (unsigned)gi * pitch;
#pragma unroll
                for (int q = 0; q < 2; q++)
                    if (st[q]) A.vout[row + (unsigned)(gj0 + 2 * ln + q)] = v[S][q];
                // edge ghosts stay current for the other kernels: mirror values on physical
                // sides, copies on periodic ones
                if (__builtin_expect(gi == 1 || gi == n, 0)) {
                    const unsigned grow = ((gi == 1) == per_i) ? (unsigned)(n + 1) * pitch : 0u;
                    const double sg = per_i ? 1.0 : ghost_sign(gi == 1 ? c0 : c1);
#pragma unroll
                    for (int q = 0; q < 2; q++)
                        if (st[q]) A.vout[grow + (unsigned)(gj0 + 2 * ln + q)] = sg * v[S][q];
                }
                if (__builtin_expect(col_ghosts, 0)) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int gj = gj0 + 2 * ln + q;
                        if (!st[q] || (gj != 1 && gj != n)) continue;
                        const unsigned gcol = ((gj == 1) == per_j) ? (unsigned)(n + 1) : 0u;
                        const double sg = per_j ? 1.0 : ghost_sign(gj == 1 ? c2 : c3);
                        A.vout[row + gcol] = sg * v[S][q];
                    }
                }
            }
        }
        // (5) the tail: row k - NP - 1 has been final since the previous step, now the rows
        // below (vA) and above it are too: its residual, k_mg_residual's expression (the
        // ghost cells' values are the mirror values the stores above keep current)
        if constexpr (TAIL != 0) {
            constexpr int SC = (U - NP + 2 * W) % W, SB = (U - NP - 1 + 2 * W) % W;
            const int gi = g0 + k - NP - 1;
            if (gi >= P.ra && gi <= P.rb) {
                const bool bot = EDGEI && !per_i && gi == 1, top = EDGEI && !per_i && gi == n;
                const double fromW = from_lower(v[SB][1]), fromE = from_upper(v[SB][0]);
                double rr[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const double me = v[SB][q];
                    const double dn = bot ? sB * me : vA[q], up = top ? sT * me : v[SC][q];
                    double w = q ? v[SB][0] : fromW, e = q ? fromE : v[SB][1];
                    if (EDGEJ && q == 1 && isW1) w = sW * me;
                    if (EDGEJ && q == 0 && isE0) e = sE * me;
                    const double fs = f_at(SB, NP + 1, q);
                    const double fo = POW2 ? fs * A.denom : fs;
                    rr[q] = fo - A.alpha * me +
                            A.beta * (quot(dn + up - 2 * me, A.dx2, A.rdx2) +
                                      quot(w + e - 2 * me, A.dx2, A.rdx2));
                }
                if constexpr (TAIL == 1) {
                    // fine rows 2 ci - 1, 2 ci and columns 2 cj - 1, 2 cj make coarse cell (ci, cj):
                    // the thread's odd column and the next thread's even one, summed in
                    // k_mg_restrict's order (r00 + r10 + r01 + r11)
                    if constexpr ((U + 1) & 1) { rodd[0] = rr[0]; rodd[1] = rr[1]; }   // gi is odd
                    else {
                        const double r01 = from_upper(rodd[0]), r11 = from_upper(rr[0]);
                        const double c = 0.25 * (rodd[1] + rr[1] + r01 + r11);
                        if (st[1])
                            A.cf[(unsigned)(gi >> 1) * (unsigned)A.cfpitch + (unsigned)((gj0 >> 1) + ln + 1)] = c;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2; q++)
                        if (st[q]) sres += rr[q] * rr[q];
                }
            }
            vA[0] = v[SB][0]; vA[1] = v[SB][1];
        }
    };

#if MGM_SKIP
    const int nsteps = (P.nload + NP < sk_hi + 1) ? P.nload + NP : sk_hi + 1;
#else
    const int nsteps = P.nload + NP;
#endif
    for (int k0 = 0; k0 < nsteps; k0 += W)
        static_for<W>([&](auto uc) __attribute__((always_inline)) { step(uc, k0); });
    if constexpr (TAIL == 2) {
        sres = wave_sum(sres);
        if (ln == 0) { A.partial[blockIdx.x] = srel; A.partial[gridDim.x + blockIdx.x] = sres; }
    }
}

// (Neighbouring strips as one workgroup of two or four wavefronts kept in step by a barrier
// every step / every eight steps, so that the 42 columns two strips share would be fetched
// once: 780 -> 860-920 us per 4096^2 V-cycle, every marching launch 10-40 % slower.  The
// strips stay independent wavefronts; the launch reads 2.3 times the level.)
// (The two wavefronts of a SIMD taking turns at s_setprio, which shortens the single-round
// launches of the compressible kernel by 8 %, was measured here too: 851 -> 1070 us per
// 4096^2 V-cycle.  This kernel waits for memory, not for issue slots: arbitration by age.)
template <int NP, int PF, bool POW2, bool PROL, bool ANYEDGE, int TAIL = 0>
__global__ __launch_bounds__(64, 2) void k_mg_smooth_march(MGMarch A)
{
    const Part P = mgm_part(A, NP, NP + 2 + PF, TAIL);
    __shared__ double fst[TAIL == 2 ? NP + 2 + PF : 1][64];   // mgm_march: FST
    if (ANYEDGE && (P.plo_j || P.phi_j))
        mgm_march<NP, PF, POW2, PROL, ANYEDGE, true, TAIL>(A, P, fst);
    else
        mgm_march<NP, PF, POW2, PROL, ANYEDGE, false, TAIL>(A, P, fst);
}

}  // namespace

// The file is compiled three times (build.py: MGM_UNIT 0, 1, 2 -- the instances without a
// tail and the host side, those with the restriction, those with the diagnostics): one unit
// with all sixteen instances takes minutes.
#ifndef MGM_UNIT
#define MGM_UNIT 0
#endif
using KernT = void (*)(MGMarch);
constexpr int MGM_NP = 20;
static bool mgm_edge(const MGMarch &A)
{
    return A.code[0] != PYROHIP_BC_PERIODIC || A.code[2] != PYROHIP_BC_PERIODIC;
}

#if MGM_UNIT == 1
// the restriction rides on a down-leg launch (no prolongation)
int mg_march_launch_tail1(pyrohip_ctx *c, MGMarch &A, bool pow2)
{
    constexpr int NP = MGM_NP, PF = MGM_PF;
    static const KernT tail1[2][2] = {{k_mg_smooth_march<NP, PF, false, false, false, 1>, k_mg_smooth_march<NP, PF, false, false, true, 1>},
                                      {k_mg_smooth_march<NP, PF, true, false, false, 1>, k_mg_smooth_march<NP, PF, true, false, true, 1>}};
    PYRO_LAUNCH(c, "k_mg_smooth_march", tail1[pow2 ? 1 : 0][mgm_edge(A) ? 1 : 0], dim3(mg_march_blocks(A)), dim3(64), 0, A);
    return 0;
}
#elif MGM_UNIT == 2
// the solve diagnostics on the last up-leg launch (with the prolongation)
int mg_march_launch_tail2(pyrohip_ctx *c, MGMarch &A, bool pow2)
{
    constexpr int NP = MGM_NP, PF = MGM_PF;
    static const KernT tail2[2][2] = {{k_mg_smooth_march<NP, PF, false, true, false, 2>, k_mg_smooth_march<NP, PF, false, true, true, 2>},
                                      {k_mg_smooth_march<NP, PF, true, true, false, 2>, k_mg_smooth_march<NP, PF, true, true, true, 2>}};
    PYRO_LAUNCH(c, "k_mg_smooth_march", tail2[pow2 ? 1 : 0][mgm_edge(A) ? 1 : 0], dim3(mg_march_blocks(A)), dim3(64), 0, A);
    return 0;
}
#else
template <int K> static int launch_k(pyrohip_ctx *c, MGMarch &A, bool pow2)
{
    constexpr int NP = 2 * K, PF = MGM_PF;
    static_assert(NP == MGM_NP, "the tails' units are built for ten iterations");
    const bool edge = mgm_edge(A);
#define MGM_ROW(P2, PR) {k_mg_smooth_march<NP, PF, P2, PR, false>, k_mg_smooth_march<NP, PF, P2, PR, true>}
    static const KernT inst[2][2][2] = {{MGM_ROW(false, false), MGM_ROW(false, true)},
                                        {MGM_ROW(true, false), MGM_ROW(true, true)}};
#undef MGM_ROW
    if (A.tail == 1) return mg_march_launch_tail1(c, A, pow2);
    if (A.tail == 2) return mg_march_launch_tail2(c, A, pow2);
    PYRO_LAUNCH(c, "k_mg_smooth_march", inst[pow2 ? 1 : 0][A.cv ? 1 : 0][edge ? 1 : 0],
                dim3(mg_march_blocks(A)), dim3(64), 0, A);
    return 0;
}

int mg_march_blocks(const MGMarch &A)
{
    return A.nchunks_side > 0 ? (A.ncs - 2) * A.nchunks + 2 * A.nchunks_side : A.ncs * A.nchunks;
}

int mg_march_launch(pyrohip_ctx *c, MGMarch &A, bool pow2, int K)
{
    PYRO_REQUIRE(mgm_has_k(K), "marching smoother: 10 iterations per launch");
    return launch_k<10>(c, A, pow2);
}

// can the level be smoothed by the marching kernel: homogeneous mirror / periodic sides (no
// ghost VALUE 0), and parts that touch at most one of the bottom / top boundaries
bool mg_march_usable(const MGMarch &A, int K)
{
    for (int s = 0; s < 4; s++)
        if (A.code[s] == PYROHIP_BC_CONST) return false;
    if (A.nchunks_side > 0 && (A.ncs < 3 || A.nchunks_side < 2 || A.CR_side + 4 * K + mgm_align(K) >= A.n))
        return false;
    // a window of rows (slab of a decomposed level): whole rows, and a part that ends at the
    // top boundary must not be the window's first (it starts up to mgm_align rows lower:
    // inside the window, not in halo rows the caller did not provide)
    // (nchunks >= 2 below; the shorter chunks of the side strips are not used with windows)
    const bool window = (A.row0 != 1 || A.row1 != A.n);
    if (A.row0 < 1 || A.row1 > A.n || A.row1 - A.row0 + 1 <= A.CR) return false;
    if (window && (A.code[0] == PYROHIP_BC_PERIODIC || A.nchunks_side > 0 || A.CR < mgm_align(K) + 4))
        return false;
    // a tail: whole level, parts of whole coarse rows, the launch it is compiled into
    if (A.tail != 0 && (window || (A.CR & 1) || (A.nchunks_side > 0 && (A.CR_side & 1)) ||
                        (A.tail == 1 ? A.cv != nullptr : A.cv == nullptr)))
        return false;
    return mgm_has_k(K) && A.n % 2 == 0 && A.n >= 2 * MGM_COLS && A.nchunks >= 2 &&
           A.CR + 4 * K + mgm_align(K) < A.n;
}
#endif   // MGM_UNIT

}  // namespace pyro
